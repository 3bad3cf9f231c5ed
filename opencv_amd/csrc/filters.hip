// filters.hip -- Gaussian low/high-pass path of the reference's filter programs on gfx950 (include/sadbm_filters.h).
//
// filter2d_u8_f64_kernel: one 64x16 tile of output pixels per 256-thread block (4 pixels per thread).  The source tile with its
// (krows-1) x (kcols-1) halo is staged once in LDS as bytes, border pixels resolved with BORDER_REFLECT_101 while
// staging; the taps are wave-uniform and come through the scalar cache.  Every thread accumulates its pixel's
// channels in float64, taps in row-major order, multiply and add rounded separately (FP contraction is switched
// off for this file), so the sums equal the CPU definition bit for bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

#include "../../include/sadbm_filters.h"
#include "sadbm_internal.hpp"

// The sums must round after every multiply and every add, as the CPU definition does: no FMA contraction in this
// file (HIP's __dmul_rn/__dadd_rn are header inlines compiled with contraction allowed, so plain operators under
// this pragma are used instead).
#pragma clang fp contract(off)

namespace {

constexpr int TILE = 16;
constexpr int MAXTAPS = 4096;

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// Thread = PX horizontally adjacent output pixels x CH channels; block = 16 x 16 threads = a (16*PX) x 16 pixel tile.
// Per tap row the thread slides a PX-pixel register window along the row: one LDS byte read + one convert per channel
// and tap feeds PX multiply-add pairs, so the kernel issues ~2.5 f64-rate instructions per tap, pixel and channel
// (2 is the floor: the definition rounds the product and the sum separately).
template <int CH, int PX>
__global__ void __launch_bounds__(TILE * TILE) filter2d_u8_f64_kernel(const uint8_t *__restrict__ src, int64_t spitch, int rows, int cols,
                                                                     const double *__restrict__ kern, int krows, int kcols,
                                                                     uint8_t *__restrict__ dst, int64_t dpitch)
{
    extern __shared__ uint8_t tile[];                          // [(TILE+krows-1)][(TILE*PX+kcols-1)*CH]
    const int tw = TILE * PX + kcols - 1, th = TILE + krows - 1, trow = tw * CH;
    const int x0 = blockIdx.x * (TILE * PX), y0 = blockIdx.y * TILE;
    const int ay = krows / 2, ax = kcols / 2;
    for (int i = threadIdx.x; i < th * tw; i += TILE * TILE) {
        const int ty = i / tw, tx = i - ty * tw;
        const uint8_t *p = src + (int64_t)reflect101(y0 + ty - ay, rows) * spitch + (int64_t)reflect101(x0 + tx - ax, cols) * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c) tile[ty * trow + tx * CH + c] = p[c];
    }
    __syncthreads();
    const int lx = threadIdx.x % TILE, ly = threadIdx.x / TILE;
    const int x = x0 + lx * PX, y = y0 + ly;
    double acc[PX][CH];
#pragma unroll
    for (int q = 0; q < PX; ++q)
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[q][c] = 0.0;
    const int jfull = kcols - kcols % PX;
    for (int i = 0; i < krows; ++i) {
        const uint8_t *t = tile + (ly + i) * trow + lx * (PX * CH);
        const double *krow = kern + i * kcols;                 // wave-uniform -> scalar loads
        // win[(j + q) % PX] holds pixel j + q of the row while tap j is applied
        double win[PX][CH];
#pragma unroll
        for (int q = 0; q < PX - 1; ++q)
#pragma unroll
            for (int c = 0; c < CH; ++c) win[q][c] = (double)t[q * CH + c];
        for (int j0 = 0; j0 < jfull; j0 += PX) {
#pragma unroll
            for (int jj = 0; jj < PX; ++jj) {
                const int j = j0 + jj;
#pragma unroll
                for (int c = 0; c < CH; ++c) win[(jj + PX - 1) % PX][c] = (double)t[(j + PX - 1) * CH + c];
                const double k = krow[j];
#pragma unroll
                for (int q = 0; q < PX; ++q)
#pragma unroll
                    for (int c = 0; c < CH; ++c) { const double prod = k * win[(jj + q) % PX][c]; acc[q][c] = acc[q][c] + prod; }
            }
        }
        for (int j = jfull; j < kcols; ++j) {                  // kcols % PX trailing taps: straight from LDS
            const double k = krow[j];
#pragma unroll
            for (int q = 0; q < PX; ++q)
#pragma unroll
                for (int c = 0; c < CH; ++c) { const double prod = k * (double)t[(j + q) * CH + c]; acc[q][c] = acc[q][c] + prod; }
        }
    }
    if (y < rows) {
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            if (x + q >= cols) break;
            uint8_t *o = dst + (int64_t)y * dpitch + (int64_t)(x + q) * CH;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const double r = rint(acc[q][c]);              // cvRound: half to even
                o[c] = (uint8_t)(r < 0.0 ? 0.0 : r > 255.0 ? 255.0 : r);
            }
        }
    }
}

__global__ void __launch_bounds__(256) highpass_kernel(const uint8_t *__restrict__ a, int64_t apitch, const uint8_t *__restrict__ low,
                                                       int64_t lpitch, int rowbytes, int add, uint8_t *__restrict__ dst, int64_t dpitch)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= rowbytes) return;
    int s = a[(int64_t)y * apitch + x] + add;
    s = s > 255 ? 255 : s;
    const int d = s - low[(int64_t)y * lpitch + x];
    dst[(int64_t)y * dpitch + x] = (uint8_t)(d < 0 ? 0 : d);
}

__global__ void __launch_bounds__(256) blend_kernel(const uint8_t *__restrict__ m1, int64_t p1, const uint8_t *__restrict__ m2, int64_t p2,
                                                    int rows, int cols, int ch, double alpha, uint8_t *__restrict__ dst, int64_t dp)
{
    const int xb = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (xb >= cols * ch) return;
    const int x = xb / ch, k = xb - x * ch;
    const uint8_t a = m1[(int64_t)y * p1 + xb];
    uint8_t v = a;
    if (k < 3 && y >= 1 && y < rows - 1 && x >= 1 && x < cols - 1) {
        // hybrid_image.cpp:91: int temp = alpha*a + (1.0-alpha)*b  (double arithmetic, truncation toward zero)
        const double pa = alpha * (double)a, pb = (1.0 - alpha) * (double)m2[(int64_t)y * p2 + xb];
        int t = (int)(pa + pb);
        v = (uint8_t)(t > 255 ? 255 : t < 0 ? 0 : t);
    }
    dst[(int64_t)y * dp + xb] = v;
}

}  // namespace

extern "C" {

int sadbm_gaussian_kernel(int size, int sigma, double *out)
{
    if (!out) return SADBM_E_NULL;
    if (size < 1 || sigma < 1 || (int64_t)size * size > MAXTAPS) return SADBM_E_PARAM;
    const int center = size / 2;                               // gaus_filter.cpp:18
    double sum = 0;
    for (int i = 0; i < size; ++i)
        for (int j = 0; j < size; ++j) {
            // gaus_filter.cpp:26: all-int expression, the division truncates before exp()
            const int q = -((i - center) * (i - center) + (j - center) * (j - center)) / (2 * sigma * sigma);
            out[i * size + j] = exp((double)q);
            sum += out[i * size + j];
        }
    for (int i = 0; i < size * size; ++i) out[i] /= sum;      // gaus_filter.cpp:36
    return SADBM_OK;
}

int sadbm_filter2d_device(sadbm_ctx *ctx, const uint8_t *dSrc, int64_t src_pitch, int rows, int cols, int channels,
                          const double *kernel, int krows, int kcols, uint8_t *dDst, int64_t dst_pitch, void *stream)
{
    if (!ctx || !kernel) return SADBM_E_NULL;
    if (channels != 1 && channels != 3 && channels != 4) return SADBM_E_TYPE;
    if (rows < 0 || cols < 0) return SADBM_E_SIZE;
    if (krows < 1 || kcols < 1 || (int64_t)krows * kcols > MAXTAPS) return SADBM_E_PARAM;
    if (rows == 0 || cols == 0) return SADBM_OK;
    if (!dSrc || !dDst) return SADBM_E_NULL;
    if (src_pitch < (int64_t)cols * channels || dst_pitch < (int64_t)cols * channels) return SADBM_E_SIZE;
    const int dev = sadbm_ctx_device(ctx);
    sadbm_detail::DeviceScope dscope(dev);
    if (dscope.err != hipSuccess) return SADBM_E_HIP;
    hipStream_t st = (hipStream_t)stream;
    // taps: the context's 32 KB device buffer (one per context = one per stream, freed by sadbm_destroy), refilled on
    // the call's stream -- a later call on the same stream overwrites it only after this call's kernel has run
    double *dtaps = nullptr;
    { const int trc = sadbm_ctx_taps(ctx, &dtaps, krows * kcols); if (trc) return trc; }
    if (hipMemcpyAsync(dtaps, kernel, sizeof(double) * krows * kcols, hipMemcpyHostToDevice, st) != hipSuccess) return SADBM_E_HIP;
    // pixels per thread: as many as still give every CU a few blocks (tools/filter_bench.py)
    int px = 8;
    auto nblocks = [&](int q) { return (long)((cols + TILE * q - 1) / (TILE * q)) * ((rows + TILE - 1) / TILE); };
    auto ldsbytes = [&](int q) { return (size_t)(TILE + krows - 1) * (TILE * q + kcols - 1) * channels; };
    while (px > 1 && (nblocks(px) < 512 || ldsbytes(px) > 64 * 1024)) px >>= 1;
    static const int px_env = [] { const char *e = getenv("SADBM_FILTER_PX"); return e ? atoi(e) : 0; }();   // read once
    if (px_env == 1 || px_env == 2 || px_env == 4 || px_env == 8) px = px_env;
    const size_t lds = ldsbytes(px);
    if (lds > 64 * 1024) return SADBM_E_PARAM;
    dim3 grid((cols + TILE * px - 1) / (TILE * px), (rows + TILE - 1) / TILE);
#define SADBM_F2D(CH, PX) hipLaunchKernelGGL((filter2d_u8_f64_kernel<CH, PX>), grid, dim3(TILE * TILE), lds, st, dSrc, src_pitch, rows, cols, \
                                             dtaps, krows, kcols, dDst, dst_pitch)
#define SADBM_F2D_PX(CH) do { if (px == 8) SADBM_F2D(CH, 8); else if (px == 4) SADBM_F2D(CH, 4); else if (px == 2) SADBM_F2D(CH, 2); else SADBM_F2D(CH, 1); } while (0)
    if (channels == 1) SADBM_F2D_PX(1);
    else if (channels == 3) SADBM_F2D_PX(3);
    else SADBM_F2D_PX(4);
#undef SADBM_F2D_PX
#undef SADBM_F2D
    return hipGetLastError() == hipSuccess ? SADBM_OK : SADBM_E_HIP;
}

int sadbm_highpass_device(sadbm_ctx *ctx, const uint8_t *dSrc, int64_t src_pitch, const uint8_t *dLow, int64_t low_pitch, int rows,
                          int rowbytes, int add, uint8_t *dDst, int64_t dst_pitch, void *stream)
{
    if (!ctx) return SADBM_E_NULL;
    if (rows < 0 || rowbytes < 0 || rows > 65535) return SADBM_E_SIZE;
    if (rows == 0 || rowbytes == 0) return SADBM_OK;
    if (!dSrc || !dLow || !dDst) return SADBM_E_NULL;
    if (src_pitch < rowbytes || low_pitch < rowbytes || dst_pitch < rowbytes) return SADBM_E_SIZE;
    sadbm_detail::DeviceScope dscope(sadbm_ctx_device(ctx));
    if (dscope.err != hipSuccess) return SADBM_E_HIP;
    hipLaunchKernelGGL(highpass_kernel, dim3((rowbytes + 255) / 256, rows), dim3(256), 0, (hipStream_t)stream, dSrc, src_pitch, dLow,
                       low_pitch, rowbytes, add, dDst, dst_pitch);
    return hipGetLastError() == hipSuccess ? SADBM_OK : SADBM_E_HIP;
}

int sadbm_blend_device(sadbm_ctx *ctx, const uint8_t *dM1, int64_t pitch1, const uint8_t *dM2, int64_t pitch2, int rows, int cols,
                       int channels, double alpha, uint8_t *dDst, int64_t dst_pitch, void *stream)
{
    if (!ctx) return SADBM_E_NULL;
    if (channels < 3 || channels > 4) return SADBM_E_TYPE;
    if (rows < 0 || cols < 0 || rows > 65535) return SADBM_E_SIZE;
    if (rows == 0 || cols == 0) return SADBM_OK;
    if (!dM1 || !dM2 || !dDst) return SADBM_E_NULL;
    const int64_t rowb = (int64_t)cols * channels;
    if (pitch1 < rowb || pitch2 < rowb || dst_pitch < rowb) return SADBM_E_SIZE;
    sadbm_detail::DeviceScope dscope(sadbm_ctx_device(ctx));
    if (dscope.err != hipSuccess) return SADBM_E_HIP;
    hipLaunchKernelGGL(blend_kernel, dim3((unsigned)((rowb + 255) / 256), rows), dim3(256), 0, (hipStream_t)stream, dM1, pitch1, dM2, pitch2,
                       rows, cols, channels, alpha, dDst, dst_pitch);
    return hipGetLastError() == hipSuccess ? SADBM_OK : SADBM_E_HIP;
}

}  // extern "C"

namespace {
// device images of one host program run: rows x pitch bytes each, freed on scope exit
struct DevImages {
    uint8_t *p[6] = {};
    ~DevImages() { for (uint8_t *q : p) if (q) (void)hipFree(q); }
};
int check_host_image(const sadbm_image *im)
{
    if (!im) return SADBM_E_NULL;
    if (im->channels != 1 && im->channels != 3 && im->channels != 4) return SADBM_E_TYPE;
    if (im->rows < 0 || im->cols < 0 || im->rows > 65535) return SADBM_E_SIZE;
    if (im->rows && im->cols && (!im->data || im->step < (int64_t)im->cols * im->channels)) return im->data ? SADBM_E_SIZE : SADBM_E_NULL;
    return SADBM_OK;
}
}  // namespace

extern "C" {

int sadbm_hybrid_image_host(const sadbm_image *img1, const sadbm_image *img2, int size, int sigma, int add, double alpha,
                            uint8_t *low, int64_t low_step, uint8_t *high, int64_t high_step, uint8_t *hybrid, int64_t hybrid_step)
{
    int rc = check_host_image(img1);
    if (rc) return rc;
    const bool two = img2 != nullptr;                          // one image: the gaus_filter program (high from the same image)
    if (two) {
        if ((rc = check_host_image(img2))) return rc;
        if (img2->rows != img1->rows || img2->cols != img1->cols) return SADBM_E_SIZE;
        if (img2->channels != img1->channels) return SADBM_E_TYPE;
        if (hybrid && img1->channels < 3) return SADBM_E_TYPE;  // add() indexes three channels (hybrid_image.cpp:89)
    } else if (hybrid) return SADBM_E_NULL;
    if (size < 1 || sigma < 1 || (int64_t)size * size > MAXTAPS) return SADBM_E_PARAM;
    const int rows = img1->rows, cols = img1->cols, ch = img1->channels;
    if (rows == 0 || cols == 0) return SADBM_OK;
    const size_t rowb = (size_t)cols * ch, pitch = (rowb + 255) & ~(size_t)255, bytes = pitch * rows;
    if ((low && low_step < (int64_t)rowb) || (high && high_step < (int64_t)rowb) || (hybrid && hybrid_step < (int64_t)rowb)) return SADBM_E_SIZE;
    std::vector<double> taps((size_t)size * size);
    if ((rc = sadbm_gaussian_kernel(size, sigma, taps.data()))) return rc;
    sadbm_ctx *ctx = nullptr;
    int dev = 0;                                               // everything below runs on the calling thread's current device
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    if ((rc = sadbm_create(dev, &ctx))) return rc;
    sadbm_detail::DeviceScope dscope(dev);
    DevImages d;                                               // 0 img1, 1 img2, 2 low, 3 tmp, 4 high, 5 hybrid
    hipStream_t st = nullptr;                                  // the device's default stream: the copies below are synchronous
    auto fail = [&](int code) { sadbm_destroy(ctx); return code; };
    for (int i = 0; i < 6; ++i)
        if (hipMalloc((void **)&d.p[i], bytes) != hipSuccess) { d.p[i] = nullptr; return fail(SADBM_E_ALLOC); }
    if (hipMemcpy2D(d.p[0], pitch, img1->data, (size_t)img1->step, rowb, rows, hipMemcpyHostToDevice) != hipSuccess) return fail(SADBM_E_HIP);
    const sadbm_image *src2 = two ? img2 : img1;
    if (hipMemcpy2D(d.p[1], pitch, src2->data, (size_t)src2->step, rowb, rows, hipMemcpyHostToDevice) != hipSuccess) return fail(SADBM_E_HIP);
    if ((rc = sadbm_filter2d_device(ctx, d.p[0], (int64_t)pitch, rows, cols, ch, taps.data(), size, size, d.p[2], (int64_t)pitch, st))) return fail(rc);
    const uint8_t *blur2 = d.p[2];
    if (two) {
        if ((rc = sadbm_filter2d_device(ctx, d.p[1], (int64_t)pitch, rows, cols, ch, taps.data(), size, size, d.p[3], (int64_t)pitch, st))) return fail(rc);
        blur2 = d.p[3];
    }
    if ((rc = sadbm_highpass_device(ctx, d.p[1], (int64_t)pitch, blur2, (int64_t)pitch, rows, (int)rowb, add, d.p[4], (int64_t)pitch, st))) return fail(rc);
    if (hybrid && (rc = sadbm_blend_device(ctx, d.p[2], (int64_t)pitch, d.p[4], (int64_t)pitch, rows, cols, ch, alpha, d.p[5], (int64_t)pitch, st))) return fail(rc);
    if (low && hipMemcpy2D(low, (size_t)low_step, d.p[2], pitch, rowb, rows, hipMemcpyDeviceToHost) != hipSuccess) return fail(SADBM_E_HIP);
    if (high && hipMemcpy2D(high, (size_t)high_step, d.p[4], pitch, rowb, rows, hipMemcpyDeviceToHost) != hipSuccess) return fail(SADBM_E_HIP);
    if (hybrid && hipMemcpy2D(hybrid, (size_t)hybrid_step, d.p[5], pitch, rowb, rows, hipMemcpyDeviceToHost) != hipSuccess) return fail(SADBM_E_HIP);
    if (hipDeviceSynchronize() != hipSuccess) return fail(SADBM_E_HIP);
    sadbm_destroy(ctx);
    return SADBM_OK;
}

int sadbm_gaus_filter_host(const sadbm_image *src, int size, int sigma, uint8_t *low, int64_t low_step, uint8_t *high, int64_t high_step)
{
    return sadbm_hybrid_image_host(src, nullptr, size, sigma, 0, 0.0, low, low_step, high, high_step, nullptr, 0);
}

}  // extern "C"
