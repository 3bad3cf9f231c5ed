// L1 (TCP) rate of per-lane 16-byte global loads as the strip matcher's staging issues them (lane i reads 16 bytes at
// base + 16 i + off): 16-byte aligned, dword aligned (off = 4, 8), byte aligned (off = 1, 2, 3); plus dwordx2 / dword loads
// at the same alignments.  The buffer is small (L1/L2 resident).  Run on the GPU box.
// Build: hipcc -O3 --offload-arch=gfx950 ubench4.hip -o ubench4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef uint32_t u32;
typedef u32 u32x4u __attribute__((ext_vector_type(4), aligned(1)));
typedef u32 u32x2u __attribute__((ext_vector_type(2), aligned(1)));
typedef u32 u32u __attribute__((aligned(1)));
constexpr int ITERS = 512, UNR = 8;

template <int WIDTH>   // 4, 2, 1 dwords per lane
__global__ void __launch_bounds__(256) loads(const uint8_t *buf, int off, int rowstride, u32 *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t *p = buf + off + lane * (4 * WIDTH) + wave * 4096 + (blockIdx.x & 7) * 16384;
    u32 acc = 0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const uint8_t *q = p + ((it * UNR + u) & 15) * rowstride;
            if (WIDTH == 4) { u32x4u v = *(const u32x4u *)q; acc += v.x ^ v.y ^ v.z ^ v.w; }
            if (WIDTH == 2) { u32x2u v = *(const u32x2u *)q; acc += v.x ^ v.y; }
            if (WIDTH == 1) { acc += *(const u32u *)q; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int WIDTH> int run(const uint8_t *d, u32 *o, int off, int cus, double ghz)
{
    const int blocks = cus * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(loads<WIDTH>, dim3(blocks), dim3(256), 0, 0, d, off, 1024, o);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(loads<WIDTH>, dim3(blocks), dim3(256), 0, 0, d, off, 1024, o);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double winst = (double)blocks * 4 * ITERS * UNR;
    printf("global_load_dwordx%d  off %2d: %7.3f ms  %6.1f cycles per wave-load per CU  (%5.1f B/clk/CU)\n", WIDTH, off, ms,
           ms * 1e-3 * ghz * 1e9 * cus / winst, winst * 64 * 4 * WIDTH / (ms * 1e-3 * ghz * 1e9 * cus));
    return 0;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
    printf("CUs %d clk %.2f GHz\n", cus, ghz);
    uint8_t *d; u32 *o;
    CK(hipMalloc((void **)&d, 1 << 20)); CK(hipMemset(d, 1, 1 << 20)); CK(hipMalloc((void **)&o, (size_t)cus * 8 * 256 * 4));
    for (int off : {0, 4, 8, 1, 2, 3, 5, 17}) if (run<4>(d, o, off, cus, ghz)) return 1;
    for (int off : {0, 4, 1, 3}) if (run<2>(d, o, off, cus, ghz)) return 1;
    for (int off : {0, 1, 3}) if (run<1>(d, o, off, cus, ghz)) return 1;
    return 0;
}
