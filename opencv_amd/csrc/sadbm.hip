// sadbm.hip -- SAD stereo block matching for gfx950 (MI355X): HIP kernels + the C ABI of include/sadbm.h.
//
// Replaces getDisparity()/sub_kernel() of /root/reference/SAD.cpp:13-61 (see include/sadbm.h for the
// line-by-line semantics).  Written from scratch for CDNA4; nothing here is derived from the reference's
// OpenCV-call-per-window structure.
//
// Pipeline per call (all on one stream, run_pipeline below):
//   1. zero_rows_kernel      : the zero frame (rows >= rows-win), SAD.cpp:31,34 -- one launch for the whole batch
//                              (not on the strip path: its last band of blocks writes those rows).
//   2. prep_plane_kernel     : channel 0 of both inputs -> zero-padded single-channel planes in ctx scratch
//                              (PADL bytes left, PADR right) so the matchers never need bounds checks; for batches on
//                              the strip matcher several frames side by side per plane (strip_frame_groups).  Skipped
//                              where the on-the-fly strip kernels read single-channel inputs themselves ("+direct").
//   3. one matcher           : sad_strip_otf_kernel<WIN,NG,DIRECT> (strip_otf_kernels.hip: 16 columns x 4 disparities per
//                              thread, plain rows staged in LDS, leftover-byte operands packed per thread) up to 64
//                              disparities; sad_strip_kernel<WIN,NSLM> (sad_strip.inc: the same work split with every operand
//                              staged as a ready-made dword in LDS) above; sad_quad_lds_kernel<WIN,KC> (quad_kernels.hip) for
//                              windows below 4 columns and a few 4-wide cases; sad_quad_kernel<WIN> = the same arithmetic
//                              with operands through L1 (A/B flag); sad_wide_kernel (column sums) for win > QUAD_MAXWIN (49);
//                              sad_generic_kernel (brute force) under SADBM_FORCE_GENERIC.
//
// The quad matcher lives in quad_kernels.hip, the strip matchers in sad_strip.inc / strip_kernels.hip /
// strip_otf_kernels.hip (separate translation units so they compile in parallel); sadbm_internal.hpp holds what they share.
// DESIGN.md section 4 has the measurements behind each choice.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <condition_variable>
#include <functional>
#include <immintrin.h>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "sadbm_internal.hpp"
#include "hostcopy.hpp"

using namespace sadbm_detail;

namespace {

#ifndef SADBM_MINBLOCKS
#define SADBM_MINBLOCKS 2000
#endif
#ifndef SADBM_TYMAX
#define SADBM_TYMAX 128
#endif
constexpr int KCG = 16;       // disparities per thread of the L1-operand variant (sad_quad_kernel)

thread_local char g_hip_err[256] = "";

// ---------------------------------------------------------------------------------------------
// 1. plane preparation: dst[y][PADL + x] = src[y][x*ch] (channel 0), zero elsewhere.
//    One thread writes 16 aligned bytes of dst.
// ---------------------------------------------------------------------------------------------
typedef uint32_t prep_u32x4u __attribute__((ext_vector_type(4), aligned(1)));
#ifndef SADBM_PREP_ROWS
#define SADBM_PREP_ROWS 2
#endif
constexpr int PREP_ROWS = SADBM_PREP_ROWS;
// Plane z of dst holds FG frames side by side: frame f of the group at columns [PADL + f*Wp, PADL + f*Wp + cols), zeros
// elsewhere (FG = 1: the plain padded plane of one frame).  Wp % 16 == 0 whenever FG > 1.
// blockIdx.z = 2 * group + (0: left image, 1: right image); dst planes are plane_bytes apart in that order.
__global__ void __launch_bounds__(1024) prep_plane_kernel(const uint8_t *__restrict__ srcL, int64_t stepL,
                                                           const uint8_t *__restrict__ srcR, int64_t stepR, int64_t src_stride,
                                                           int ch, int rows, int cols, uint8_t *__restrict__ dst,
                                                           int pitch, int64_t plane_bytes, int FG, int Wp, uint32_t Wp_magic, int n_pairs)
{
    const bool right = (blockIdx.z & 1) != 0;
    const uint8_t *src = right ? srcR : srcL;
    const int64_t step = right ? stepR : stepL;
    const int group = blockIdx.z >> 1;
    // one thread writes 16 aligned bytes of PREP_ROWS consecutive rows of dst (pitch is a multiple of 128); the loads of
    // all rows are issued before the first store
    const int wi = blockIdx.x * blockDim.x + threadIdx.x;
    const int y0 = blockIdx.y * PREP_ROWS;
    if (wi * 16 >= pitch || y0 >= rows) return;
    const int xv = wi * 16 - PADL;
    int f = 0, xb = xv;
    if (FG > 1 && xv >= 0) { f = (int)__umulhi((uint32_t)xv, Wp_magic); xb = xv - f * Wp; }   // xv / Wp: exact for xv, Wp < 65536
    const int pair = group * FG + f;
    uint4 v[PREP_ROWS];
#pragma unroll
    for (int r = 0; r < PREP_ROWS; ++r) v[r] = make_uint4(0, 0, 0, 0);
    if (f < FG && pair < n_pairs) {
        const uint8_t *s0 = src + (int64_t)pair * src_stride + (int64_t)y0 * step;
        if (ch == 1 && xb >= 0 && xb + 16 <= cols) {
#pragma unroll
            for (int r = 0; r < PREP_ROWS; ++r)
                if (y0 + r < rows) {
                    const prep_u32x4u q = *(const prep_u32x4u *)(s0 + (int64_t)r * step + xb);
                    v[r] = make_uint4(q.x, q.y, q.z, q.w);
                }
        } else if (ch == 3 && xb >= 0 && xb + 16 <= cols) {
            // BGR (what cv::imread hands the reference, SAD.cpp:65-66): 48 bytes in three 16-byte loads, every third byte kept
#pragma unroll
            for (int r = 0; r < PREP_ROWS; ++r)
                if (y0 + r < rows) {
                    const uint8_t *s = s0 + (int64_t)r * step + (int64_t)xb * 3;
                    const prep_u32x4u a = *(const prep_u32x4u *)s, b = *(const prep_u32x4u *)(s + 16), c = *(const prep_u32x4u *)(s + 32);
                    const uint32_t D[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
                    uint32_t w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t t = __builtin_amdgcn_perm(D[3 * j + 1], D[3 * j], 0x0C060300u);   // bytes 0, 3, 6 of the 12
                        w[j] = __builtin_amdgcn_perm(D[3 * j + 2], t, 0x05020100u);                      // + byte 9
                    }
                    v[r] = make_uint4(w[0], w[1], w[2], w[3]);
                }
        } else if (ch == 4 && xb >= 0 && xb + 16 <= cols) {
#pragma unroll
            for (int r = 0; r < PREP_ROWS; ++r)
                if (y0 + r < rows) {
                    const uint8_t *s = s0 + (int64_t)r * step + (int64_t)xb * 4;
                    uint32_t w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const prep_u32x4u q = *(const prep_u32x4u *)(s + 16 * j);
                        const uint32_t t1 = __builtin_amdgcn_perm(q.y, q.x, 0x0C0C0400u), t2 = __builtin_amdgcn_perm(q.w, q.z, 0x0C0C0400u);
                        w[j] = __builtin_amdgcn_perm(t2, t1, 0x05040100u);
                    }
                    v[r] = make_uint4(w[0], w[1], w[2], w[3]);
                }
        } else if (ch == 1 && xb >= 0 && xb < cols && cols >= 16) {
            // last, partial chunk of a row: the 16 bytes that END at the row's end, shifted down (never reads past the row)
            const int sb = 8 * (16 - (cols - xb));            // 8 .. 120 bits
#pragma unroll
            for (int r = 0; r < PREP_ROWS; ++r)
                if (y0 + r < rows) {
                    const prep_u32x4u q = *(const prep_u32x4u *)(s0 + (int64_t)r * step + cols - 16);
                    const uint64_t lo = (uint64_t)q.x | ((uint64_t)q.y << 32), hi = (uint64_t)q.z | ((uint64_t)q.w << 32);
                    const uint64_t rlo = sb >= 64 ? hi >> (sb - 64) : (lo >> sb) | (hi << (64 - sb));
                    const uint64_t rhi = sb >= 64 ? 0 : hi >> sb;
                    v[r] = make_uint4((uint32_t)rlo, (uint32_t)(rlo >> 32), (uint32_t)rhi, (uint32_t)(rhi >> 32));
                }
        } else if (xb + 16 > 0 && xb < cols) {
            for (int r = 0; r < PREP_ROWS && y0 + r < rows; ++r) {
                const uint8_t *s = s0 + (int64_t)r * step;
                uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const int x = xb + b;
                    if (x >= 0 && x < cols) w[b >> 2] |= (uint32_t)s[(int64_t)x * ch] << (8 * (b & 3));
                }
                v[r] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    }
    uint8_t *d = dst + (int64_t)blockIdx.z * plane_bytes + (int64_t)y0 * pitch + wi * 16;
#pragma unroll
    for (int r = 0; r < PREP_ROWS; ++r)
        if (y0 + r < rows) *(uint4 *)(d + (int64_t)r * pitch) = v[r];
}

// The zero frame of SAD.cpp:31,34: output rows [row0, row0+nrows) of every pair (any pitch / alignment).
__global__ void __launch_bounds__(256) zero_rows_kernel(uint8_t *out, int64_t out_pitch, int64_t out_stride, int row0, int nrows, int cols)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x >= cols) return;
    uint8_t *o = out + (int64_t)blockIdx.z * out_stride + (int64_t)(row0 + blockIdx.y) * out_pitch + x;
    if (x + 3 < cols && (((uintptr_t)o) & 3) == 0) { *(uint32_t *)o = 0; return; }
    for (int b = 0; b < 4 && x + b < cols; ++b) o[b] = 0;
}

// Brute-force matcher for any window and any search range (SADBM_FORCE_GENERIC; windows beyond 255 columns): one thread per
// output pixel, window SAD with v_sad_u8 on words from the padded planes.  Correctness path, not tuned.  Keys are 64-bit
// (cost << 32 | k), and the cost takes the reference's round trip through `float sub_Sum` (SAD.cpp:10,20-21), which stops
// being exact at 2^24, i.e. from 257 x 257 windows on.
__global__ void __launch_bounds__(256) sad_generic_kernel(Params p)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= p.W || y >= p.nvalid) return;
    const uint8_t *L = p.L + (int64_t)blockIdx.z * p.plane_stride;
    const uint8_t *R = p.R + (int64_t)blockIdx.z * p.plane_stride;
    uint8_t *out = p.out + (int64_t)blockIdx.z * p.out_stride + (int64_t)y * p.out_pitch;
    const int ocols = p.W - p.win;
    if (x >= ocols) { out[x] = 0; return; }
    const int kmax = min(x, p.D - 1);
    uint64_t best = ~0ull;
    // four bytes per v_sad_u8 (byte-granular dword loads; the planes are padded, so the word that holds the window's last
    // 1-3 bytes is readable -- its other bytes are masked away in both operands)
    typedef uint32_t u32u __attribute__((aligned(1)));
    const int nw = p.win >> 2, rem = p.win & 3;
    const uint32_t tail_mask = rem ? (0xFFFFFFFFu >> (8 * (4 - rem))) : 0u;
    for (int k = 0; k <= kmax; ++k) {
        uint32_t cost = 0;                                    // < 2^32 for win <= 4096
        for (int dy = 0; dy < p.win; ++dy) {
            const uint8_t *pl = L + (int64_t)(y + dy) * p.pitch + x;
            const uint8_t *pr = R + (int64_t)(y + dy) * p.pitch + x - k;
            for (int w = 0; w < nw; ++w) cost = __builtin_amdgcn_sad_u8(*(const u32u *)(pl + 4 * w), *(const u32u *)(pr + 4 * w), cost);
            if (rem) cost = __builtin_amdgcn_sad_u8(*(const u32u *)(pl + 4 * nw) & tail_mask, *(const u32u *)(pr + 4 * nw) & tail_mask, cost);
        }
        // sub_Sum = (float)Add[0]; return (int)sub_Sum  (SAD.cpp:20-21); Temp.at<float> = that int (SAD.cpp:48): the compare of
        // SAD.cpp:54 sees float(int(float(cost))) -- monotone in the middle value, which is an exact integer
        const uint32_t c = (uint32_t)(int)(float)cost;
        best = min(best, ((uint64_t)c << 32) | (uint32_t)k);
    }
    if (x + 1 < p.D) best = min(best, (uint64_t)(uint32_t)(x + 1));
    const uint32_t loc = (uint32_t)best;
    out[x] = (p.flags & SADBM_RAW_LOC) ? (uint8_t)loc : (uint8_t)((loc * 16u) & 0xFFu);
}

// Inclusive prefix sum over the 64 lanes of a wave (row_shr 1, 2, 4, 8 inside the rows of 16, then the gfx9 row broadcasts).
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);      // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);      // row_bcast:31 into rows 2 and 3
    return v;
}

// Windows beyond the quad matcher (50..255 columns): ~25 lane-ops per evaluation whatever the window, instead of the brute
// force's O(win^2 / 4).  A block is 256 adjacent columns x 8, 16 or 32 output rows x 64 disparities per pass.
//   1. Thread t keeps the vertical running sums of |L - R| over the window's rows in column XB + t for the 64 disparities
//      (SAD.cpp:13-22's sum split into columns; <= 255 * 255: packed 16-bit, four per v_mqsad_pk_u16_u8 -- its reference
//      operand is the L byte alone, the three zero bytes are masked out; an L byte of 0 would be masked too, so those lanes
//      match 255 against the complemented R bytes).  One row enters and one leaves per step.
//   2. Per output row and 16 disparities the block turns the sums into prefix sums P over its 256 columns (DPP scan
//      in each wave, wave totals and P through LDS); thread t < NX = 257 - win takes P[t + win - 1] - P[t - 1].
//   3. First-minimum key = cost << 8 | k (cost < 255^3 < 2^24); with D > 64 it waits in a thread-private LDS cell between passes.
constexpr int WIDE_TY = 32;
constexpr int WIDE_LDS = (4 * 256 + 16) * 16;                // prefix sums | wave totals
__global__ void __launch_bounds__(256) sad_wide_kernel(Params p)
{
    extern __shared__ uint4 wide_lds[];
    uint4 *const P = wide_lds, *const Wt = wide_lds + 4 * 256;                  // prefix sums [4 quarters][256] | wave totals [4 quarters][4]
    uint32_t *const best_s = (uint32_t *)(wide_lds + 4 * 256 + 16);             // [WIDE_TY][256], allocated when D > 64
    typedef uint32_t u32u __attribute__((aligned(1)));
    const int tid = threadIdx.x, NX = 257 - p.win;
    const int xc = blockIdx.x * NX + tid;
    const int yb = blockIdx.y * p.TY, ny = min(p.TY, p.nvalid - yb);           // p.TY <= WIDE_TY
    const uint8_t *L = p.L + (int64_t)blockIdx.z * p.plane_stride + (int64_t)yb * p.pitch + xc;
    const uint8_t *R = p.R + (int64_t)blockIdx.z * p.plane_stride + (int64_t)yb * p.pitch + xc;
    uint8_t *out = p.out + (int64_t)blockIdx.z * p.out_stride + (int64_t)yb * p.out_pitch + xc;
    const int ocols = p.W - p.win;
    const bool col_ok = xc < p.W;                      // columns beyond the frame feed no output column (x + win - 1 < W - 1)
    const bool out_ok = tid < NX && xc < ocols;
    const int kmax = min(xc, p.D - 1);                 // slots k > column keep cost 0 (SAD.cpp:40 skips them): the x + 1 key below
    const int lane = tid & 63, wave = tid >> 6;
    for (int kg = 0; kg < p.D; kg += 64) {
        uint32_t cs[8][4];                             // chunk c (k = kg + 8c + j): {j3,j2} {j1,j0} {j7,j6} {j5,j4} as 16-bit pairs
#pragma unroll
        for (int c = 0; c < 8; ++c) cs[c][0] = cs[c][1] = cs[c][2] = cs[c][3] = 0;
        for (int r = 0; r < ny + p.win - 1; ++r) {
            const bool leave = r >= p.win, emit = r >= p.win - 1;
            uint32_t rw[16], qw[16], lref = 0, mref = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) { rw[i] = 0; qw[i] = 0; }
            // columns left of kg hold no competing slot in this pass (k > column, SAD.cpp:44) and feed no window that does: they
            // load nothing, so the look-back stays within 63 columns of column 0 (the planes' left padding) whatever D is
            if (col_ok && xc >= kg) {
                // disparity kg + i pairs L[xc] with R[xc - kg - i]: the 64 bytes that end at R[xc - kg] (left padding: zeros)
                const uint8_t *pr = R + (int64_t)r * p.pitch - kg - 63;
                const uint32_t lb = L[(int64_t)r * p.pitch], lm = lb ? 0u : 0xFFFFFFFFu;
                lref = lb ? lb : 255u;
#pragma unroll
                for (int i = 0; i < 16; ++i) rw[i] = *(const u32u *)(pr + 4 * i) ^ lm;
                if (leave) {
                    const uint8_t *qr = pr - (int64_t)p.win * p.pitch;
                    const uint32_t mb = L[(int64_t)(r - p.win) * p.pitch], mm = mb ? 0u : 0xFFFFFFFFu;
                    mref = mb ? mb : 255u;
#pragma unroll
                    for (int i = 0; i < 16; ++i) qw[i] = *(const u32u *)(qr + 4 * i) ^ mm;
                }
            }
            // keys are cost << 8 | (k & 255): within a pass the minimum is the first minimum; across passes (ascending k, D up to
            // any size) a later pass only wins with a strictly smaller cost (SAD.cpp:54)
            uint32_t b = KEY_MAX, bprev = KEY_MAX;
            if (emit && out_ok && kg > 0) bprev = best_s[(r - (p.win - 1)) * 256 + tid];
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {           // two chunks of 8 disparities per exchange
                if (kg + 16 * c2 < p.D) {              // uniform over the block (slots >= D of the pair are summed but never compete)
                    if (col_ok) {
#pragma unroll
                        for (int c = 2 * c2; c < 2 * c2 + 2; ++c) {
                            // word 15 - 2c holds j = 3..0 in bytes 0..3, word 14 - 2c holds j = 7..4
                            uint64_t a1 = (uint64_t)cs[c][0] | ((uint64_t)cs[c][1] << 32), a0 = (uint64_t)cs[c][2] | ((uint64_t)cs[c][3] << 32);
                            a1 = __builtin_amdgcn_mqsad_pk_u16_u8((uint64_t)rw[15 - 2 * c], lref, a1);
                            a0 = __builtin_amdgcn_mqsad_pk_u16_u8((uint64_t)rw[14 - 2 * c], lref, a0);
                            cs[c][0] = (uint32_t)a1; cs[c][1] = (uint32_t)(a1 >> 32); cs[c][2] = (uint32_t)a0; cs[c][3] = (uint32_t)(a0 >> 32);
                            if (leave) {               // every 16-bit sum holds the leaving row's share: no borrow crosses a field
                                const uint64_t d1 = __builtin_amdgcn_mqsad_pk_u16_u8((uint64_t)qw[15 - 2 * c], mref, 0ull);
                                const uint64_t d0 = __builtin_amdgcn_mqsad_pk_u16_u8((uint64_t)qw[14 - 2 * c], mref, 0ull);
                                cs[c][0] -= (uint32_t)d1; cs[c][1] -= (uint32_t)(d1 >> 32); cs[c][2] -= (uint32_t)d0; cs[c][3] -= (uint32_t)(d0 >> 32);
                            }
                        }
                    }
                    if (emit) {
                        // prefix sums over the block's 256 columns, 16 disparities at a time: DPP scan inside the wave, wave totals
                        // through LDS; the window sum of column x is then P[x + win - 1] - P[x - 1]  (P < 256 * 65025 < 2^24)
                        uint32_t v[16];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const uint32_t w = cs[2 * c2 + (q >> 2)][q & 3];
                            v[2 * q] = wave_scan_add(w & 0xFFFFu); v[2 * q + 1] = wave_scan_add(w >> 16);
                        }
                        if (lane == 63) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) Wt[4 * q + wave] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                        }
                        __syncthreads();
                        for (int w = 0; w < wave; ++w) {           // wave-uniform
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const uint4 u = Wt[4 * q + w];
                                v[4 * q] += u.x; v[4 * q + 1] += u.y; v[4 * q + 2] += u.z; v[4 * q + 3] += u.w;
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) P[256 * q + tid] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                        __syncthreads();
                        if (out_ok) {
                            uint32_t a[16];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const uint4 u = P[256 * q + tid + p.win - 1];                      // tid + win - 1 <= 255
                                a[4 * q] = u.x; a[4 * q + 1] = u.y; a[4 * q + 2] = u.z; a[4 * q + 3] = u.w;
                            }
                            if (tid > 0) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const uint4 u = P[256 * q + tid - 1];
                                    a[4 * q] -= u.x; a[4 * q + 1] -= u.y; a[4 * q + 2] -= u.z; a[4 * q + 3] -= u.w;
                                }
                            }
                            // ascending k; a[8h + s]: slot s of chunk 2 c2 + h, slots {j3, j2, j1, j0, j7, j6, j5, j4}
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const int k0 = kg + 16 * c2 + 8 * h;
#pragma unroll
                                for (int j = 0; j < 8; ++j) {
                                    const int sidx = 8 * h + (j < 4 ? 3 - j : 11 - j);
                                    if (k0 + j <= kmax) b = min(b, (a[sidx] << 8) | (uint32_t)((k0 + j) & 255));
                                }
                            }
                        }
                    }
                }
            }
            if (emit && tid < NX && xc < p.W) {
                if ((bprev >> 8) <= (b >> 8)) b = bprev;             // KEY_MAX >> 8 exceeds every cost (< 2^24)
                if (kg + 64 < p.D) {
                    if (out_ok) best_s[(r - (p.win - 1)) * 256 + tid] = b;
                } else {
                    // slots k > column keep cost 0 (SAD.cpp:40): the first of them, k = column + 1, wins unless a valid slot has cost 0
                    if (xc + 1 < p.D && (b >> 8) != 0) b = (uint32_t)((xc + 1) & 255);
                    const uint32_t loc = b & 0xFFu;
                    out[(int64_t)(r - (p.win - 1)) * p.out_pitch] = !out_ok ? (uint8_t)0 : (p.flags & SADBM_RAW_LOC) ? (uint8_t)loc : (uint8_t)((loc * 16u) & 0xFFu);
                }
            }
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct sadbm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;          // owned
    uint8_t *planes = nullptr;             // scratch: L planes then R planes
    size_t planes_cap = 0;
    bool planes_fixed = false;             // planes is not ours to grow or free (sadbm_compute_device_static)
    bool host_inputs = false;              // set by the host API round its zero-copy call: the inputs are pinned host memory
    uint8_t *stage_in = nullptr;           // host API staging (device copies of the caller's host images)
    size_t stage_in_cap = 0;
    uint8_t *stage_out = nullptr;
    size_t stage_out_cap = 0;
    uint8_t *pin_in = nullptr, *pin_out = nullptr;   // pinned host staging of the host API (pageable -> device copies are ~20x slower)
    size_t pin_in_cap = 0, pin_out_cap = 0;
    hipStream_t stream2 = nullptr, stream3 = nullptr;   // host API, large frames: D2H of finished row chunks / H2D of the next chunk (lazily created)
    hipEvent_t chunk_ev[3 * 16] = {};      // per chunk: computed / copied back / uploaded
    uint8_t *pin_in_dev = nullptr, *pin_out_dev = nullptr;   // device addresses of the pinned buffers (zero-copy path), resolved when they grow
    double *taps = nullptr;                // filter taps of sadbm_filter2d_device (filters.hip), one buffer per context
    char last_kernel[64] = "none";
    int cus = 256;                         // compute units of the device (grid shaping)
    // Calls on one context are serialised (mu) and ordered: the scratch planes are rewritten by every call, so when the
    // caller's stream changes the new stream first waits for the work the previous one still holds (xev).
    std::recursive_mutex mu;      // serialises calls on this context; recursive: the host call holds it across its device calls
    hipStream_t last_stream = nullptr;
    bool has_last = false;
    hipEvent_t xev = nullptr;
    // profiling: event triples (before prep, before matcher, after matcher) per pipeline run
    hipEvent_t *ev = nullptr;
    int ev_cap = 0, ev_n = 0, prof_on = 0;
    std::vector<uint8_t> ev_noprep;                            // per profiled launch: no prep pass was launched (inputs read in place)
};

namespace {

int hip_fail(hipError_t e, const char *what)
{
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", what, hipGetErrorString(e));
    return SADBM_E_HIP;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(e_, #call); } while (0)

int grow(uint8_t **buf, size_t *cap, size_t need)
{
    if (*cap >= need) return SADBM_OK;
    if (*buf) { hipError_t e = hipFree(*buf); *buf = nullptr; *cap = 0; if (e != hipSuccess) return hip_fail(e, "hipFree"); }
    size_t want = need + need / 8 + 4096;
    hipError_t e = hipMalloc((void **)buf, want);
    if (e != hipSuccess) { *buf = nullptr; hip_fail(e, "hipMalloc"); return SADBM_E_ALLOC; }
    *cap = want;
    return SADBM_OK;
}

int grow_pinned(uint8_t **buf, size_t *cap, size_t need)
{
    if (*cap >= need) return SADBM_OK;
    if (*buf) { hipError_t e = hipHostFree(*buf); *buf = nullptr; *cap = 0; if (e != hipSuccess) return hip_fail(e, "hipHostFree"); }
    size_t want = need + need / 8 + 4096;
    hipError_t e = hipHostMalloc((void **)buf, want, hipHostMallocDefault);
    if (e != hipSuccess) { *buf = nullptr; hip_fail(e, "hipHostMalloc"); return SADBM_E_ALLOC; }
    *cap = want;
    return SADBM_OK;
}

// host-side row copies of the host API (channel-0 gather with SSSE3 / AVX2, persistent copy pool): hostcopy.hpp, which the
// sanitizer build compiles on its own (make asan -> hostcopy_asan_selftest)
using sadbm_hostcopy::copy_rows;

int check_params(int rows, int cols, int channels, int win, int max_disp, uint32_t flags)
{
    if (rows < 0 || cols < 0) return SADBM_E_SIZE;
    if (channels != 1 && channels != 3 && channels != 4) return SADBM_E_TYPE;
    // The reference takes any winSize / DSR (plain ints, SAD.cpp:9,11).  Here: raw loc output is a byte, so SADBM_RAW_LOC needs
    // max_disp <= 256 (the reference's own encoding, (loc*16) mod 256, has no such limit); the brute-force matcher that serves
    // windows beyond 255 columns sums costs in 32 bits, i.e. up to 4096 x 4096 windows.
    if (win < 1 || max_disp < 1) return SADBM_E_PARAM;
    if ((flags & SADBM_RAW_LOC) && max_disp > 256) return SADBM_E_PARAM;
    if (win > 4096 && rows > win && cols > win) return SADBM_E_PARAM;
    return SADBM_OK;
}

int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// Tuning / diagnostics knobs: the environment is read ONCE, at the first call; sadbm_reload_env()
// re-reads it (tests and tools that flip a knob between calls).
struct EnvKnobs {
    int strip_ty = 0;            // SADBM_STRIP_TY   rows per block of the strip matcher (0 = model)
    int strip_fg = 0;            // SADBM_STRIP_FG   frames per plane group (0 = choose, 1 = off, n = at most n)
    int matcher = 0;             // SADBM_MATCHER    0 default, 1 strip, 2 quad
    int prep_bx = 0;             // SADBM_PREP_BX    block width of prep_plane_kernel (0 = choose)
    int kc = 0;                  // SADBM_KC         disparities per thread of the quad matcher (0 = choose)
    int quad_half = -1;          // SADBM_QUAD_HALF  half-width (128-thread) blocks of the quad matcher: -1 choose, 0 never, 1 always
    int otf = -1;                // SADBM_OTF        on-the-fly form of the strip matcher (strip_otf_kernels.hip): -1 choose, 0 never, 1 whenever D <= 128
    int otf_ng = 0;              // SADBM_OTF_NG     its disparity groups per block: 0 choose (fewest that cover D), 1, 2 or 4
    int tail = -1;               // SADBM_TAIL       on-the-fly kernels: units per XCD split into half-height blocks at the end of the launch (-1 choose, 0 none)
    int trace = 0;               // SADBM_TRACE      1: one line per matcher launch on stderr (kernel, grid shape)
    int direct = -1;             // SADBM_DIRECT     on-the-fly form reading the caller's 1-channel images itself (no prep pass): -1 choose, 0 never, 1 whenever possible
    int wide_from = QUAD_MAXWIN + 1;   // SADBM_WIDE_FROM  narrowest window of sad_wide_kernel (26 .. 50; the quad matcher serves the ones below)
    long zc_bytes = 1 << 20;     // SADBM_ZEROCOPY_BYTES  host API: inputs up to this size are read by the GPU from pinned host memory
};
EnvKnobs read_env()
{
    EnvKnobs k;
    auto geti = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    k.strip_ty = geti("SADBM_STRIP_TY", 0); if (k.strip_ty < 1) k.strip_ty = 0;
    k.strip_fg = geti("SADBM_STRIP_FG", 0); if (k.strip_fg < 1) k.strip_fg = 0;
    if (const char *e = getenv("SADBM_MATCHER")) k.matcher = strcmp(e, "strip") == 0 ? 1 : strcmp(e, "quad") == 0 ? 2 : 0;
    k.prep_bx = geti("SADBM_PREP_BX", 0); if (k.prep_bx < 64 || k.prep_bx > 1024) k.prep_bx = 0; else k.prep_bx &= ~63;
    k.kc = geti("SADBM_KC", 0); if (k.kc != 8 && k.kc != 16) k.kc = 0;
    k.quad_half = geti("SADBM_QUAD_HALF", -1); if (k.quad_half < 0 || k.quad_half > 1) k.quad_half = -1;
    k.otf = geti("SADBM_OTF", -1); if (k.otf < 0 || k.otf > 1) k.otf = -1;
    k.tail = geti("SADBM_TAIL", -1); if (k.tail < -1) k.tail = -1;
    k.trace = geti("SADBM_TRACE", 0);
    k.direct = geti("SADBM_DIRECT", -1); if (k.direct < 0 || k.direct > 1) k.direct = -1;
    k.otf_ng = geti("SADBM_OTF_NG", 0); if (k.otf_ng != 1 && k.otf_ng != 2 && k.otf_ng != 4 && k.otf_ng != 8) k.otf_ng = 0;
    k.wide_from = geti("SADBM_WIDE_FROM", QUAD_MAXWIN + 1); if (k.wide_from <= MAXWIN || k.wide_from > QUAD_MAXWIN + 1) k.wide_from = QUAD_MAXWIN + 1;
    if (const char *e = getenv("SADBM_ZEROCOPY_BYTES")) k.zc_bytes = atol(e);
    return k;
}
EnvKnobs g_env_store;
std::once_flag g_env_once;
const EnvKnobs &env() { std::call_once(g_env_once, [] { g_env_store = read_env(); }); return g_env_store; }

int strip_band_rows(int bx, int n_pairs, int nvalid, int win, int NG, int n_cus)
{
    if (env().strip_ty) return env().strip_ty;
    const long cus = n_cus, per_cu = 16 / NG, slots = cus * per_cu;   // 16 waves per CU at 128 VGPRs; a block is NG waves
    const long cols = (long)bx * n_pairs;
    double best = 1e30; int best_ty = nvalid < 128 ? (nvalid > 0 ? nvalid : 1) : 128;
    auto consider = [&](int ty) {
        const long rnby = (nvalid + ty - 1) / ty, blocks = cols * rnby;
        const long full = blocks / slots, rem = blocks - full * slots;
        double rounds = (double)full;
        if (rem > 0) rounds += (per_cu >= 2 && rem <= slots / 2) ? 0.65 : 1.0;
        const double cost = (rounds + 0.4) * (ty + win - 1);   // +0.4: finer grids balance better (measured with SADBM_STRIP_TY sweeps, profiles/r01*)
        if (cost < best) { best = cost; best_ty = ty; }
    };
    // short bands first: a job that cannot fill one round of slots has no "whole rounds" candidate below, and fewer,
    // longer blocks then beat one row per block (profiles/r02b_small_frames_rows_per_block.txt)
    for (int ty = 1; ty <= 64 && ty <= nvalid; ++ty) consider(ty);
    for (long m = 1; m <= 24; ++m) {
        long nby = slots * m / cols;
        if (nby < 1) continue;
        if (nby > nvalid) nby = nvalid;
        const int ty = (int)((nvalid + nby - 1) / nby);
        if (ty > 512) continue;
        consider(ty);
    }
    return best_ty;
}

// Rows per block of the on-the-fly kernels (64-256 threads, 4-16 blocks per CU).  Fitted to SADBM_STRIP_TY sweeps on MI355X
// (tools/ty_otf.py, profiles/r02e_otf_rows_per_block.txt): with that many small blocks per CU the rounds of block slots blur
// into a stream -- a job of R rounds of block slots takes about R + 0.4 block durations, whether R is 0.7 or 16 (a little
// less when the blocks just fill a round; never less than ~0.6: with few blocks a step costs its latency) -- and a block
// lasts ty output steps, win-1 warm-up steps at about half price (no SUB pass, no key phase) and one step's worth of set-up.
int otf_band_rows(int bx, int n_groups, int nvalid, int win, int NG, int n_cus)
{
    if (env().strip_ty) return env().strip_ty;
    const double slots = (double)n_cus * (16 / NG);
    const long cols = (long)bx * n_groups;
    double best = 1e30; int best_ty = 1;
    for (int ty = 1; ty <= 256 && ty <= nvalid; ++ty) {
        const long nby = (nvalid + ty - 1) / ty;
        if (ty > 1 && nby == (nvalid + ty - 2) / (ty - 1)) continue;     // same band count as ty - 1: the shorter bands win
        const double R = (double)(cols * nby) / slots;
        double f = R + 0.4;
        if (R < 4.0 && R - (double)(long)R >= 0.96) f -= 0.2;            // the blocks just fill their last round
        if (f < 0.6) f = 0.6;                                              // too few blocks: a step costs its latency
        const double cost = f * (ty + 0.5 * (win - 1) + 1.0);
        if (cost < best) { best = cost; best_ty = ty; }
    }
    // the last band is shorter: even the rows out over the bands
    const long nby = (nvalid + best_ty - 1) / best_ty;
    return (int)((nvalid + nby - 1) / nby);
}

// Frames per plane group for the strip matcher: fewest 256-column blocks over the batch, groups at most 16K columns
// wide; one frame per plane unless grouping saves at least 1.5 % of the blocks.  Wp = frame pitch in columns (x16).
// direct: for the direct-reading kernels, which tell a staging unit's frame from its last column: Wp is a multiple of 16 for
// single frames too.  *nblocks = 256-column blocks of the batch.
void strip_frame_groups(int cols, int n_pairs, bool direct, int *FG, int *Wp, long *nblocks)
{
    const int wp = (cols + 15) & ~15;
    *FG = 1; *Wp = direct ? wp : cols;
    *nblocks = (long)n_pairs * ((cols + 255) / 256);
    const int efg = env().strip_fg;
    if (n_pairs < 2 || efg == 1) return;
    auto width = [&](int nf) { return ((long)(nf - 1) * wp + cols + 255) / 256; };   // column blocks of a plane holding nf frames
    auto blocks = [&](int fg) { return (long)(n_pairs / fg) * width(fg) + (n_pairs % fg ? width(n_pairs % fg) : 0); };
    long best = blocks(1); int best_fg = 1;
    const int fg_max = efg > 1 ? efg : 16384 / wp;
    const int fg_cap = (65535 - cols) / wp + 1;               // plane columns stay below 65536 (prep_plane_kernel's division)
    for (int fg = 2; fg <= fg_max && fg <= n_pairs && fg <= 1024 && fg <= fg_cap; ++fg) {
        const long b = blocks(fg);
        if (b < best) { best = b; best_fg = fg; }
    }
    if (best_fg > 1 && (double)best <= 0.985 * (double)blocks(1)) { *FG = best_fg; *Wp = wp; *nblocks = best; }
}

// Default matcher choice, from tools/sweep.py on MI355X (profiles/r01h_sweep_*.txt, profiles/r01k_sweep_*.txt): the strip
// matcher's blocks cover 64 disparities at least, so up to 32 disparities the quad matcher wins (strip 1.05-1.46x its
// time).  From 33 on the strip matcher wins: batches 0.68-0.88x (D = 40..64) and 0.68-0.80x (D >= 100), single small
// frames 0.43-0.97x; the worst cases measured are one 1080p frame at D = 40 (1.00-1.12x) and one 720p frame with
// win >= 21 (1.01-1.10x).
// Small jobs are latency bound and the strip matcher's blocks finish sooner: up to ~0.9 Mpixel per call it also wins at
// D <= 32 (one 384x288 frame, win 7, D 30 -- the reference's own configuration: 19 us vs 35 us; 8 such frames 0.80x; one
// 1280x720 frame 0.87-1.2x).
bool strip_is_default(int win, int D, int cols, int rows, int n_pairs)
{
    if (win < 4) return false;
    if (D > 32) return true;
    // (tools/small_jobs.py, profiles/r02b_small_jobs_strip_vs_quad.txt: crossover 0.9-1.2 Mpixel at D = 30/32; at D <= 16 the
    // quad matcher's blocks span 512 columns and a small job has too few of them: crossover ~2.5 Mpixel)
    return (long)n_pairs * rows * cols <= (D <= 16 ? 2500000L : 900000L);
}

// The on-the-fly form of the strip matcher against the other two, from tools/otf_check.py sweep on MI355X
// (profiles/r02e_otf_sweep*.txt: D = 4..64, windows 4..25, batches and single frames from 384x288 to 1080p): it is the
// fastest in 580 of 640 cases, by 10-45 %; the exceptions are 4-wide windows from 17 disparities up (quad matcher 2-16 %
// faster) and a few single 1080p frames at D = 33..40 (quad matcher 0-7 % faster), which keep the old choice / are let go.
bool otf_is_default(int win, int D, int cols, int rows, int n_pairs)
{
    (void)cols; (void)rows; (void)n_pairs;
    // 65-128 disparities, not a multiple of 16, windows whose keys have room for poisoned sums (SG::POISON): 512-thread blocks
    // of the on-the-fly form against the staged kernel's masked argmin, 32 x 1080p at 7x7 / D = 100: 1.73 against 1.89 ms
    if (D > 64) return D <= 128 && D % 16 != 0 && win >= 5 && win <= 11 && win != 10;
    return win >= 5 || D <= 16;
}

// What run_pipeline decides before it touches the GPU: which matcher family, and for the strip family the kernel form, the
// frame groups, whether the inputs are read in place, the scratch planes and the grid.  No HIP calls: sadbm_plan_describe()
// exposes it, and the CPU-only tests check the policy and its invariants through that.
struct StripPlan {
    bool use_strip = false, use_otf = false, direct = false, sdirect = false, strip_zeroes = false;
    int ng = 0;                           // groups of 16 disparities per block (block = 64 ng threads)
    int FG = 1, Wp = 0, n_groups = 0, vcols = 0, pitch = 0;
    size_t plane_bytes = 0, planes_need = 0;
    int bx = 0, TY = 0, nby = 0, tail = 0;
    long nblocks = 0;
    size_t lds = 0;
    strip_kernel_t kernel = nullptr;
};
constexpr int SDIRECT_NEDGE = 2;
StripPlan plan_strip(int cus, bool host_inputs, int64_t pitchL, int64_t pitchR, int n_pairs, int rows_out, int nvalid, int cols,
                     int channels, int win, int D, uint32_t flags, bool sdirect_ok = true)
{
    StripPlan sp;
    const int need_rows = nvalid + win - 1;
    // matcher choice: SADBM_KERNEL_STRIP / SADBM_KERNEL_QUAD force one; env SADBM_MATCHER=strip|quad overrides the default
    bool use_strip = false;
    int strip_ng = D <= 64 ? 4 : D <= 128 ? 8 : 16;          // disparity groups of 16 per block (multiple of 4)
    strip_kernel_t sk = (win <= MAXWIN && cols < 65000 && D <= 256) ? pick_strip_kernel(win, strip::NS + strip_ng) : nullptr;
    // short searches: the on-the-fly form (blocks of 1, 2 or 4 groups of 16 disparities, strip_otf_kernels.hip)
    int otf_lds = 0;
    int otf_ng = D <= 16 ? 1 : D <= 32 ? 2 : D <= 64 ? 4 : 8;
    if (env().otf_ng > otf_ng) otf_ng = env().otf_ng;
    strip_kernel_t sk_otf = (D <= 128 && win <= MAXWIN && cols < 65000 && env().otf != 0) ? pick_strip_otf_kernel(win, otf_ng, false, &otf_lds) : nullptr;
    const bool use_otf = sk_otf != nullptr && (env().otf == 1 || otf_is_default(win, D, cols, nvalid, n_pairs));
    if (use_otf) { sk = sk_otf; strip_ng = otf_ng; }
    if (sk && !(flags & (SADBM_FORCE_GENERIC | SADBM_KERNEL_GLOBAL | SADBM_KERNEL_QUAD))) {
        use_strip = (flags & SADBM_KERNEL_STRIP) != 0 || use_otf || strip_is_default(win, D, cols, nvalid, n_pairs);
        if (env().matcher && !(flags & SADBM_KERNEL_STRIP)) use_strip = env().matcher == 1;
    }

    // Scratch planes.  The quad / generic matchers get one padded plane pair per frame.  The strip matcher works on
    // 256-column blocks, so frames whose width is not a multiple of 256 would leave the last block of every row part
    // empty (1920: 6 %, 640: 19 %): it gets FG frames side by side in one plane, Wp columns apart, and treats the group
    // as one wide frame whose strips know their frame (sad_strip.inc).
    int FG = 1, Wp = cols;
    long blocksg = 0;
    if (use_strip) strip_frame_groups(cols, n_pairs, false, &FG, &Wp, &blocksg);
    // The on-the-fly form reads single-channel inputs itself: no prep pass (7-8 % of a step at D <= 32, 4 % at 64, and one
    // launch less per call), with the same frame groups (tools/fg_stride.py, profiles/r02f_otf_direct_frame_groups.txt: 64 x
    // 1080p at 7x7 / D = 30: 0.98 ms against 0.955 + 0.085).  Not for inputs the GPU reads over PCIe (the host API's zero-copy
    // staging): the matcher reads every byte ~5 times.
    // (64-thread blocks with windows from 13 columns up: two staging units per lane, the direct variant spills 8-39 VGPRs.
    // Until late in round 3 large jobs of 256-thread blocks with windows up to 13 kept the planes too: that variant stages the E words of
    // L, which won back a little more than the prep pass cost.  Since the in-place masks of the on-the-fly kernels the direct variant is
    // as fast or faster on every such job -- 64 x 1080p at 9x9 / D = 64: 1.738 ms against 1.747 with the pass, 13x13: 1.90 against 2.01,
    // tools/step_direct_ab.py -- without scratch planes and with half the traffic.)
    // Round 3: interleaved BGR frames -- what cv::imread hands the reference (SAD.cpp:65-66) -- are read in place too where the variant
    // that gathers channel 0 itself is built (17-32 disparities, windows up to 8: strip_otf_kernels.hip): the reference's own call,
    // one BGR pair at winSize 7 / DSR 30, is then one launch.
    bool direct = use_strip && use_otf && !host_inputs && env().direct != 0 && cols + 32 < 65000 &&
                  (channels == 1 ? !(otf_ng == 1 && win >= 13 && env().direct < 0)
                                 : pick_strip_otf_kernel(win, otf_ng, true, nullptr, channels) != nullptr) &&
                  (int64_t)need_rows * (pitchL > pitchR ? pitchL : pitchR) < 0x7ff00000LL;
    if (direct) {
        strip_frame_groups(cols, n_pairs, true, &FG, &Wp, &blocksg);
        sk = pick_strip_otf_kernel(win, otf_ng, true, &otf_lds, channels);
    }
    // The staged kernel reads single-channel device images in place as well (round 3: no prep pass, no scratch planes; step traffic
    // 2.1-2.5x -> 1.1x the algorithmic 3 W H): its staging offsets are 32-bit from one base, so both batches must share a row
    // pitch and lie within 4 GB of each other (sdirect_ok: run_pipeline checks the pointers).
    // Rows that a 16-byte unit cannot load without leaving its batch are assembled byte-wise into SDIRECT_NEDGE spare row images in
    // LDS: with a pitch of 384 bytes and more these are at most the first row of the batch (look-back <= 275 bytes left of column
    // 0) and its last one (windows reach <= 330 bytes past the row's end).  256-thread blocks (D <= 64 under SADBM_OTF=0, 4-wide
    // windows) keep the planes: four of them share a CU's LDS, the spare row images would cost one.
    const bool sdirect = use_strip && !use_otf && channels == 1 && !host_inputs && env().direct != 0 && sdirect_ok && pitchL == pitchR &&
                         pitchL >= 384 && pitchL < 0x7ff00000LL && cols + 32 < 65000 && strip_ng >= 8;
    if (sdirect) strip_frame_groups(cols, n_pairs, true, &FG, &Wp, &blocksg);
    if (use_strip && !direct && !sdirect && ((size_t)(PADL + (size_t)(FG - 1) * Wp + cols + PADR + 127) & ~(size_t)127) * (size_t)need_rows > 0x7ff00000u) {
        FG = 1; Wp = cols;                                    // the strip matcher addresses the R plane with 32-bit offsets
        if ((size_t)((PADL + cols + PADR + 127) & ~127) * (size_t)need_rows > 0x7ff00000u) use_strip = false;
    }
    const bool strip_zeroes = use_strip && rows_out - nvalid <= 64;
    const int n_groups = (n_pairs + FG - 1) / FG;
    const int vcols = (FG - 1) * Wp + cols;                    // columns of a group's plane that hold frames
    const int pitch = (PADL + vcols + PADR + 127) & ~127;
    const size_t plane_bytes = (size_t)pitch * (size_t)need_rows;
    // scratch layout: [group][L plane | R plane], + slack: the strip matcher stages whole 256-column blocks
    // and may read a few hundred bytes past the last row (values only reach columns >= cols-win, which are 0)
    const size_t planes_need = 2 * plane_bytes * (size_t)n_groups + 4096;
    sp.use_strip = use_strip; sp.use_otf = use_strip && use_otf; sp.direct = direct; sp.sdirect = use_strip && sdirect; sp.strip_zeroes = strip_zeroes;
    sp.ng = strip_ng; sp.FG = FG; sp.Wp = Wp; sp.n_groups = n_groups; sp.vcols = vcols; sp.pitch = pitch;
    sp.plane_bytes = plane_bytes; sp.planes_need = planes_need; sp.kernel = sk;
    if (!use_strip) return sp;

    const int bx = (vcols + strip::XBW - 1) / strip::XBW;
    // rows per block: a block is 64*NG threads and two blocks fit a CU only up to NG = 8, so the grid is a few
    // hundred large blocks and its quantisation matters.  Minimise rounds x steps-per-block over band heights
    // that fill whole rounds of block slots (a round whose CUs hold a single block runs ~0.65 as long).
    const int TY = (use_otf && strip_ng <= 4) ? otf_band_rows(bx, n_groups, nvalid, win, strip_ng, cus) : strip_band_rows(bx, n_groups, nvalid, win, strip_ng, cus);
    int PR = 0, ROWW = 0;
    if (!use_otf) strip_geometry(win, strip::NS + strip_ng, &PR, &ROWW);
    sp.lds = use_otf ? (size_t)otf_lds : (size_t)(4 * ROWW + 3 * 64 * PR + 2 * 4 * 64 + 64 + 8 + (sdirect ? SDIRECT_NEDGE * ROWW : 0)) * 4;
    sp.bx = bx; sp.TY = TY; sp.nby = (nvalid + TY - 1) / TY;
    sp.nblocks = (long)sp.bx * sp.nby * n_groups;
    const long chunk_units = (sp.nblocks + 7) / 8;            // per XCD
    if (use_otf && env().tail != 0 && TY >= (env().tail > 0 ? 1 : 8)) {
        // on-the-fly kernels: the last round of block slots of every XCD runs as half-height blocks: 64 x 1080p at 7x7 / D = 30
        // -3.7 %, 32 x 4K -1.5 %, config 2 -1 % (SADBM_TAIL sweep: one round is the best size); jobs of fewer than three rounds
        // gain nothing or lose 1-2 % to the extra warm-up rows, they keep whole blocks
        const long slots_xcd = (long)(cus / 8 > 0 ? cus / 8 : 1) * (16 / strip_ng);
        long t = env().tail > 0 ? env().tail : slots_xcd;
        if (t > chunk_units / 2) t = chunk_units / 2;
        if (env().tail < 0 && chunk_units < 3 * slots_xcd) t = 0;
        sp.tail = (int)t;
    }
    return sp;
}

// Kernels that need more than 64 KB of dynamic LDS are told so ONCE per (device, kernel, size), not on every launch
// (hipFuncSetAttribute costs a few microseconds of a 30 us call).
int allow_large_lds(int device, const void *kernel, size_t lds)
{
    struct Seen { int device; const void *kernel; size_t lds; };
    static std::mutex mu;
    static std::vector<Seen> seen;
    std::lock_guard<std::mutex> guard(mu);
    for (const Seen &e : seen)
        if (e.device == device && e.kernel == kernel && e.lds >= lds) return SADBM_OK;
    HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (Seen &e : seen)
        if (e.device == device && e.kernel == kernel) { e.lds = lds; return SADBM_OK; }
    seen.push_back({device, kernel, lds});
    return SADBM_OK;
}

// The whole device pipeline for n_pairs frames (or one band).  Input rows available: rows_in; output
// rows written: rows_out; output rows [0,nvalid) are computed, the rest zero-filled.
int run_pipeline(sadbm_ctx *ctx, const uint8_t *dL, int64_t pitchL, const uint8_t *dR, int64_t pitchR,
                 int64_t pair_stride_in, int n_pairs, int rows_in, int rows_out, int nvalid, int cols, int channels,
                 int win, int D, uint32_t flags, uint8_t *dOut, int64_t out_pitch, int64_t pair_stride_out,
                 hipStream_t st)
{
    if (n_pairs <= 0 || rows_out <= 0 || cols <= 0) return SADBM_OK;
    DeviceScope dscope(ctx->device);
    HIPCHK(dscope.err);
    std::lock_guard<std::recursive_mutex> guard(ctx->mu);
    // The scratch planes are rewritten by every call: work queued on another stream by the previous call must be done
    // before this call's kernels touch them.  (If that stream has been destroyed meanwhile, wait for the whole device.)
    if (ctx->has_last && ctx->last_stream != st) {
        hipError_t e = ctx->xev ? hipEventRecord(ctx->xev, ctx->last_stream) : hipErrorInvalidHandle;
        if (e == hipSuccess) e = hipStreamWaitEvent(st, ctx->xev, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); HIPCHK(hipDeviceSynchronize()); }
    }
    ctx->last_stream = st; ctx->has_last = true;
    if (nvalid < 0) nvalid = 0;
    if (cols - win <= 0) nvalid = 0;
    // disparities k >= cols never compete (k > column, SAD.cpp:44) and the zero-cost slot that can win is k = column + 1 < cols:
    // searching [0, cols) gives what [0, DSR) gives
    if (D > cols) D = cols;
    // zero frame: rows [nvalid, rows_out) entirely (SAD.cpp:31,34); columns >= cols-win are written as zeros by the
    // matchers themselves.  The strip matcher also writes these rows (blocks of its last band), unless there are many.
    auto zero_rows = [&]() -> int {
        for (int z0 = nvalid; z0 < rows_out; z0 += 65535) {   // grid.y <= 65535
            const int zn = rows_out - z0 < 65535 ? rows_out - z0 : 65535;
            dim3 zg((cols + 1023) / 1024, zn, n_pairs);
            hipLaunchKernelGGL(zero_rows_kernel, zg, dim3(256), 0, st, dOut, out_pitch, pair_stride_out, z0, zn, cols);
        }
        HIPCHK(hipGetLastError());
        return SADBM_OK;
    };
    if (nvalid == 0) { snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "none"); return zero_rows(); }

    const int need_rows = nvalid + win - 1;                   // <= rows_in (checked by callers)

    // matcher choice, frame groups, direct reads, grid shape of the strip family: plan_strip (no HIP calls in it)
    // direct form of the staged kernel: 32-bit staging offsets from SDIRECT_BIAS below the lower batch must reach every byte of both
    const uint8_t *sd_base = dL < dR ? dL : dR;
    const int64_t sd_ext_l = (int64_t)(n_pairs - 1) * pair_stride_in + (int64_t)(need_rows - 1) * pitchL + cols;
    const int64_t sd_ext_r = (int64_t)(n_pairs - 1) * pair_stride_in + (int64_t)(need_rows - 1) * pitchR + cols;
    const int64_t sd_ll = (int64_t)(dL - sd_base) + strip::SDIRECT_BIAS, sd_lr = (int64_t)(dR - sd_base) + strip::SDIRECT_BIAS;
    const bool sdirect_ok = pair_stride_in >= 0 && sd_ll + sd_ext_l + 65536 < 0xF0000000LL && sd_lr + sd_ext_r + 65536 < 0xF0000000LL;
    const StripPlan sp = plan_strip(ctx->cus, ctx->host_inputs, pitchL, pitchR, n_pairs, rows_out, nvalid, cols, channels, win, D, flags, sdirect_ok);
    const bool use_strip = sp.use_strip, use_otf = sp.use_otf, direct = sp.direct || sp.sdirect, strip_zeroes = sp.strip_zeroes;
    const int FG = sp.FG, Wp = sp.Wp, n_groups = sp.n_groups, pitch = sp.pitch;
    const size_t plane_bytes = sp.plane_bytes, planes_need = sp.planes_need;
    const strip_kernel_t sk = sp.kernel;
    // static scratch is never grown: refuse before anything is launched (no half-written output, nothing queued on the caller's
    // stream, no half-built graph under capture); launches that read the inputs in place need no scratch and are never refused
    if (ctx->planes_fixed && !direct && planes_need > ctx->planes_cap) return SADBM_E_LIMIT;
    if (nvalid < rows_out && !strip_zeroes) { int zrc = zero_rows(); if (zrc) return zrc; }
    int rc = (ctx->planes_fixed || direct) ? SADBM_OK : grow(&ctx->planes, &ctx->planes_cap, planes_need);
    if (rc) return rc;
    uint8_t *pL = ctx->planes, *pR = ctx->planes + plane_bytes;

    const bool prof = ctx->prof_on && ctx->ev_n < ctx->ev_cap;
    hipEvent_t *ev = prof ? ctx->ev + 3 * ctx->ev_n : nullptr;
    if (prof) ctx->ev_noprep[ctx->ev_n] = direct ? 1 : 0;
    if (prof && !direct) HIPCHK(hipEventRecord(ev[0], st));   // two events back to back would read as ~5 us of "prep"
    const uint32_t Wp_magic = (uint32_t)(0x100000000ull / (uint64_t)(Wp > 0 ? Wp : 1)) + 1u;
    // block width: the multiple of 64 threads that wastes the fewest lanes over a row of pitch/16 chunks (4K: 264 chunks -> 320)
    const int cpr = pitch / 16;
    int pbx = 256; long pwaste = 1L << 60;
    for (int b = 128; b <= 512; b += 64) {                     // measured (SADBM_PREP_BX sweep): 64-thread blocks are slow, ties -> larger
        const long w = (long)((cpr + b - 1) / b) * b - cpr;
        if (w <= pwaste) { pwaste = w; pbx = b; }
    }
    if (env().prep_bx) pbx = env().prep_bx;
    dim3 pg((cpr + pbx - 1) / pbx, (need_rows + PREP_ROWS - 1) / PREP_ROWS, 2 * n_groups);   // n_groups <= 32767 (callers chunk)
    if (!direct) {
        hipLaunchKernelGGL(prep_plane_kernel, pg, dim3(pbx), 0, st, dL, pitchL, dR, pitchR, pair_stride_in, channels, need_rows, cols, pL, pitch,
                           (int64_t)plane_bytes, FG, Wp, Wp_magic, n_pairs);
        HIPCHK(hipGetLastError());
    }

    Params p;
    p.L = pL + PADL; p.R = pR + PADL; p.plane_stride = (int64_t)(2 * plane_bytes); p.pitch = pitch;
    p.W = cols; p.nvalid = nvalid; p.win = win; p.D = D;
    p.out = dOut; p.out_pitch = out_pitch; p.out_stride = pair_stride_out; p.flags = flags;
    p.dword_store = ((((uintptr_t)dOut) | (uintptr_t)out_pitch | (uintptr_t)pair_stride_out) & 3) == 0;
    p.NQ = 0; p.NCp = 0; p.TY = 0; p.LW = 0; p.RW = 0; p.nbx = p.nby = p.npairs = 0;
    if (prof) HIPCHK(hipEventRecord(ev[1], st));

    if (use_strip) {
        strip::SParams q;
        q.NG = sp.ng;
        q.Lrow0 = pL; q.pair_stride = (int64_t)(2 * plane_bytes); q.lr_delta = (uint32_t)plane_bytes;
        q.pitch = pitch; q.W = cols; q.nvalid = nvalid; q.win = win; q.D = D;
        q.out = dOut; q.out_pitch = out_pitch; q.out_stride = pair_stride_out; q.flags = flags; q.dword_store = p.dword_store;
        q.dL = dL; q.dR = dR; q.in_stride = pair_stride_in; q.pitchL = (int)pitchL; q.pitchR = (int)pitchR;
        q.extL = sp.direct ? (int)((int64_t)(need_rows - 1) * pitchL + (int64_t)cols * channels) : 0;
        q.extR = sp.direct ? (int)((int64_t)(need_rows - 1) * pitchR + (int64_t)cols * channels) : 0;
        q.sdirect = sp.sdirect ? 1 : 0; q.nedge = sp.sdirect ? SDIRECT_NEDGE : 0; q.ll_delta = 0; q.loA = q.lenA = q.loB = q.lenB = 0;
        if (sp.sdirect) {
            q.Lrow0 = sd_base - strip::SDIRECT_BIAS;          // an address, never dereferenced below sd_base
            q.pair_stride = (int64_t)FG * pair_stride_in; q.pitch = (int)pitchL;
            q.ll_delta = (uint32_t)sd_ll; q.lr_delta = (uint32_t)sd_lr;
            q.loA = sd_ll; q.lenA = sd_ext_l; q.loB = sd_lr; q.lenB = sd_ext_r;
        }
        const int TY = sp.TY;
        const size_t lds = sp.lds;
        q.TY = TY;
        if (lds > 64 * 1024) { int arc = allow_large_lds(ctx->device, (const void *)sk, lds); if (arc) return arc; }
        q.nbx = sp.bx; q.nby = sp.nby; q.ngroups = n_groups;
        q.FG = FG; q.Wp = Wp; q.npairs = n_pairs;
        q.rows_out = strip_zeroes ? rows_out : nvalid;
        if (sp.nblocks > 0x3ffffff0L) return SADBM_E_SIZE;
        q.tail = sp.tail;
        const long chunk_units = (sp.nblocks + 7) / 8;         // per XCD
        dim3 g((unsigned)(8 * (chunk_units + q.tail)));         // 1-D, XCD-aware order decoded in the kernel
        if (env().trace)
            fprintf(stderr, "sadbm: strip%s win %d D %d NG %d direct %d  %dx%d x%d  FG %d Wp %d  bx %d nby %d groups %d TY %d  tail %d  lds %zu\n", use_otf ? "_otf" : "",
                    win, D, q.NG, (int)direct, cols, nvalid, n_pairs, FG, Wp, q.nbx, q.nby, q.ngroups, TY, q.tail, lds);
        hipLaunchKernelGGL(sk, g, dim3(64 * q.NG), lds, st, q);
        snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), use_otf ? "sad_strip_otf_kernel<%d>" : "sad_strip_kernel<%d>", win);
        if (use_otf) snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "sad_strip_otf_kernel<%d,%d>%s", win, sp.ng, direct ? "+direct" : "");
        else if (sp.sdirect) snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "sad_strip_kernel<%d>+direct", win);
        HIPCHK(hipGetLastError());
        if (prof) { HIPCHK(hipEventRecord(ev[2], st)); ctx->ev_n++; }
        return SADBM_OK;
    }

    // beyond 256 disparities (reference encoding only: (loc*16) mod 256, SAD.cpp:57) the column-sum matcher serves every window
    // up to 255 columns: its 64-disparity passes merge by cost, so the key only needs k mod 256
    if ((win >= env().wide_from || D > 256) && win <= 255 && !(flags & SADBM_FORCE_GENERIC) && !((flags & SADBM_KERNEL_GLOBAL) && D <= 256)) {
        const int NX = 257 - win;                                // output columns per block (win <= 255)
        // rows per block: every block sums win - 1 rows before its first output row, but a grid below a few blocks per CU
        // leaves the loads' and the exchanges' latency exposed (profiles/r02l_wide_windows.txt)
        int TY = WIDE_TY;                                        // the tallest of 32, 16, 8 that still gives 1.5 blocks per CU
        while (TY > 8 && (long)((cols + NX - 1) / NX) * ((nvalid + TY - 1) / TY) * n_pairs < 3L * ctx->cus / 2) TY >>= 1;
        if (env().strip_ty) TY = env().strip_ty < WIDE_TY ? env().strip_ty : WIDE_TY;
        p.TY = TY;
        dim3 g((cols + NX - 1) / NX, (nvalid + TY - 1) / TY, n_pairs);
        if (g.y > 65535u || g.z > 65535u) return SADBM_E_SIZE;
        hipLaunchKernelGGL(sad_wide_kernel, g, dim3(256), (size_t)WIDE_LDS + (D > 64 ? (size_t)WIDE_TY * 256 * 4 : 0), st, p);
        snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "sad_wide_kernel");
    } else if (win > 255 || (win > MAXWIN && (flags & SADBM_KERNEL_GLOBAL)) || (flags & SADBM_FORCE_GENERIC)) {
        dim3 g((cols + 255) / 256, nvalid, n_pairs);
        hipLaunchKernelGGL(sad_generic_kernel, g, dim3(256), 0, st, p);
        snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "sad_generic_kernel");
    } else {
        const bool use_global = (flags & SADBM_KERNEL_GLOBAL) != 0;
        // disparities per thread: 16 keeps the per-row L work amortised for wide searches; 8 gives twice
        // the blocks and 5 waves/SIMD for D <= 64 (measured: cfg2 0.32 -> 0.25 ms, cfg3 0.78 vs 0.85 ms)
        int KC = use_global ? KCG : (D <= 64 ? 8 : 16);
        if (!use_global && env().kc) KC = env().kc;
        const int NC = (D + KC - 1) / KC;                     // <= 256/KC
        p.NCp = pow2ceil(NC);
        const int quads = (cols + 3) / 4;
        p.NQ = 256 / p.NCp;
        if (p.NQ > 16 && p.NQ > pow2ceil(quads)) p.NQ = pow2ceil(quads) < 16 ? 16 : pow2ceil(quads);
        if (!use_global && p.NQ * p.NCp == 256 && p.NQ >= 32) {
            // half-width blocks (128 threads) when the frame's column quads fill them better: 384 columns = 96
            // quads = 3 x 32 exactly but 2 x 64 with a third of the threads idle; 1920 = 15 x 32 against 8 x 64.
            const int h = p.NQ / 2;
            const long full = (long)((quads + p.NQ - 1) / p.NQ) * p.NQ, half = (long)((quads + h - 1) / h) * h;
            bool use_half = half * 100 < full * 92;     // a half-width slot costs 7-11 % more (profiles/r02b_quad_half_width_blocks.txt)
            if (env().quad_half >= 0) use_half = env().quad_half != 0;
            if (use_half) p.NQ = h;
        }
        const int bx = (quads + p.NQ - 1) / p.NQ;
        // rows per block: long bands amortise the win-1 warm-up rows; shorten them until the grid
        // has a few blocks per CU.
        int TY = SADBM_TYMAX;
        const long minblocks = (long)SADBM_MINBLOCKS * 256 / (p.NQ * p.NCp);   // counted in 256-thread blocks
        while (TY > 16 && (long)bx * ((nvalid + TY - 1) / TY) * n_pairs < minblocks) TY >>= 1;
        p.TY = TY;
        dim3 g(bx, (nvalid + TY - 1) / TY, n_pairs);
        if (use_global) {
            quad_kernel_t k = pick_quad_kernel(win);
            hipLaunchKernelGGL(k, g, dim3(p.NQ * p.NCp), 0, st, p);
            snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "sad_quad_kernel<%d>", win);
        } else {
            const int NW = (win + 6) / 4;
            p.LW = p.NQ + NW;
            p.RW = p.NQ + NW + p.NCp * KC / 4 + 1;
            const size_t lds = (size_t)(2 * 256 * 4 + 4 * (p.LW + 4 * p.RW)) * 4;
            quad_kernel_t k = pick_quad_lds_kernel(win, KC);
            p.nbx = bx; p.nby = (nvalid + TY - 1) / TY; p.npairs = n_pairs;
            const long nblocks = (long)p.nbx * p.nby * p.npairs;
            if (nblocks > 0x7ffffff0L) return SADBM_E_SIZE;
            hipLaunchKernelGGL(k, dim3((unsigned)((nblocks + 7) & ~7L)), dim3(p.NQ * p.NCp), lds, st, p);
            snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "sad_quad_lds_kernel<%d,%d>", win, KC);
        }
    }
    HIPCHK(hipGetLastError());
    if (prof) { HIPCHK(hipEventRecord(ev[2], st)); ctx->ev_n++; }
    return SADBM_OK;
}

}  // namespace

namespace sadbm_detail {
namespace {
struct StreamPool {
    std::mutex mu;
    std::vector<std::vector<hipStream_t>> idle;               // per device
};
// never destroyed: at process exit the HIP runtime may already be gone
StreamPool &stream_pool() { static StreamPool *p = new StreamPool(); return *p; }
constexpr size_t STREAM_POOL_MAX = 256;                        // idle streams kept per device; beyond that they are destroyed after all
}  // namespace

hipError_t stream_acquire(int device, hipStream_t *out)
{
    StreamPool &sp = stream_pool();
    {
        std::lock_guard<std::mutex> guard(sp.mu);
        if (device >= 0 && (size_t)device < sp.idle.size() && !sp.idle[device].empty()) {
            *out = sp.idle[device].back();
            sp.idle[device].pop_back();
            return hipSuccess;
        }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void stream_release(int device, hipStream_t s)
{
    if (!s) return;
    StreamPool &sp = stream_pool();
    {
        std::lock_guard<std::mutex> guard(sp.mu);
        if (device >= 0 && device < 1024) {
            if ((size_t)device >= sp.idle.size()) sp.idle.resize((size_t)device + 1);
            if (sp.idle[device].size() < STREAM_POOL_MAX) { sp.idle[device].push_back(s); return; }
        }
    }
    (void)hipStreamDestroy(s);
}
}  // namespace sadbm_detail

namespace {

// The convenience call sadbm_get_disparity() keeps one context per host thread; it is released when the thread exits
// (stream, device scratch, pinned staging -- about 25 MB per 4K pair).  Main-thread thread_locals are destroyed before
// static-storage objects, i.e. while the HIP runtime is still alive.
struct TlsCtx {
    sadbm_ctx *c = nullptr;
    ~TlsCtx() { if (c) { sadbm_ctx *d = c; c = nullptr; sadbm_destroy(d); } }
};
thread_local TlsCtx g_tls;

}  // namespace

extern "C" {

int sadbm_version(void) { return SADBM_VERSION; }

const char *sadbm_strerror(int code)
{
    switch (code) {
    case SADBM_OK: return "ok";
    case SADBM_E_NULL: return "null pointer argument";
    case SADBM_E_TYPE: return "unsupported channel count or left/right channel mismatch";
    case SADBM_E_SIZE: return "bad image size or pitch (right must be at least as large as left)";
    case SADBM_E_PARAM: return "bad parameter (need win >= 1, max_disp >= 1; SADBM_RAW_LOC needs max_disp <= 256; windows beyond 4096 columns are not supported)";
    case SADBM_E_HIP: return "HIP runtime error";
    case SADBM_E_ALLOC: return "device memory allocation failed";
    case SADBM_E_BUSY: return "pipeline slot still in flight, or ticket unknown";
    case SADBM_E_LIMIT: return "input needs more scratch than the static buffer of sadbm_compute_device_static holds";
    case SADBM_E_RCCL: return "RCCL error (sadbm_last_hip_error() has the text), or librccl could not be loaded";
    default: return "unknown error";
    }
}

const char *sadbm_last_hip_error(void) { return g_hip_err; }
char *sadbm_internal_error_buffer(void) { return g_hip_err; }   // multi.hip writes its error text here (256 bytes)
// multi.hip's host-side row copies: the same pool-parallel copy / channel-0 gather the host API uses
void sadbm_internal_copy_rows(uint8_t *dst, int64_t dstep, const uint8_t *src, int64_t sstep, size_t cols, int rows, int ch)
{
    copy_rows(dst, dstep, src, sstep, cols, rows, ch);
}

int sadbm_create(int device, sadbm_ctx **out)
{
    if (!out) return SADBM_E_NULL;
    *out = nullptr;
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) { snprintf(g_hip_err, sizeof(g_hip_err), "device %d of %d", device, n); return SADBM_E_HIP; }
    DeviceScope dscope(device);
    HIPCHK(dscope.err);
    sadbm_ctx *c = new (std::nothrow) sadbm_ctx();
    if (!c) return SADBM_E_ALLOC;
    c->device = device;
    hipError_t e = stream_acquire(device, &c->stream);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->xev, hipEventDisableTiming);
    if (e != hipSuccess) { stream_release(device, c->stream); delete c; return hip_fail(e, "hipStreamCreate"); }
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c->cus = v; }
    *out = c;
    return SADBM_OK;
}

void sadbm_destroy(sadbm_ctx *c)
{
    if (!c) return;
    DeviceScope dscope(c->device);
    // the scratch may still be read by kernels the caller queued on its own stream: hipFree waits for the device
    if (c->stream) { (void)hipStreamSynchronize(c->stream); stream_release(c->device, c->stream); }
    if (c->planes && !c->planes_fixed) (void)hipFree(c->planes);
    if (c->stage_in) (void)hipFree(c->stage_in);
    if (c->stage_out) (void)hipFree(c->stage_out);
    if (c->taps) (void)hipFree(c->taps);
    if (c->pin_in) (void)hipHostFree(c->pin_in);
    if (c->pin_out) (void)hipHostFree(c->pin_out);
    if (c->xev) (void)hipEventDestroy(c->xev);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); stream_release(c->device, c->stream2); }
    if (c->stream3) { (void)hipStreamSynchronize(c->stream3); stream_release(c->device, c->stream3); }
    for (hipEvent_t e : c->chunk_ev) if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < 3 * c->ev_cap; ++i) (void)hipEventDestroy(c->ev[i]);
    delete[] c->ev;
    if (g_tls.c == c) g_tls.c = nullptr;
    delete c;
}

void sadbm_reload_env(void) { (void)env(); g_env_store = read_env(); }

int sadbm_ctx_taps(sadbm_ctx *ctx, double **taps, int count)
{
    if (!ctx || !taps) return SADBM_E_NULL;
    if (count < 1 || count > 4096) return SADBM_E_PARAM;
    DeviceScope dscope(ctx->device);
    HIPCHK(dscope.err);
    std::lock_guard<std::recursive_mutex> guard(ctx->mu);
    if (!ctx->taps) {
        hipError_t e = hipMalloc((void **)&ctx->taps, 4096 * sizeof(double));
        if (e != hipSuccess) { ctx->taps = nullptr; hip_fail(e, "hipMalloc"); return SADBM_E_ALLOC; }
    }
    *taps = ctx->taps;
    return SADBM_OK;
}

const char *sadbm_last_kernel_name(const sadbm_ctx *ctx) { return ctx ? ctx->last_kernel : ""; }

int sadbm_plan_describe(int rows, int cols, int channels, int n_pairs, int win, int max_disp, uint32_t flags, int cus, char *text, int text_len)
{
    if (!text || text_len <= 0) return SADBM_E_NULL;
    int rc = check_params(rows, cols, channels, win, max_disp, flags);
    if (rc) return rc;
    if (n_pairs < 1) return SADBM_E_SIZE;
    if (cus <= 0) cus = 256;
    int nvalid = rows - win;
    if (nvalid < 0 || cols - win <= 0) nvalid = 0;
    if (nvalid == 0 || rows == 0 || cols == 0) { snprintf(text, (size_t)text_len, "matcher=none"); return SADBM_OK; }
    if (max_disp > cols) max_disp = cols;                     // as run_pipeline
    const StripPlan sp = plan_strip(cus, false, (int64_t)cols * channels, (int64_t)cols * channels, n_pairs, rows, nvalid, cols, channels, win,
                                    max_disp, flags);
    if (sp.use_strip) {
        char name[64];
        if (sp.use_otf) snprintf(name, sizeof(name), "sad_strip_otf_kernel<%d,%d>%s", win, sp.ng, sp.direct ? "+direct" : "");
        else snprintf(name, sizeof(name), "sad_strip_kernel<%d>%s", win, sp.sdirect ? "+direct" : "");
        snprintf(text, (size_t)text_len, "matcher=%s kernel=%s threads=%d FG=%d Wp=%d groups=%d bx=%d TY=%d nby=%d blocks=%ld tail=%d lds=%zu scratch=%zu prep=%d",
                 sp.use_otf ? "strip_otf" : "strip", name, 64 * sp.ng, sp.FG, sp.Wp, sp.n_groups, sp.bx, sp.TY, sp.nby, sp.nblocks, sp.tail, sp.lds,
                 (sp.direct || sp.sdirect) ? (size_t)0 : sp.planes_need, (sp.direct || sp.sdirect) ? 0 : 1);
    } else if ((win >= env().wide_from || max_disp > 256) && win <= 255 && !(flags & SADBM_FORCE_GENERIC) && !((flags & SADBM_KERNEL_GLOBAL) && max_disp <= 256)) {
        snprintf(text, (size_t)text_len, "matcher=wide kernel=sad_wide_kernel scratch=%zu prep=1", sp.planes_need);
    } else if (win > 255 || (win > MAXWIN && (flags & SADBM_KERNEL_GLOBAL)) || (flags & SADBM_FORCE_GENERIC)) {
        snprintf(text, (size_t)text_len, "matcher=generic kernel=sad_generic_kernel scratch=%zu prep=1", sp.planes_need);
    } else if (flags & SADBM_KERNEL_GLOBAL) {
        snprintf(text, (size_t)text_len, "matcher=quad kernel=sad_quad_kernel<%d> scratch=%zu prep=1", win, sp.planes_need);
    } else {
        int KC = max_disp <= 64 ? 8 : 16;
        if (env().kc) KC = env().kc;
        snprintf(text, (size_t)text_len, "matcher=quad kernel=sad_quad_lds_kernel<%d,%d> scratch=%zu prep=1", win, KC, sp.planes_need);
    }
    return SADBM_OK;
}

int sadbm_ctx_device(const sadbm_ctx *ctx) { return ctx ? ctx->device : -1; }

int sadbm_profile_begin(sadbm_ctx *ctx, int max_launches)
{
    if (!ctx) return SADBM_E_NULL;
    if (max_launches < 1 || max_launches > 65536) return SADBM_E_PARAM;
    DeviceScope dscope(ctx->device);
    HIPCHK(dscope.err);
    if (ctx->ev_cap < max_launches) {
        hipEvent_t *ne = new (std::nothrow) hipEvent_t[3 * (size_t)max_launches];
        if (!ne) return SADBM_E_ALLOC;
        for (int i = 0; i < 3 * ctx->ev_cap; ++i) ne[i] = ctx->ev[i];
        for (int i = 3 * ctx->ev_cap; i < 3 * max_launches; ++i) {
            hipError_t e = hipEventCreate(&ne[i]);
            if (e != hipSuccess) { for (int j = 3 * ctx->ev_cap; j < i; ++j) (void)hipEventDestroy(ne[j]); delete[] ne; return hip_fail(e, "hipEventCreate"); }
        }
        delete[] ctx->ev;
        ctx->ev = ne; ctx->ev_cap = max_launches;
    }
    ctx->ev_noprep.assign((size_t)ctx->ev_cap, 0);
    ctx->ev_n = 0; ctx->prof_on = 1;
    return SADBM_OK;
}

int sadbm_profile_end(sadbm_ctx *ctx, int *n_launches, float *matcher_ms_total, float *prep_ms_total)
{
    if (!ctx) return SADBM_E_NULL;
    ctx->prof_on = 0;
    float mt = 0.f, pt = 0.f;
    for (int i = 0; i < ctx->ev_n; ++i) {
        float a = 0.f, b = 0.f;
        HIPCHK(hipEventSynchronize(ctx->ev[3 * i + 2]));
        if (!ctx->ev_noprep[i]) HIPCHK(hipEventElapsedTime(&a, ctx->ev[3 * i], ctx->ev[3 * i + 1]));
        HIPCHK(hipEventElapsedTime(&b, ctx->ev[3 * i + 1], ctx->ev[3 * i + 2]));
        pt += a; mt += b;
    }
    if (n_launches) *n_launches = ctx->ev_n;
    if (matcher_ms_total) *matcher_ms_total = mt;
    if (prep_ms_total) *prep_ms_total = pt;
    return SADBM_OK;
}

int sadbm_compute_band_device(sadbm_ctx *ctx, const uint8_t *dL, int64_t pitchL, const uint8_t *dR, int64_t pitchR,
                              int frame_rows, int row0, int rows_in, int rows_out, int cols, int channels, int win,
                              int max_disp, uint32_t flags, uint8_t *dOut, int64_t out_pitch, void *stream)
{
    if (!ctx) return SADBM_E_NULL;
    int rc = check_params(frame_rows, cols, channels, win, max_disp, flags);
    if (rc) return rc;
    if (row0 < 0 || rows_in < 0 || rows_out < 0 || row0 + rows_out > frame_rows || row0 + rows_in > frame_rows) return SADBM_E_SIZE;
    if (rows_out == 0 || cols == 0) return SADBM_OK;
    if (!dL || !dR || !dOut) return SADBM_E_NULL;
    if (pitchL < (int64_t)cols * channels || pitchR < (int64_t)cols * channels || out_pitch < cols) return SADBM_E_SIZE;
    int nvalid = frame_rows - win - row0;                     // SAD.cpp:34 on the frame row index
    if (nvalid > rows_out) nvalid = rows_out;
    if (nvalid < 0) nvalid = 0;
    if (cols - win <= 0) nvalid = 0;
    if (nvalid > 0 && rows_in < nvalid + win - 1) return SADBM_E_SIZE;
    hipStream_t st = (hipStream_t)stream;   // NULL = the device's default stream, as in HIP
    return run_pipeline(ctx, dL, pitchL, dR, pitchR, 0, 1, rows_in, rows_out, nvalid, cols, channels, win, max_disp, flags,
                        dOut, out_pitch, 0, st);
}

int sadbm_compute_device(sadbm_ctx *ctx, const uint8_t *dL, int64_t pitchL, const uint8_t *dR, int64_t pitchR, int rows,
                         int cols, int channels, int win, int max_disp, uint32_t flags, uint8_t *dOut,
                         int64_t out_pitch, void *stream)
{
    return sadbm_compute_band_device(ctx, dL, pitchL, dR, pitchR, rows, 0, rows, rows, cols, channels, win, max_disp,
                                     flags, dOut, out_pitch, stream);
}

// ---- the entry point that allocates nothing (include/sadbm.h) ----
// `scratch` is the device array; in host code its NAME is only a stand-in (passing it to a kernel compiles and then
// faults), so its address on each device is resolved once with hipGetSymbolAddress and only that pointer is used.
static __device__ __align__(256) uint8_t scratch[SADBM_STATIC_SCRATCH_BYTES];

namespace {
constexpr int STATIC_MAXDEV = 64;
struct StaticDev { uint8_t *ptr = nullptr; int cus = 256; bool ready = false; };
StaticDev g_static_dev[STATIC_MAXDEV];
std::mutex g_static_mu;

// once per device, with that device current: every GPU holds its own copy of the array at its own address
int static_scratch_for(int device, StaticDev **out)
{
    if (device < 0 || device >= STATIC_MAXDEV) { snprintf(g_hip_err, sizeof(g_hip_err), "device %d out of range", device); return SADBM_E_HIP; }
    std::lock_guard<std::mutex> guard(g_static_mu);
    StaticDev &d = g_static_dev[device];
    if (!d.ready) {
        DeviceScope dscope(device);
        HIPCHK(dscope.err);
        void *ptr = nullptr;
        HIPCHK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) { snprintf(g_hip_err, sizeof(g_hip_err), "hipGetSymbolAddress returned NULL"); return SADBM_E_HIP; }
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) d.cus = v;
        d.ptr = (uint8_t *)ptr;
        d.ready = true;
    }
    *out = &d;
    return SADBM_OK;
}
}  // namespace

int sadbm_compute_device_static(int device, const uint8_t *dL, int64_t pitchL, const uint8_t *dR, int64_t pitchR, int rows,
                                int cols, int channels, int win, int max_disp, uint32_t flags, uint8_t *dOut,
                                int64_t out_pitch, void *stream)
{
    int rc = check_params(rows, cols, channels, win, max_disp, flags);
    if (rc) return rc;
    if (rows == 0 || cols == 0) return SADBM_OK;
    if (!dL || !dR || !dOut) return SADBM_E_NULL;
    if (pitchL < (int64_t)cols * channels || pitchR < (int64_t)cols * channels || out_pitch < cols) return SADBM_E_SIZE;
    StaticDev *sd = nullptr;
    rc = static_scratch_for(device, &sd);
    if (rc) return rc;
    int nvalid = rows - win;                                  // SAD.cpp:34
    if (nvalid < 0 || cols - win <= 0) nvalid = 0;
    // a context that lives on this call's stack and owns nothing: its scratch is the device's static array (sd->ptr,
    // never the array's name), which run_pipeline neither grows nor frees; too large a frame -> SADBM_E_LIMIT
    sadbm_ctx tmp;
    tmp.device = device;
    tmp.planes = sd->ptr;
    tmp.planes_cap = SADBM_STATIC_SCRATCH_BYTES;
    tmp.planes_fixed = true;
    tmp.cus = sd->cus;
    return run_pipeline(&tmp, dL, pitchL, dR, pitchR, 0, 1, rows, rows, nvalid, cols, channels, win, max_disp, flags, dOut,
                        out_pitch, 0, (hipStream_t)stream);
}

int sadbm_compute_batch_device(sadbm_ctx *ctx, const uint8_t *dL, const uint8_t *dR, int64_t pitch_in,
                               int64_t pair_stride_in, int n_pairs, int rows, int cols, int channels, int win,
                               int max_disp, uint32_t flags, uint8_t *dOut, int64_t out_pitch,
                               int64_t pair_stride_out, void *stream)
{
    if (!ctx) return SADBM_E_NULL;
    int rc = check_params(rows, cols, channels, win, max_disp, flags);
    if (rc) return rc;
    if (n_pairs < 0) return SADBM_E_SIZE;
    if (n_pairs == 0 || rows == 0 || cols == 0) return SADBM_OK;
    if (!dL || !dR || !dOut) return SADBM_E_NULL;
    if (pitch_in < (int64_t)cols * channels || out_pitch < cols) return SADBM_E_SIZE;
    if (n_pairs > 1 && (pair_stride_in < pitch_in * (int64_t)rows || pair_stride_out < out_pitch * (int64_t)rows)) return SADBM_E_SIZE;
    int nvalid = rows - win;
    if (nvalid < 0 || cols - win <= 0) nvalid = 0;
    hipStream_t st = (hipStream_t)stream;   // NULL = the device's default stream, as in HIP
    // grid.z carries the pair index, twice the group index in prep_plane_kernel (<= 65535 per launch)
    for (int n0 = 0; n0 < n_pairs; n0 += 32767) {
        int nn = n_pairs - n0 < 32767 ? n_pairs - n0 : 32767;
        rc = run_pipeline(ctx, dL + (int64_t)n0 * pair_stride_in, pitch_in, dR + (int64_t)n0 * pair_stride_in, pitch_in,
                          pair_stride_in, nn, rows, rows, nvalid, cols, channels, win, max_disp, flags,
                          dOut + (int64_t)n0 * pair_stride_out, out_pitch, pair_stride_out, st);
        if (rc) return rc;
    }
    return SADBM_OK;
}

int sadbm_get_disparity_ctx(sadbm_ctx *ctx, const sadbm_image *left, const sadbm_image *right, int win, int max_disp,
                            uint32_t flags, sadbm_plane *out)
{
    if (!ctx || !left || !right || !out) return SADBM_E_NULL;
    int rc = check_params(left->rows, left->cols, left->channels, win, max_disp, flags);
    if (rc) return rc;
    if (right->channels != left->channels) return SADBM_E_TYPE;             // cv::absdiff would throw (SAD.cpp:17)
    if (right->rows < left->rows || right->cols < left->cols) return SADBM_E_SIZE;   // Mat ROI would throw (SAD.cpp:47)
    if (out->rows != left->rows || out->cols != left->cols) return SADBM_E_SIZE;
    const int rows = left->rows, cols = left->cols, ch = left->channels;
    if (rows == 0 || cols == 0) return SADBM_OK;                             // empty Mat: loops skipped (SAD.cpp:26-36)
    if (!left->data || !right->data || !out->data) return SADBM_E_NULL;
    const size_t rowb = (size_t)cols * ch;
    if (left->step < (int64_t)rowb || right->step < (int64_t)rowb || out->step < cols) return SADBM_E_SIZE;
    DeviceScope dscope(ctx->device);
    HIPCHK(dscope.err);
    std::lock_guard<std::recursive_mutex> guard(ctx->mu);                    // the pinned / device staging buffers belong to the context
    // colour inputs: only channel 0 is ever read (SAD.cpp:19-20) -- it is gathered while the rows are copied to pinned memory
    const size_t pin = ((size_t)cols + 255) & ~(size_t)255, pout = pin;
    const size_t in_bytes = 2 * pin * rows, out_bytes = pout * rows;
    // Small frames (the reference's own 384x288 pair): the GPU reads the pinned staging buffer and writes the pinned
    // result directly over PCIe -- three DMA set-ups (2 x H2D, 1 x D2H) cost more than the bytes they would move.
    const bool zero_copy = (long)in_bytes <= env().zc_bytes;
    if (ctx->pin_in_cap < in_bytes) ctx->pin_in_dev = nullptr;
    rc = grow_pinned(&ctx->pin_in, &ctx->pin_in_cap, in_bytes);
    if (rc) return rc;
    if (ctx->pin_out_cap < out_bytes) ctx->pin_out_dev = nullptr;
    rc = grow_pinned(&ctx->pin_out, &ctx->pin_out_cap, out_bytes);
    if (rc) return rc;
    uint8_t *hL = ctx->pin_in, *hR = ctx->pin_in + pin * rows;
    uint8_t *dL, *dR, *dO;
    if (zero_copy) {
        if (!ctx->pin_in_dev || !ctx->pin_out_dev) {
            void *di = nullptr, *dout = nullptr;
            HIPCHK(hipHostGetDevicePointer(&di, ctx->pin_in, 0));
            HIPCHK(hipHostGetDevicePointer(&dout, ctx->pin_out, 0));
            ctx->pin_in_dev = (uint8_t *)di; ctx->pin_out_dev = (uint8_t *)dout;
        }
        dL = ctx->pin_in_dev; dR = dL + pin * rows; dO = ctx->pin_out_dev;
    } else {
        rc = grow(&ctx->stage_in, &ctx->stage_in_cap, in_bytes);
        if (rc) return rc;
        rc = grow(&ctx->stage_out, &ctx->stage_out_cap, out_bytes);
        if (rc) return rc;
        dL = ctx->stage_in; dR = ctx->stage_in + pin * rows; dO = ctx->stage_out;
    }
    hipStream_t st = ctx->stream;
    if (!zero_copy && in_bytes >= ((size_t)4 << 20)) {
        // Large frames, in row chunks: while chunk j's rows are gathered into pinned memory (copy pool), chunk j-1 is on its
        // way to the device, the rows it completed are matched, and finished rows travel back on a second stream and are
        // copied out -- the call then costs little more than the host-side copies (a 4K pair: 25 MB of them).
        int nch = (int)(in_bytes >> 21);                        // one chunk per 2 MB of input, 2 .. 8
        nch = nch < 2 ? 2 : nch > 8 ? 8 : nch;
        const int chunk = (rows + nch - 1) / nch;
        if (!ctx->stream2) HIPCHK(stream_acquire(ctx->device, &ctx->stream2));
        if (!ctx->stream3) HIPCHK(stream_acquire(ctx->device, &ctx->stream3));
        hipStream_t up = ctx->stream3;                          // uploads run on their own stream: chunk j+1 travels while chunk j is matched
        for (int i = 0; i < 3 * 16; ++i)
            if (!ctx->chunk_ev[i]) HIPCHK(hipEventCreateWithFlags(&ctx->chunk_ev[i], hipEventDisableTiming));
        int done = 0, copied = 0;                               // output rows computed so far / chunks copied out so far
        int first[16], last[16];                                // output rows [first, last) of chunk j's compute
        auto copy_out = [&](int j) {
            if (last[j] > first[j])
                copy_rows(out->data + (int64_t)first[j] * out->step, out->step, ctx->pin_out + pout * (size_t)first[j], (int64_t)pout,
                          (size_t)cols, last[j] - first[j]);
        };
        for (int j = 0; j < nch; ++j) {
            const int y0 = j * chunk < rows ? j * chunk : rows, y1 = y0 + chunk < rows ? y0 + chunk : rows;
            if (y1 > y0) {
                copy_rows(hL + pin * (size_t)y0, (int64_t)pin, left->data + (int64_t)y0 * left->step, left->step, (size_t)cols, y1 - y0, ch);
                HIPCHK(hipMemcpyAsync(dL + pin * (size_t)y0, hL + pin * (size_t)y0, pin * (size_t)(y1 - y0), hipMemcpyHostToDevice, up));
                copy_rows(hR + pin * (size_t)y0, (int64_t)pin, right->data + (int64_t)y0 * right->step, right->step, (size_t)cols, y1 - y0, ch);
                HIPCHK(hipMemcpyAsync(dR + pin * (size_t)y0, hR + pin * (size_t)y0, pin * (size_t)(y1 - y0), hipMemcpyHostToDevice, up));
            }
            HIPCHK(hipEventRecord(ctx->chunk_ev[32 + j], up));
            HIPCHK(hipStreamWaitEvent(st, ctx->chunk_ev[32 + j], 0));
            // output rows whose windows (win rows down, SAD.cpp:39) are on the device now; the last chunk closes the frame
            int upto = j == nch - 1 ? rows : y1 - (win - 1);
            if (upto < done) upto = done;
            first[j] = done; last[j] = upto;
            if (upto > done) {
                rc = sadbm_compute_band_device(ctx, dL + pin * (size_t)done, (int64_t)pin, dR + pin * (size_t)done, (int64_t)pin, rows, done,
                                               y1 - done, upto - done, cols, 1, win, max_disp, flags, dO + pout * (size_t)done, (int64_t)pout, st);
                if (rc) { (void)hipStreamSynchronize(up); (void)hipStreamSynchronize(st); (void)hipStreamSynchronize(ctx->stream2); return rc; }
                HIPCHK(hipEventRecord(ctx->chunk_ev[2 * j], st));
                HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->chunk_ev[2 * j], 0));
                HIPCHK(hipMemcpyAsync(ctx->pin_out + pout * (size_t)done, dO + pout * (size_t)done, pout * (size_t)(upto - done),
                                      hipMemcpyDeviceToHost, ctx->stream2));
                done = upto;
            }
            HIPCHK(hipEventRecord(ctx->chunk_ev[2 * j + 1], ctx->stream2));
            while (copied < j && hipEventQuery(ctx->chunk_ev[2 * copied + 1]) == hipSuccess) copy_out(copied++);   // whatever has arrived
        }
        for (; copied < nch; ++copied) {
            HIPCHK(hipEventSynchronize(ctx->chunk_ev[2 * copied + 1]));
            copy_out(copied);
        }
        HIPCHK(hipStreamSynchronize(st));
        return SADBM_OK;
    }
    // caller's (pageable, strided) rows -> pinned staging -> device, one copy for both images
    copy_rows(hL, (int64_t)pin, left->data, left->step, (size_t)cols, rows, ch);
    copy_rows(hR, (int64_t)pin, right->data, right->step, (size_t)cols, rows, ch);
    if (!zero_copy) HIPCHK(hipMemcpyAsync(dL, hL, in_bytes, hipMemcpyHostToDevice, st));
    ctx->host_inputs = zero_copy;
    rc = sadbm_compute_device(ctx, dL, (int64_t)pin, dR, (int64_t)pin, rows, cols, 1, win, max_disp, flags, dO, (int64_t)pout, st);
    ctx->host_inputs = false;
    if (rc) return rc;
    if (!zero_copy) HIPCHK(hipMemcpyAsync(ctx->pin_out, dO, out_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    copy_rows(out->data, out->step, ctx->pin_out, (int64_t)pout, (size_t)cols, rows);
    return SADBM_OK;
}

int sadbm_get_disparity(const sadbm_image *left, const sadbm_image *right, int win, int max_disp, uint32_t flags,
                        sadbm_plane *out)
{
    // on the calling thread's current device (hipSetDevice), as a HIP program expects; a thread that moves to another
    // device gets a new context there
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    if (g_tls.c && g_tls.c->device != dev) sadbm_thread_release();
    if (!g_tls.c) {
        int rc = sadbm_create(dev, &g_tls.c);
        if (rc) return rc;
    }
    return sadbm_get_disparity_ctx(g_tls.c, left, right, win, max_disp, flags, out);
}

void sadbm_thread_release(void)
{
    if (g_tls.c) { sadbm_ctx *d = g_tls.c; g_tls.c = nullptr; sadbm_destroy(d); }
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// streaming pipeline (include/sadbm.h): depth x { pinned in/out, device in/out, ctx with its own stream }
// ---------------------------------------------------------------------------------------------
struct sadbm_pipeline {
    int device, depth, rows, cols, channels, win, max_disp;
    uint32_t flags;
    size_t in_step, out_step;          // bytes per row of the pinned / device buffers (256-byte multiples)
    int64_t next;                      // next ticket
    struct Slot {
        sadbm_ctx *ctx;
        uint8_t *hL, *hR, *hO;         // pinned host
        uint8_t *dL, *dR, *dO;         // device
        hipEvent_t done;
        int64_t ticket;                // ticket in flight or finished, -1 = never used
        bool in_flight;
    } *slots;
};

extern "C" {

void sadbm_pipeline_destroy(sadbm_pipeline *p)
{
    if (!p) return;
    DeviceScope dscope(p->device);
    for (int i = 0; i < p->depth && p->slots; ++i) {
        sadbm_pipeline::Slot &s = p->slots[i];
        if (s.ctx) { (void)hipStreamSynchronize(s.ctx->stream); }
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.hL) (void)hipHostFree(s.hL);
        if (s.hR) (void)hipHostFree(s.hR);
        if (s.hO) (void)hipHostFree(s.hO);
        if (s.dL) (void)hipFree(s.dL);
        if (s.dR) (void)hipFree(s.dR);
        if (s.dO) (void)hipFree(s.dO);
        sadbm_destroy(s.ctx);
    }
    delete[] p->slots;
    delete p;
}

int sadbm_pipeline_create(int device, int depth, int rows, int cols, int channels, int win, int max_disp, uint32_t flags,
                          sadbm_pipeline **out)
{
    if (!out) return SADBM_E_NULL;
    *out = nullptr;
    int rc = check_params(rows, cols, channels, win, max_disp, flags);
    if (rc) return rc;
    if (depth < 1 || depth > 16 || rows < 1 || cols < 1) return SADBM_E_PARAM;
    sadbm_pipeline *p = new (std::nothrow) sadbm_pipeline();
    if (!p) return SADBM_E_ALLOC;
    memset(p, 0, sizeof(*p));
    p->device = device; p->depth = depth; p->rows = rows; p->cols = cols; p->channels = channels;
    p->win = win; p->max_disp = max_disp; p->flags = flags;
    p->in_step = ((size_t)cols * channels + 255) & ~(size_t)255;
    p->out_step = ((size_t)cols + 255) & ~(size_t)255;
    p->slots = new (std::nothrow) sadbm_pipeline::Slot[depth];
    if (!p->slots) { delete p; return SADBM_E_ALLOC; }
    memset(p->slots, 0, sizeof(sadbm_pipeline::Slot) * depth);
    for (int i = 0; i < depth; ++i) {
        sadbm_pipeline::Slot &s = p->slots[i];
        s.ticket = -1;
        rc = sadbm_create(device, &s.ctx);
        hipError_t e = hipSuccess;
        if (!rc) e = hipHostMalloc((void **)&s.hL, p->in_step * rows, hipHostMallocDefault);
        if (!rc && e == hipSuccess) e = hipHostMalloc((void **)&s.hR, p->in_step * rows, hipHostMallocDefault);
        if (!rc && e == hipSuccess) e = hipHostMalloc((void **)&s.hO, p->out_step * rows, hipHostMallocDefault);
        if (!rc && e == hipSuccess) e = hipMalloc((void **)&s.dL, p->in_step * rows);
        if (!rc && e == hipSuccess) e = hipMalloc((void **)&s.dR, p->in_step * rows);
        if (!rc && e == hipSuccess) e = hipMalloc((void **)&s.dO, p->out_step * rows);
        if (!rc && e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
        if (rc || e != hipSuccess) {
            if (!rc) { hip_fail(e, "pipeline allocation"); rc = SADBM_E_ALLOC; }
            sadbm_pipeline_destroy(p);
            return rc;
        }
    }
    *out = p;
    return SADBM_OK;
}

int sadbm_pipeline_inputs(sadbm_pipeline *p, uint8_t **left, uint8_t **right, int64_t *step)
{
    if (!p || !left || !right || !step) return SADBM_E_NULL;
    sadbm_pipeline::Slot &s = p->slots[p->next % p->depth];
    if (s.in_flight) return SADBM_E_BUSY;
    *left = s.hL; *right = s.hR; *step = (int64_t)p->in_step;
    return SADBM_OK;
}

int sadbm_pipeline_submit(sadbm_pipeline *p, const sadbm_image *left, const sadbm_image *right, int64_t *ticket)
{
    if (!p || !ticket) return SADBM_E_NULL;
    if ((left == nullptr) != (right == nullptr)) return SADBM_E_NULL;
    sadbm_pipeline::Slot &s = p->slots[p->next % p->depth];
    if (s.in_flight) return SADBM_E_BUSY;
    const size_t rowb = (size_t)p->cols * p->channels;
    if (left) {
        if (left->rows != p->rows || left->cols != p->cols || left->channels != p->channels) return SADBM_E_SIZE;
        if (right->channels != p->channels) return SADBM_E_TYPE;
        if (right->rows < p->rows || right->cols < p->cols) return SADBM_E_SIZE;
        if (!left->data || !right->data) return SADBM_E_NULL;
        if (left->step < (int64_t)rowb || right->step < (int64_t)rowb) return SADBM_E_SIZE;
        copy_rows(s.hL, (int64_t)p->in_step, left->data, left->step, rowb, p->rows);
        copy_rows(s.hR, (int64_t)p->in_step, right->data, right->step, rowb, p->rows);
    }
    DeviceScope dscope(p->device);
    HIPCHK(dscope.err);
    hipStream_t st = s.ctx->stream;
    HIPCHK(hipMemcpy2DAsync(s.dL, p->in_step, s.hL, p->in_step, rowb, p->rows, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpy2DAsync(s.dR, p->in_step, s.hR, p->in_step, rowb, p->rows, hipMemcpyHostToDevice, st));
    int rc = sadbm_compute_device(s.ctx, s.dL, (int64_t)p->in_step, s.dR, (int64_t)p->in_step, p->rows, p->cols, p->channels,
                                  p->win, p->max_disp, p->flags, s.dO, (int64_t)p->out_step, st);
    if (rc) return rc;
    HIPCHK(hipMemcpy2DAsync(s.hO, p->out_step, s.dO, p->out_step, (size_t)p->cols, p->rows, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(s.done, st));
    s.ticket = p->next; s.in_flight = true;
    *ticket = p->next++;
    return SADBM_OK;
}

int sadbm_pipeline_wait(sadbm_pipeline *p, int64_t ticket, const uint8_t **result, int64_t *result_step, sadbm_plane *out)
{
    if (!p) return SADBM_E_NULL;
    if (ticket < 0 || ticket >= p->next) return SADBM_E_BUSY;
    sadbm_pipeline::Slot &s = p->slots[ticket % p->depth];
    if (s.ticket != ticket) return SADBM_E_BUSY;              // the slot was reused: the result is gone
    if (s.in_flight) {
        HIPCHK(hipEventSynchronize(s.done));
        s.in_flight = false;
    }
    if (result) *result = s.hO;
    if (result_step) *result_step = (int64_t)p->out_step;
    if (out) {
        if (!out->data) return SADBM_E_NULL;
        if (out->rows != p->rows || out->cols != p->cols || out->step < p->cols) return SADBM_E_SIZE;
        copy_rows(out->data, out->step, s.hO, (int64_t)p->out_step, (size_t)p->cols, p->rows);
    }
    return SADBM_OK;
}

}  // extern "C"

