// quad_kernels.hip -- the "quad" matcher of libsadbm (see sadbm.hip for the pipeline and DESIGN.md section 4.2):
//   * one thread owns 4 adjacent output columns (a "quad", x0 = 4q) and a chunk of KC = 8/16 consecutive
//     disparities, and marches down a band of rows;
//   * per row and disparity the four horizontal window sums are built from v_sad_u8 on 4-byte words:
//     words that lie inside all four windows are summed once, the ragged ends use byte masks;
//   * the vertical window sum is a running sum in registers:  C += H(row+win-1) - H(row-1);
//   * costs are accumulated in the high half of a 32-bit register whose low half holds the disparity,
//     so v_min_u32 on the register IS the first-minimum argmin (cost<<16 | k);
//   * chunks of one quad live in different threads of the block and are merged through LDS once per row.
// Arithmetic of getDisparity()/sub_kernel(), /root/reference/SAD.cpp:13-61.
#ifndef SADBM_QUAD_PRIO_PHASE
#define SADBM_QUAD_PRIO_PHASE 5                                // sad_quad_lds_kernel: issue priority by phase of the step (see its march loop; 0 none, 2: experiment builds)
#endif
#include "sadbm_internal.hpp"

namespace sadbm_detail {
namespace {

#ifndef SADBM_MERGE_ENDS
#define SADBM_MERGE_ENDS 1
#endif
#ifndef SADBM_WPE
#define SADBM_WPE 2
#endif
constexpr int KCG = 16;       // disparities per thread of the L1-operand variant (sad_quad_kernel)
typedef uint32_t __attribute__((aligned(1))) u32u;   // unaligned dword load (global_load_dword handles it)

// ---------------------------------------------------------------------------------------------
// 2. the matcher
// ---------------------------------------------------------------------------------------------
template <int WIN>
struct Geo {
    static constexpr int NW = (WIN + 6) / 4;                 // words covering bytes [0, WIN+3) of a quad's span
    static constexpr bool HI16 = (WIN * WIN * 255 <= 65535); // cost fits the high half of the key
    // Slots that never compete (k > column, SAD.cpp:44; k >= D) are fixed per thread for the whole march, so where the
    // cost field has a spare top bit they start from a "poisoned" running sum that no valid cost reaches -- the running
    // add/subtract carries the offset along and the argmin needs no per-slot conditions.  (16-bit costs: windows up to
    // 11x11; 32-bit costs: always.)
    static constexpr bool POISON = HI16 ? (WIN * WIN * 255 < 32768) : true;
    static constexpr uint32_t POISON_VALUE = HI16 ? 0x80000000u : 0x00800000u;
    // bytes of word j that fall into the window of column x0+a, i.e. positions [a, a+WIN)
    static constexpr uint32_t mask(int j, int a)
    {
        uint32_t m = 0;
        for (int b = 0; b < 4; ++b) {
            int pos = 4 * j + b;
            if (pos >= a && pos < a + WIN) m |= 0xFFu << (8 * b);
        }
        return m;
    }
    static constexpr bool full(int j, int a) { return mask(j, a) == 0xFFFFFFFFu; }
    // word j is inside all four windows
    static constexpr bool shared_all(int j) { return full(j, 0) && full(j, 1) && full(j, 2) && full(j, 3); }
    // word j is inside the windows of a >= 1..3 but not a = 0 (at most one such j >= 1 exists)
    static constexpr bool shared_hi(int j) { return j >= 1 && full(j, 3) && !full(j, 0); }
    static constexpr int popc(uint32_t m) { int n = 0; for (int b = 0; b < 4; ++b) n += (m >> (8 * b)) & 1; return n; }
    static constexpr bool partial(int j, int a) { return mask(j, a) != 0 && !full(j, a); }
    static constexpr int npartial(int a) { int n = 0; for (int j = 0; j < NW; ++j) n += partial(j, a); return n; }
    static constexpr int partial_j(int a, int idx)       // idx-th partial word of alignment a, or -1
    {
        for (int j = 0; j < NW; ++j) if (partial(j, a)) { if (idx == 0) return j; --idx; }
        return -1;
    }
    // v_perm_b32 selector: in-mask bytes of the low source (word j1) first, then those of the high source
    // (word j2), remaining bytes constant 0x00 (selector 0x0c)
    static constexpr uint32_t perm_sel(uint32_t m1, uint32_t m2)
    {
        uint32_t sel = 0; int n = 0;
        for (int b = 0; b < 4; ++b) if ((m1 >> (8 * b)) & 1) sel |= (uint32_t)b << (8 * n++);
        for (int b = 0; b < 4; ++b) if ((m2 >> (8 * b)) & 1) sel |= (uint32_t)(4 + b) << (8 * n++);
        for (; n < 4; ++n) sel |= 0x0cu << (8 * n);
        return sel;
    }
    static constexpr bool use_hi(int a)
    {
        for (int j = 0; j < NW; ++j) if (shared_hi(j) && full(j, a)) return true;
        return false;
    }
};

template <bool HI16>
__device__ __forceinline__ uint32_t sad_acc(uint32_t a, uint32_t b, uint32_t acc)
{
    if constexpr (HI16) return __builtin_amdgcn_sad_hi_u8(a, b, acc);   // acc + (sum|a_i-b_i| << 16)
    else                return __builtin_amdgcn_sad_u8(a, b, acc);
}

// Horizontal window sums of one image row for one disparity, folded into the running sums.
//   ADD:  C[a] += H_a(row)          SUB: C[a] -= H_a(row)
// Lw: the NW words L[row][x0 .. x0+4NW) ; Rw: the NW words R[row][x0-k .. x0-k+4NW).
template <int WIN>
__device__ __forceinline__ void row_bases(const uint32_t (&Lw)[Geo<WIN>::NW], const uint32_t *Rw, uint32_t &b0, uint32_t &b1)
{
    using G = Geo<WIN>;
    b0 = 0;
#pragma unroll
    for (int j = 0; j < G::NW; ++j)
        if (G::shared_all(j)) b0 = sad_acc<G::HI16>(Lw[j], Rw[j], b0);
    b1 = b0;
#pragma unroll
    for (int j = 0; j < G::NW; ++j)
        if (G::shared_hi(j)) b1 = sad_acc<G::HI16>(Lw[j], Rw[j], b1);
}

template <int WIN, int A>
__device__ __forceinline__ uint32_t row_ends(const uint32_t (&Lw)[Geo<WIN>::NW], const uint32_t *Rw, uint32_t acc)
{
    using G = Geo<WIN>;
    // Ragged window ends.  When the window of column x0+A has a partial word at both ends and together
    // they hold <= 4 bytes (WIN = 3 mod 4), one v_perm_b32 per image packs both into a single word, so
    // the pair costs perm + sad instead of 2 x (and + sad); the L-side perm is loop-invariant in k.
    constexpr int j1 = G::partial_j(A, 0), j2 = G::partial_j(A, 1);
    constexpr bool merge = SADBM_MERGE_ENDS && G::npartial(A) == 2 &&
                           (G::popc(G::mask(j1 < 0 ? 0 : j1, A)) + G::popc(G::mask(j2 < 0 ? 0 : j2, A)) <= 4);
#pragma unroll
    for (int j = 0; j < G::NW; ++j) {
        const uint32_t m = G::mask(j, A);
        if (G::shared_all(j)) continue;
        if (G::shared_hi(j) && G::full(j, A)) continue;
        if (m == 0) continue;
        if (m == 0xFFFFFFFFu) acc = sad_acc<G::HI16>(Lw[j], Rw[j], acc);
        else if (!merge)      acc = sad_acc<G::HI16>(Lw[j] & m, Rw[j] & m, acc);
    }
    if constexpr (merge) {
        constexpr uint32_t sel = G::perm_sel(G::mask(j1, A), G::mask(j2, A));
        const uint32_t a = __builtin_amdgcn_perm(Lw[j2], Lw[j1], sel);
        const uint32_t b = __builtin_amdgcn_perm(Rw[j2], Rw[j1], sel);
        acc = sad_acc<G::HI16>(a, b, acc);
    }
    return acc;
}

template <int WIN, int A>
__device__ __forceinline__ uint32_t base_sel(uint32_t b0, uint32_t b1)
{
    // b1 includes the shared_hi word; it applies to alignment A iff that word is full for A
    return Geo<WIN>::use_hi(A) ? b1 : b0;
}

template <int WIN>
__device__ __forceinline__ void load_row_L(const uint8_t *row, int x0, uint32_t (&Lw)[Geo<WIN>::NW])
{
#pragma unroll
    for (int j = 0; j < Geo<WIN>::NW; ++j) Lw[j] = *(const uint32_t *)(row + x0 + 4 * j);
}

// One marching step for the 16 disparities of this thread.
//   HAS_OLD = false: C += H(new row)              (warm-up)
//   HAS_OLD = true : C += H(new row) - H(old row)
template <int WIN, bool HAS_OLD>
__device__ __forceinline__ void march_step(const uint8_t *Lnew, const uint8_t *Rnew, const uint8_t *Lold,
                                           const uint8_t *Rold, int x0, int k0, uint32_t (&C)[KCG][4])
{
    using G = Geo<WIN>;
    constexpr int NW = G::NW;
    constexpr int KC = KCG, TMAX = KC / 4;
    (void)KC;
    uint32_t Ln[NW], Lo[NW];
    load_row_L<WIN>(Lnew, x0, Ln);
    if constexpr (HAS_OLD) load_row_L<WIN>(Lold, x0, Lo);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // disparities k0+c+4t, t=0..3 share byte alignment: word j of disparity t is Wn[j - t + 3]
        constexpr int NR = NW + TMAX - 1;
        uint32_t Wn[NR], Wo[NR];
        const uint8_t *pn = Rnew + (x0 - k0 - c - 4 * (TMAX - 1));
#pragma unroll
        for (int i = 0; i < NR; ++i) Wn[i] = *(const u32u *)(pn + 4 * i);
        if constexpr (HAS_OLD) {
            const uint8_t *po = Rold + (x0 - k0 - c - 4 * (TMAX - 1));
#pragma unroll
            for (int i = 0; i < NR; ++i) Wo[i] = *(const u32u *)(po + 4 * i);
        }
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            const int kk = c + 4 * t;
            const uint32_t *Rn = &Wn[TMAX - 1 - t];
            uint32_t bn0, bn1, bo0 = 0, bo1 = 0;
            row_bases<WIN>(Ln, Rn, bn0, bn1);
            const uint32_t *Ro = &Wo[TMAX - 1 - t];
            if constexpr (HAS_OLD) row_bases<WIN>(Lo, Ro, bo0, bo1);
#define SADBM_UPD(A)                                                                        \
            {                                                                               \
                uint32_t n = row_ends<WIN, A>(Ln, Rn, C[kk][A]);                            \
                n += base_sel<WIN, A>(bn0, bn1);                                            \
                if constexpr (HAS_OLD) n -= row_ends<WIN, A>(Lo, Ro, base_sel<WIN, A>(bo0, bo1)); \
                C[kk][A] = n;                                                               \
            }
            SADBM_UPD(0) SADBM_UPD(1) SADBM_UPD(2) SADBM_UPD(3)
#undef SADBM_UPD
        }
    }
}

// argmin over this thread's 16 disparities for the four columns of its quad.  `slow` (wave-uniform):
// some lane sits at the left border (k > x0+a: the reference keeps cost 0 there, handled by the caller
// through the j+1 candidate) or in the last chunk (k >= D); those slots are excluded.
template <int WIN, int KC>
__device__ __forceinline__ void chunk_min(const uint32_t (&C)[KC][4], int k0, int kmaxrel0, int dlim, bool slow, uint32_t (&best)[4])
{
    using G = Geo<WIN>;
#pragma unroll
    for (int a = 0; a < 4; ++a) best[a] = KEY_MAX;
    if (!slow) {
#pragma unroll
        for (int kk = 0; kk < KC; ++kk)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                uint32_t key = G::HI16 ? C[kk][a] : ((C[kk][a] << 8) | (uint32_t)(k0 + kk));
                best[a] = min(best[a], key);
            }
    } else {
#pragma unroll
        for (int kk = 0; kk < KC; ++kk)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                uint32_t key = G::HI16 ? C[kk][a] : ((C[kk][a] << 8) | (uint32_t)(k0 + kk));
                key = (kk <= min(kmaxrel0 + a, dlim)) ? key : KEY_MAX;
                best[a] = min(best[a], key);
            }
    }
}

template <int WIN>
__global__ void __launch_bounds__(256, SADBM_WPE) sad_quad_kernel(Params p)
{
    using G = Geo<WIN>;
    constexpr int KC = KCG;
    __shared__ uint4 merge[2][256];

    const int tid = threadIdx.x;
    const int q = tid & (p.NQ - 1);
    const int c = tid / p.NQ;
    const int x0 = (blockIdx.x * p.NQ + q) * 4;
    const int k0 = c * KC;
    const int yb = blockIdx.y * p.TY;
    const int ny = min(p.TY, p.nvalid - yb);                 // >= 1 by grid construction
    const uint8_t *L = p.L + (int64_t)blockIdx.z * p.plane_stride;
    const uint8_t *R = p.R + (int64_t)blockIdx.z * p.plane_stride;
    uint8_t *out = p.out + (int64_t)blockIdx.z * p.out_stride;
    const int ocols = p.W - p.win;                            // columns [0,ocols) carry results

    // validity of disparity k0+kk for column x0+a:  k <= min(x0+a, D-1)
    // kmaxrel0 + a bounds kk for the left border; the D bound is folded in through dlim.
    const int dlim = p.D - 1 - k0;                            // kk <= dlim
    const int kmaxrel0 = x0 - k0;                                 // kk <= kmaxrel0 + a
    // a thread computes iff its quad has an output column and its chunk has a valid disparity
    const bool active = (x0 < ocols) && (dlim >= 0) && (kmaxrel0 + 3 >= 0);
    // fold dlim into kmaxrel0 when it is the tighter bound for every a (kk <= min(kmaxrel0+a, dlim))
    const bool slow_thread = active && (kmaxrel0 < KC - 1 || dlim < KC - 1);
    const bool slow = __any(slow_thread);

    uint32_t C[KC][4];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            C[kk][a] = G::HI16 ? (uint32_t)(k0 + kk) : 0u;
            if (G::POISON && kk > min(kmaxrel0 + a, dlim)) C[kk][a] |= G::POISON_VALUE;
        }

    const int nsteps = WIN - 1 + ny;
    for (int r = 0; r < nsteps; ++r) {
        const uint8_t *Ln = L + (int64_t)(yb + r) * p.pitch;
        const uint8_t *Rn = R + (int64_t)(yb + r) * p.pitch;
        if (active) {
            if (r < WIN) {
                march_step<WIN, false>(Ln, Rn, Ln, Rn, x0, k0, C);
            } else {
                const uint8_t *Lo = L + (int64_t)(yb + r - WIN) * p.pitch;
                const uint8_t *Ro = R + (int64_t)(yb + r - WIN) * p.pitch;
                march_step<WIN, true>(Ln, Rn, Lo, Ro, x0, k0, C);
            }
        }
        if (r < WIN - 1) continue;                            // block-uniform

        // ---- emit output row y ----
        const int y = yb + r - (WIN - 1);
        uint32_t best[4] = {KEY_MAX, KEY_MAX, KEY_MAX, KEY_MAX};
        if (active) chunk_min<WIN, KC>(C, k0, kmaxrel0, dlim, G::POISON ? false : slow, best);
        if (p.NCp > 1) {
            const int par = r & 1;
            if (c > 0) merge[par][tid] = make_uint4(best[0], best[1], best[2], best[3]);
            __syncthreads();
            if (c == 0) {
                for (int cc = 1; cc < p.NCp; ++cc) {
                    uint4 o = merge[par][cc * p.NQ + q];
                    best[0] = min(best[0], o.x); best[1] = min(best[1], o.y);
                    best[2] = min(best[2], o.z); best[3] = min(best[3], o.w);
                }
            }
        }
        if (c == 0 && x0 < p.W) {
            uint32_t bytes = 0;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int j = x0 + a;
                uint32_t key = best[a];
                // SAD.cpp:40,44: slots k>j keep cost 0; the first of them (k=j+1) wins unless a valid
                // slot has cost 0 (then that smaller k is already the minimum).
                if (j + 1 < p.D) key = min(key, (uint32_t)(j + 1));
                uint32_t loc = key & 0xFFu;
                uint32_t v = (p.flags & SADBM_RAW_LOC) ? loc : ((loc * 16u) & 0xFFu);   // SAD.cpp:57
                if (j >= ocols) v = 0;
                bytes |= v << (8 * a);
            }
            uint8_t *o = out + (int64_t)y * p.out_pitch + x0;
            if (p.dword_store && x0 + 3 < p.W) {
                *(uint32_t *)o = bytes;
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (x0 + a < p.W) o[a] = (uint8_t)(bytes >> (8 * a));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 2b. the matcher, LDS-staged (default).  Same arithmetic as sad_quad_kernel; the two image rows a
//     marching step touches are staged once per block in LDS -- the L segment as is, the R segment as
//     four copies shifted by 0..3 bytes so that every R word a lane needs (byte offset x0-k+4j) is an
//     ALIGNED dword of copy (k&3) (misaligned ds_read is ~10x slower on gfx950).  The per-lane unaligned
//     global loads of sad_quad_kernel kept the L1 (TCP) ~94 % busy; LDS serves them conflict-free
//     (consecutive lanes read consecutive dwords).
// LDS map (dwords): 2 buffers x 2 row slots (new, old) x { L[LW] | R0[RW] R1 R2 R3 }, then [merge 2x256x4].
// ---------------------------------------------------------------------------------------------
template <int WIN, int KC, bool SUB>
__device__ __forceinline__ void march_pass_lds(const uint32_t *lp, const uint32_t *rp, int RW, uint32_t (&C)[KC][4])
{
    using G = Geo<WIN>;
    constexpr int NW = G::NW;
    constexpr int TMAX = KC / 4;
    constexpr int NR = NW + TMAX - 1;
    uint32_t Lw[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) Lw[j] = lp[j];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint32_t Wr[NR];
        const uint32_t *pr = rp + c * RW;
#pragma unroll
        for (int i = 0; i < NR; ++i) Wr[i] = pr[i];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            const int kk = c + 4 * t;
            const uint32_t *Rw = &Wr[TMAX - 1 - t];
            uint32_t b0, b1;
            row_bases<WIN>(Lw, Rw, b0, b1);
            if constexpr (!SUB) {
                C[kk][0] = row_ends<WIN, 0>(Lw, Rw, C[kk][0]) + base_sel<WIN, 0>(b0, b1);
                C[kk][1] = row_ends<WIN, 1>(Lw, Rw, C[kk][1]) + base_sel<WIN, 1>(b0, b1);
                C[kk][2] = row_ends<WIN, 2>(Lw, Rw, C[kk][2]) + base_sel<WIN, 2>(b0, b1);
                C[kk][3] = row_ends<WIN, 3>(Lw, Rw, C[kk][3]) + base_sel<WIN, 3>(b0, b1);
            } else {
                C[kk][0] -= row_ends<WIN, 0>(Lw, Rw, base_sel<WIN, 0>(b0, b1));
                C[kk][1] -= row_ends<WIN, 1>(Lw, Rw, base_sel<WIN, 1>(b0, b1));
                C[kk][2] -= row_ends<WIN, 2>(Lw, Rw, base_sel<WIN, 2>(b0, b1));
                C[kk][3] -= row_ends<WIN, 3>(Lw, Rw, base_sel<WIN, 3>(b0, b1));
            }
        }
    }
}

// KC = disparities per thread: 16 (4 waves/SIMD) for D >= 128, 8 (5 waves/SIMD, more blocks) for D <= 64.
template <int WIN, int KC>
__global__ void __launch_bounds__(256, (KC == 16 ? (WIN <= 16 ? 4 : 3) : (WIN <= 16 ? 5 : 4))) sad_quad_lds_kernel(Params p)
{
    using G = Geo<WIN>;
    constexpr int TMAX = KC / 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *rows = smem;                                    // [2 buffers][2 slots][ROWW] at LDS offset 0

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int q = tid & (p.NQ - 1);
    const int c = tid / p.NQ;
    // 1-D launch, XCD j walks the contiguous chunk j of the (column block, band, pair) order (as sad_strip_kernel: the
    // left-border blocks take longer, and with 8 column blocks per row a 3-D grid would put them all on one XCD)
    const int chunk = (int)(gridDim.x >> 3);
    const int vb = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (vb >= p.nbx * p.nby * p.npairs) return;               // grid padded to a multiple of 8 (block-uniform exit)
    const int bxi = vb % p.nbx, tb = vb / p.nbx, byi = tb % p.nby, pz = tb / p.nby;
    const int xb = bxi * p.NQ * 4;
    const int x0 = xb + q * 4;
    const int k0 = c * KC;
    const int yb = byi * p.TY;
    const int ny = min(p.TY, p.nvalid - yb);
    const uint8_t *L = p.L + (int64_t)pz * p.plane_stride;
    const uint8_t *R = p.R + (int64_t)pz * p.plane_stride;
    uint8_t *out = p.out + (int64_t)pz * p.out_stride;
    const int ocols = p.W - p.win;
    const int LW = p.LW, RW = p.RW, ROWW = LW + 4 * RW;
    uint4 *merge = reinterpret_cast<uint4 *>(smem + 4 * ROWW);   // [2][256], 16-byte aligned (4*ROWW words)
    const int PADK = p.NCp * KC;                              // R segment starts at column xb - PADK
    const int xlim = p.W + PADR - 4;                          // last addressable dword of a padded row

    const int dlim = p.D - 1 - k0;
    const int kmaxrel0 = x0 - k0;
    const bool active = (x0 < ocols) && (dlim >= 0) && (kmaxrel0 + 3 >= 0);
    const bool slow_thread = active && (kmaxrel0 < KC - 1 || dlim < KC - 1);
    const bool slow = __any(slow_thread);
    const int lofs = q;
    const int rofs = LW + q + (PADK - k0) / 4 - (TMAX - 1);

    uint32_t C[KC][4];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            C[kk][a] = G::HI16 ? (uint32_t)(k0 + kk) : 0u;
            if (G::POISON && kk > min(kmaxrel0 + a, dlim)) C[kk][a] |= G::POISON_VALUE;
        }

    // Staging roles (no per-element index arithmetic): thread group g (64 threads, or the whole block when
    // it is smaller) copies shifted R copy g, g+ng, ...; every thread helps with the L segment.
    const int SG = nthr < 64 ? nthr : 64;
    const int sg = tid / SG, sl = tid - sg * SG, ng = nthr / SG;
    // stage the rows of marching step r into buffer (r & 1): slot 0 = new row yb+r, slot 1 = old row yb+r-WIN
    auto stage = [&](int r) {
        uint32_t *dst = rows + (r & 1) * 2 * ROWW;
        const uint8_t *Ln = L + (int64_t)(yb + r) * p.pitch, *Rn = R + (int64_t)(yb + r) * p.pitch;
        const int64_t back = (int64_t)WIN * p.pitch;          // old row = new row - WIN rows
        const bool has_old = r >= WIN;
        for (int i = tid; i < LW; i += nthr) {
            const int off = xb + 4 * i;
            uint32_t vn = 0, vo = 0;
            if (off <= xlim) {
                vn = *(const uint32_t *)(Ln + off);
                if (has_old) vo = *(const uint32_t *)(Ln + off - back);
            }
            dst[i] = vn;
            if (has_old) dst[ROWW + i] = vo;
        }
        for (int cc = sg; cc < 4; cc += ng) {
            uint32_t *d = dst + LW + cc * RW;
            for (int i = sl; i < RW; i += SG) {
                const int off = xb - PADK + 4 * i - cc;
                uint32_t vn = 0, vo = 0;
                if (off <= xlim) {
                    vn = *(const u32u *)(Rn + off);
                    if (has_old) vo = *(const u32u *)(Rn + off - back);
                }
                d[i] = vn;
                if (has_old) d[ROWW + i] = vo;
            }
        }
    };

    // Common geometries: every thread owns <= 1 L word and <= 2 R words of a row, so the next step's rows are
    // fetched into registers BEFORE the compute of this step and written to the other LDS buffer after it -- the
    // global-load latency hides behind ~600 VALU instructions.  256 threads (RW <= 128): words sl and sl+64 of
    // copy sg; 128 threads (RW <= 64, the half-width block of narrow or badly dividing frames): word sl of the
    // copies sg and sg+2.
    const bool geo256 = (nthr == 256) && (LW <= 256) && (RW <= 128);
    const bool geo128 = (nthr == 128) && (LW <= 128) && (RW <= 64);
    const bool fastgeo = geo256 || geo128;
    const int pf_c1 = geo128 ? sg + 2 : sg, pf_w1 = geo128 ? sl : sl + 64;   // second R word: (copy, word)
    const int pf_lo = xb + 4 * tid;                           // L word of this thread
    const int pf_r0 = xb - PADK + 4 * sl - sg, pf_r1 = xb - PADK + 4 * pf_w1 - pf_c1;
    const int pf_d0 = LW + sg * RW + sl, pf_d1 = LW + pf_c1 * RW + pf_w1;    // their places in a staged row
    // the padded planes of this pair from their first byte (p.L / p.R point PADL bytes into row 0); a plane is < 2 GB
    const auto rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(L) - PADL, 0, 0x7fffffff, 0x00020000);
    const auto rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(R) - PADL, 0, 0x7fffffff, 0x00020000);
    const uint32_t uo_l = (uint32_t)(pf_lo + PADL), uo_r0 = (uint32_t)(pf_r0 + PADL), uo_r1 = (uint32_t)(pf_r1 + PADL);   // >= 0: PADK + 3 <= PADL
    const bool pf_lv = fastgeo && tid < LW && pf_lo <= xlim;
    const bool pf_r0v = fastgeo && sl < RW && pf_r0 <= xlim, pf_r1v = fastgeo && pf_w1 < RW && pf_r1 <= xlim;

    const uint32_t *lp0 = rows + lofs, *rp0 = rows + rofs;
    const int nsteps = WIN - 1 + ny;
    stage(0);
    __syncthreads();
    for (int r = 0; r < nsteps; ++r) {
        const int bo = (r & 1) * 2 * ROWW;
        const bool more = r + 1 < nsteps;
        const bool nold = r + 1 >= WIN;                       // step r+1 has a leaving row
        uint32_t pl[2] = {0, 0}, pr0[2] = {0, 0}, pr1[2] = {0, 0};
        if (fastgeo && more) {
            // buffer loads: one descriptor per padded plane (block-uniform), the row as the scalar offset, the lane's
            // column as an unsigned 32-bit offset -- no 64-bit VALU address arithmetic in the loop
            const uint32_t rown = (uint32_t)((yb + r + 1) * p.pitch), rowo = rown - (uint32_t)(WIN * p.pitch);
            if (pf_lv) { pl[0] = __builtin_amdgcn_raw_buffer_load_b32(rsL, uo_l, rown, 0); if (nold) pl[1] = __builtin_amdgcn_raw_buffer_load_b32(rsL, uo_l, rowo, 0); }
            if (pf_r0v) { pr0[0] = __builtin_amdgcn_raw_buffer_load_b32(rsR, uo_r0, rown, 0); if (nold) pr0[1] = __builtin_amdgcn_raw_buffer_load_b32(rsR, uo_r0, rowo, 0); }
            if (pf_r1v) { pr1[0] = __builtin_amdgcn_raw_buffer_load_b32(rsR, uo_r1, rown, 0); if (nold) pr1[1] = __builtin_amdgcn_raw_buffer_load_b32(rsR, uo_r1, rowo, 0); }
        }
        if (active) {
            // per-thread LDS pointers: one VGPR base per image, small immediate offsets per word
            march_pass_lds<WIN, KC, false>(lp0 + bo, rp0 + bo, RW, C);
            // issue priority (s_setprio) as in the on-the-fly strip kernels: the SUB pass at the default, the ADD pass and the chunk minimum raised --
            // -2.1 ... -3.9 % on windows 2, 3, 31, 40 (profiles/r03o_issue_priority.txt; passes raised and minimum at the default: -0.7 ... +2.4 %)
            if (SADBM_QUAD_PRIO_PHASE == 5) __builtin_amdgcn_s_setprio(0);
            if (r >= WIN) march_pass_lds<WIN, KC, true>(lp0 + bo + ROWW, rp0 + bo + ROWW, RW, C);
        }
        if (more) {                                           // other buffer: last read in step r-1
            if (fastgeo) {
                uint32_t *dst = rows + ((r + 1) & 1) * 2 * ROWW;
                if (tid < LW) { dst[tid] = pl[0]; if (nold) dst[ROWW + tid] = pl[1]; }
                if (sl < RW) { dst[pf_d0] = pr0[0]; if (nold) dst[ROWW + pf_d0] = pr0[1]; }
                if (pf_w1 < RW) { dst[pf_d1] = pr1[0]; if (nold) dst[ROWW + pf_d1] = pr1[1]; }
            } else {
                stage(r + 1);
            }
        }
        const bool emit = r >= WIN - 1;                       // block-uniform
        uint32_t best[4] = {KEY_MAX, KEY_MAX, KEY_MAX, KEY_MAX};
        const int par = r & 1;
        if (SADBM_QUAD_PRIO_PHASE == 5) __builtin_amdgcn_s_setprio(1); else if (SADBM_QUAD_PRIO_PHASE == 2) __builtin_amdgcn_s_setprio(0);
        if (emit) {
            if (active) chunk_min<WIN, KC>(C, k0, kmaxrel0, dlim, G::POISON ? false : slow, best);
            if (c > 0) merge[par * 256 + tid] = make_uint4(best[0], best[1], best[2], best[3]);
        }
        __syncthreads();
        if (SADBM_QUAD_PRIO_PHASE == 2) __builtin_amdgcn_s_setprio(1);
        if (!emit) continue;
        const int y = yb + r - (WIN - 1);
        if (c == 0 && x0 < p.W) {
            for (int cc = 1; cc < p.NCp; ++cc) {
                uint4 o = merge[par * 256 + cc * p.NQ + q];
                best[0] = min(best[0], o.x); best[1] = min(best[1], o.y);
                best[2] = min(best[2], o.z); best[3] = min(best[3], o.w);
            }
            uint32_t bytes = 0;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int j = x0 + a;
                uint32_t key = best[a];
                // SAD.cpp:40,44: slots k>j keep cost 0; the first of them (k=j+1) wins unless a valid
                // slot has cost 0 (then that smaller k is already the minimum).
                if (j + 1 < p.D) key = min(key, (uint32_t)(j + 1));
                uint32_t loc = key & 0xFFu;
                uint32_t v = (p.flags & SADBM_RAW_LOC) ? loc : ((loc * 16u) & 0xFFu);   // SAD.cpp:57
                if (j >= ocols) v = 0;
                bytes |= v << (8 * a);
            }
            uint8_t *o = out + (int64_t)y * p.out_pitch + x0;
            if (p.dword_store && x0 + 3 < p.W) {
                *(uint32_t *)o = bytes;
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (x0 + a < p.W) o[a] = (uint8_t)(bytes >> (8 * a));
            }
        }
    }
}

template <int... I>
quad_kernel_t pick_quad_kernel_t(int win, Seq<I...>)
{
    static const quad_kernel_t table[] = { sad_quad_kernel<I + 1>... };
    return table[win - 1];
}
template <int... I>
quad_kernel_t pick_quad_lds_kernel_t(int win, int kc, Seq<I...>)
{
    static const quad_kernel_t t16[] = { sad_quad_lds_kernel<I + 1, 16>... };
    static const quad_kernel_t t8[] = { sad_quad_lds_kernel<I + 1, 8>... };
    return kc == 16 ? t16[win - 1] : t8[win - 1];
}

}  // namespace

quad_kernel_t pick_quad_kernel(int win) { return pick_quad_kernel_t(win, MakeSeq<MAXWIN>::type()); }
quad_kernel_t pick_quad_lds_kernel(int win, int kc) { return pick_quad_lds_kernel_t(win, kc, MakeSeq<QUAD_MAXWIN>::type()); }

}  // namespace sadbm_detail
