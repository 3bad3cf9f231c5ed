// strip_otf_kernels.hip -- the strip matcher for short searches ("on-the-fly" form): blocks of NG = 1, 2, 4 or 8 groups of 16
// disparities (64 ... 512 threads).  Compiled once per NG (-DSADBM_OTF_PART_NG=1 / 2 / 4 / 8) so the builds run in parallel.
// (NG = 8 serves searches of 65-128 disparities that are not a multiple of 16, where the poisoned sums below beat the staged
// kernel's masked argmin; at multiples of 16 and at NG = 16 the staged kernel is as fast or faster: profiles/r02g_otf_wide_searches.txt.)
//
// Same arithmetic as getDisparity()/sub_kernel() of /root/reference/SAD.cpp:13-61 and the same work split, window
// decomposition, running sums, key phase and finalisation as sad_strip_kernel (sad_strip.inc): a thread owns 16 columns x 4
// disparities of one byte-alignment class.  What differs is where the packed leftover-byte ("E") operands come from.
// sad_strip_kernel has them pre-packed in LDS (4 arrays for L, 16 for R): that row image is as large for 32 disparities
// as for 64, and at 32 it leaves room for 2 waves per SIMD only.  Here the block stages just the plain rows (L once, R
// at the four byte shifts: 1.6 KB per row image instead of 7.2) and every thread builds its E operands itself with
// v_perm_b32 from the aligned words it has loaded for the whole-word SADs anyway: both runs of an E word lie inside one
// aligned word each, and the 16 (quad, disparity) pairs of a thread need only 7 distinct R positions, so a row pass
// costs 16 + 28 permutes on top of its 4*NV + 64 SADs.  A block = 16 strips x 16 NG disparities = 64 NG threads; the 4 NG
// (class, group) combinations are dealt to the waves four at a time (combination q = 4 wave + kgl: group q mod NG, class
// q / NG -- one wave of 4 classes at NG = 1, two waves of 2 classes x 2 groups at NG = 2, one class per wave at NG = 4); the
// one or two staging units per lane and step are prefetched round the ADD and SUB passes.
#ifndef SADBM_OTF_PRIO_PHASE
#define SADBM_OTF_PRIO_PHASE (NG_ >= 2 ? 5 : 0)             // sad_strip_otf_kernel: issue priority by phase of the step (see the march loop)
#endif
#include <type_traits>
#include "sadbm_internal.hpp"

#ifndef SADBM_OTF_PART_NG
#error "compile with -DSADBM_OTF_PART_NG=1, 2, 4 or 8"
#endif

namespace sadbm_detail {

#include "sad_strip.inc"

namespace strip {

template <int WIN, int NG_, bool DIRECT_ = false>
struct OG {
    using G = SG<WIN>;
    static constexpr int NG = NG_;                               // disparity groups of 16 per block
    static constexpr int NSL = NS + NG, RPS = NSL + 2;           // R slots that carry strips / staged per shifted copy
    // an E word from aligned words: low run = nl bytes from byte bl of word wl, high run = nh bytes from byte bh of word wh
    // (word indices relative to the quad's first word)
    static constexpr int wl(int a) { return G::nl(a) ? G::lo0(a) / 4 : G::hi0(a) / 4; }
    static constexpr int bl(int a) { return G::lo0(a) % 4; }
    static constexpr int wh(int a) { return G::nh(a) ? G::hi0(a) / 4 : wl(a); }
    static constexpr int bh(int a) { return G::hi0(a) % 4; }
    static constexpr uint32_t sel(int a)                         // v_perm_b32(S0 = word wh, S1 = word wl)
    {
        uint32_t s = 0; int n = 0;
        for (int b = 0; b < G::nl(a); ++b) s |= (uint32_t)(bl(a) + b) << (8 * n++);
        for (int b = 0; b < G::nh(a); ++b) s |= (uint32_t)(4 + bh(a) + b) << (8 * n++);
        for (; n < 4; ++n) s |= 0x0cu << (8 * n);
        return s;
    }
    // E words whose bytes are ONE run inside one aligned word need no compaction: both operands masked in place give the same SAD, and
    // v_and_b32 (VOP2) issues faster than v_perm_b32 (profiles/r01_ubench_valu_mix_vs_occupancy.txt).  Worth it since the issue priorities of
    // round 3 left these kernels bounded by their instruction stream alone.
#ifndef SADBM_OTF_ANDMASK
#define SADBM_OTF_ANDMASK 1
#endif
    static constexpr bool single(int a) { return SADBM_OTF_ANDMASK && G::ne(a) > 0 && (G::nl(a) == 0 || G::nh(a) == 0); }
    static constexpr uint32_t amask(int a)
    {
        const int b0 = G::nl(a) ? bl(a) : bh(a);
        uint32_t m = 0;
        for (int b = 0; b < G::ne(a); ++b) m |= 0xFFu << (8 * (b0 + b));
        return m;
    }
    static constexpr uint32_t sel_inplace(int a)                 // v_perm_b32(x, x, sel) == x & amask(a): for the lanes that pack staged E words of L (LVS)
    {
        uint32_t s = 0;
        for (int b = 0; b < 4; ++b) s |= (((amask(a) >> (8 * b)) & 0xFFu) ? (uint32_t)b : 0x0cu) << (8 * b);
        return s;
    }
    template <int A>
    static __device__ __forceinline__ uint32_t eword(uint32_t whi, uint32_t wlo)
    {
        if constexpr (G::ne(A) == 0) return 0u;
        else if constexpr (single(A)) return wlo & amask(A);     // wl(A) is the run's word whichever run it is
        else return __builtin_amdgcn_perm(whi, wlo, sel(A));
    }
    static constexpr bool run_ok(int a) { return (G::nl(a) == 0 || bl(a) + G::nl(a) <= 4) && (G::nh(a) == 0 || bh(a) + G::nh(a) <= 4); }
    static constexpr int wmax1(int a) { return G::ne(a) ? (wl(a) > wh(a) ? wl(a) : wh(a)) : 0; }
    static constexpr int WMAX = G::imax(G::imax(wmax1(0), wmax1(1)), G::imax(wmax1(2), wmax1(3)));
    // LVS: the E words of L are the same for every (class, group) of a strip -- 8 or 16 threads would pack the same 16 words
    // per pass -- so blocks of two waves and more stage them once per row (the block's last wave: 64 units of 4 words, two
    // loads and four permutes each, LV[a][4 NS]) and the passes read them with four ds_read_b128; R stays on the fly.
    // Only at NG = 4, where a wave without plain units takes the LV units: a block is as fast as its slowest wave, and the
    // packing wave of a two-wave block would win back 8 of its 32 permutes.  (NG = 1: one wave would pack for itself alone.)
    // Only where the 14 more live registers do not spill: windows up to 13 of the plane-reading variant (2-4.5 % faster
    // there; 15 and up, and the direct variant, measured 1-28 % slower: profiles/r02e_otf_lv_staging_ab.txt).
#ifdef SADBM_OTF_NO_LVS
    static constexpr bool LVS = false;                           // experiment builds
#else
    static constexpr bool LVS = NG >= 4 && WIN <= 13 && !DIRECT_;
#endif
    static constexpr int NLW = LVS ? G::NV : G::imax(G::NV, 3 + WMAX + 1);   // L words a strip reads (index of quad 3 + WMAX)
    static constexpr int NRW = G::imax(G::NV, 3 + WMAX + 1) + 4; // R words: index (word - t + 4), t = 0..3
    static constexpr int NLWP = (NLW + 3) & ~3, NRWP = (NRW + 3) & ~3;
    // LDS row image (dwords): LP[4 (NS+2)] | LV[4][4 NS] (LVS) | RP[4][4 RPS]
    static constexpr int NLP = 4 * (NS + 2), OFF_LV = NLP, SRP = 4 * RPS, OFF_RP = NLP + (LVS ? 16 * NS : 0), ROWW = OFF_RP + 4 * SRP;
    static constexpr int NUNITS = (NS + 2) + 4 * RPS;            // plain 16-byte staging units per row image
    static constexpr int UPL = (NUNITS + 64 * NG - 1) / (64 * NG);   // plain units per lane
    static constexpr bool ok = G::ok && run_ok(0) && run_ok(1) && run_ok(2) && run_ok(3) && NLWP <= 12 && NRWP <= 16 && UPL <= 2;
};

// One image row folded into the running sums of this thread (ADD: entering row, SUB: leaving row).
//   pL: LP + 4*strip; pR: this thread's first R word in RP[c] (one slot below the slot of its (strip, group))
template <int WIN, int NG_, bool DIRECT, bool SUB>
__device__ __forceinline__ void strip_pass_otf(const uint32_t *pL, const uint32_t *pR, uint32_t (&V)[SG<WIN>::VT][SG<WIN>::NV], uint32_t (&E)[4][4][2])
{
    using G = SG<WIN>;
    using O = OG<WIN, NG_, DIRECT>;
    constexpr int NV = G::NV;
    uint32_t lw[O::NLWP], rw[O::NRWP];
#pragma unroll
    for (int nb = 0; nb < O::NLWP; nb += 4) lds128(pL + nb, &lw[nb]);
#pragma unroll
    for (int nb = 0; nb < O::NRWP; nb += 4) lds128(pR + nb, &rw[nb]);
    // ---- V: whole words (packed V: disparities t = 2 tp and 2 tp + 1 share a register, see SG::PKV) ----
    if constexpr (G::PKV) {
#pragma unroll
        for (int tp = 0; tp < 2; ++tp)
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                uint32_t v = SUB ? 0u : V[tp][n];
                v = __builtin_amdgcn_sad_u8(lw[n], rw[n - 2 * tp + 4], v);
                v = __builtin_amdgcn_sad_hi_u8(lw[n], rw[n - 2 * tp + 3], v);
                if constexpr (!SUB) V[tp][n] = v;
                else V[tp][n] -= v;
            }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const uint32_t r = rw[n - t + 4];
                if constexpr (!SUB) V[t][n] = G::HI16 ? __builtin_amdgcn_sad_hi_u8(lw[n], r, V[t][n]) : __builtin_amdgcn_sad_u8(lw[n], r, V[t][n]);
                else V[t][n] -= G::HI16 ? __builtin_amdgcn_sad_hi_u8(lw[n], r, 0u) : __builtin_amdgcn_sad_u8(lw[n], r, 0u);
            }
    }
    // ---- E: leftover bytes, packed here; two column offsets per register (a = 2p low half, 2p+1 high half) ----
    auto e_pair = [&](auto pc) {
        constexpr int p = decltype(pc)::value;
        constexpr int a0 = 2 * p, a1 = 2 * p + 1;
        uint32_t l0[4] = {0, 0, 0, 0}, l1[4] = {0, 0, 0, 0}, r0[7], r1[7];
        if constexpr (O::LVS) {
            if (G::ne(a0) > 0) lds128(pL + O::OFF_LV + a0 * (4 * NS), l0);
            if (G::ne(a1) > 0) lds128(pL + O::OFF_LV + a1 * (4 * NS), l1);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                l0[i] = O::template eword<a0>(lw[i + O::wh(a0)], lw[i + O::wl(a0)]);
                l1[i] = O::template eword<a1>(lw[i + O::wh(a1)], lw[i + O::wl(a1)]);
            }
        }
#pragma unroll
        for (int d = 0; d < 7; ++d) {                            // d = i - t + 3
            r0[d] = O::template eword<a0>(rw[d + 1 + O::wh(a0)], rw[d + 1 + O::wl(a0)]);
            r1[d] = O::template eword<a1>(rw[d + 1 + O::wh(a1)], rw[d + 1 + O::wl(a1)]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t e = SUB ? 0u : E[t][i][p];
                if (G::ne(a0) > 0) e = __builtin_amdgcn_sad_u8(l0[i], r0[i - t + 3], e);
                if (G::ne(a1) > 0) e = __builtin_amdgcn_sad_hi_u8(l1[i], r1[i - t + 3], e);
                if constexpr (!SUB) E[t][i][p] = e;
                else E[t][i][p] -= e;
            }
    };
    e_pair(std::integral_constant<int, 0>());
    e_pair(std::integral_constant<int, 1>());
}

// DIRECT: the staging units are read from the caller's single-channel images themselves (no prep_plane_kernel pass, no
// scratch planes; one frame per "group").  A unit that would leave the frame's bytes [0, ext) -- left of column 0 in the
// first row, past the last row's end -- is assembled byte by byte with zeros outside; everywhere else bytes beyond a
// row's ends are the neighbouring rows' (or the pitch padding's) instead of zeros, and like the zeros of the padded
// planes they only ever reach slots that never compete (k > column) and columns >= cols - win, which are written as 0.
// CH (DIRECT only): channels of the caller's images; 3 / 4 = interleaved BGR / BGRA as cv::imread hands the reference (SAD.cpp:65-66): a
// unit's 16 channel-0 bytes (SAD.cpp:19-20: Add[0]) are gathered from 48 / 64 bytes with v_perm_b32 when the unit is written to LDS
// (round 3: the reference's own call -- one BGR pair -- then is ONE launch, no prep_plane_kernel, no scratch planes).
template <int WIN, int NG_, bool DIRECT, int CH = 1>
__global__ void __launch_bounds__(64 * NG_, 4) sad_strip_otf_kernel(SParams p)
{
    static_assert(CH == 1 || (DIRECT && (CH == 3 || CH == 4)), "colour frames are read in place only");
    using G = SG<WIN>;
    using O = OG<WIN, NG_, DIRECT>;
    constexpr int NV = G::NV;
    constexpr int ROWW = O::ROWW;
    constexpr int NG = O::NG;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nwaves = NG, nthr = 64 * NG;
    uint32_t *rows = smem;                                    // [2 buffers][2 slots][ROWW]
    uint32_t *cells = smem + 4 * ROWW;                        // [2 parities][4 a][64 quads]
    uint32_t *fxs = cells + 2 * 4 * 64;                       // [64] frame / column of each output quad of the block

    // 1-D XCD-aware launch, as sad_strip_kernel: XCD j (= blockIdx & 7) walks the contiguous chunk j of the (column block, band,
    // group) order.  The last p.tail units of every chunk are split into two half-height blocks each: the blocks that run
    // last are then half as long, and so is the stretch at the end of the launch in which CUs run dry one after another.
    const int chunk = (int)(gridDim.x >> 3);                   // blocks per XCD: chunk_units + tail
    const int chunk_units = chunk - p.tail;
    const int q = (int)(blockIdx.x >> 3);
    int vb = (int)(blockIdx.x & 7) * chunk_units, half = -1;
    if (q < chunk_units - p.tail) vb += q;
    else { const int h = q - (chunk_units - p.tail); vb += chunk_units - p.tail + (h >> 1); half = h & 1; }
    if (vb >= p.nbx * p.nby * p.ngroups) return;              // chunks are padded (block-uniform exit)
    const int bxi = vb % p.nbx, tb = vb / p.nbx, byi = tb % p.nby, pz = tb / p.nby;
    const int XB = bxi * XBW;
    const int nf = min(p.FG, p.npairs - pz * p.FG);
    if (XB >= (nf - 1) * p.Wp + p.W) return;                  // block-uniform
    int yb = byi * p.TY;
    int ny = min(p.TY, p.nvalid - yb);
    if (half >= 0) {
        const int h0 = (ny + 1) >> 1;
        if (half == 0) ny = h0; else { yb += h0; ny -= h0; }
        if (ny <= 0) return;                                  // block-uniform
    }
    const bool last_band = yb + ny >= p.nvalid;               // writes the zero frame below the computed rows
    const uint8_t *Lbase = p.Lrow0 + (int64_t)pz * p.pair_stride;   // padded planes (not DIRECT)
    const int ocols = p.W - p.win;

    for (int i = tid; i < 2 * 4 * 64; i += nthr) cells[i] = KEY_MAX;
    if (tid < 64) {
        const int xv = XB + 4 * tid, f = xv / p.Wp, xl = xv - f * p.Wp;
        fxs[tid] = (f < nf) ? ((uint32_t)f << 16) | (uint32_t)xl : 0xFFFFu;
    }

    // ---- staging role of this lane: units tid (+ nthr) of every row image (16 bytes of LP, or of RP[c]) ----
    constexpr int UPL = O::UPL;
    // DIRECT: a unit belongs to the frame of its LAST byte (frames are Wp = a multiple of 16 >= cols columns apart).  Only the units
    // of R's byte-shifted copies can start in the previous frame: the c bytes before column 0 of copy c.  They come from the row
    // above instead -- nobody needs either: class c holds the disparities k >= c, and a competing slot reads R columns
    // x + j - k <= cols - 2 - c of its own frame, or >= 0 (k <= x).
    uint32_t uoff[UPL]; int udst[UPL]; bool uright[UPL]; const uint8_t *ubase[UPL];
#pragma unroll
    for (int j = 0; j < UPL; ++j) {
        const int u = tid + j * nthr;
        uoff[j] = 0; udst[j] = -1; uright[j] = false; ubase[j] = nullptr;
        int P = 0;                                            // first column of the unit in the group's plane
        if (u < NS + 2) {
            P = XB + 16 * u; udst[j] = 4 * u;
            uoff[j] = (uint32_t)(PADL + P);
        } else if (u < O::NUNITS) {
            const int v = u - (NS + 2), c = v / O::RPS, slot = v - c * O::RPS;
            P = XB - c + 16 * (slot - NG);
            uoff[j] = p.lr_delta + (uint32_t)(PADL + P);
            udst[j] = O::OFF_RP + c * O::SRP + 4 * slot; uright[j] = true;
        }
        if constexpr (DIRECT) {
            const int f = P + 15 >= 0 ? (P + 15) / p.Wp : 0;
            uoff[j] = (uint32_t)(P - f * p.Wp);              // column within the frame, -35 ... Wp - 16
            if (f < nf) ubase[j] = (uright[j] ? p.dR : p.dL) + (int64_t)(pz * p.FG + f) * p.in_stride;   // else: no such frame, zeros
        }
    }
    // 16 bytes of row `row`: at plane offset `off` (padded planes) or at column `off` of the frame at `b` (DIRECT); colour frames: the
    // 16 CH bytes that hold them (Raw), channel 0 picked out by gather0 when the unit is written to LDS
    struct Raw { u32x4u q[CH]; };
    auto fetch = [&](bool right, const uint8_t *b, uint32_t off, int row) -> Raw {
        Raw rw;
        if constexpr (!DIRECT) {
            rw.q[0] = *(const u32x4u *)(Lbase + (int64_t)row * p.pitch + off);
        } else {
#pragma unroll
            for (int j = 0; j < CH; ++j) rw.q[j] = u32x4u{0, 0, 0, 0};
            if (b == nullptr) return rw;
            const int o = (right ? row * p.pitchR : row * p.pitchL) + (int)off * CH;   // byte offset within the frame
            const int ext = right ? p.extR : p.extL;
            if (o >= 0 && o + 16 * CH <= ext) {
#pragma unroll
                for (int j = 0; j < CH; ++j) rw.q[j] = *(const u32x4u *)(b + o + 16 * j);
                return rw;
            }
            uint32_t w[4 * CH];                                   // first row left of column 0, or the tail of the last rows
#pragma unroll
            for (int i = 0; i < 4 * CH; ++i) w[i] = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {                        // pixel i of the unit: its channel-0 byte, where it would have been loaded
                const int oi = o + i * CH;
                if (oi >= 0 && oi < ext) w[(i * CH) >> 2] |= (uint32_t)b[oi] << (8 * ((i * CH) & 3));
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) rw.q[j] = u32x4u{w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]};
        }
        return rw;
    };
    // channel 0 of 16 interleaved pixels: BGR word j = bytes 12j, 12j+3, 12j+6, 12j+9 (dwords 3j .. 3j+2, two permutes), BGRA word j =
    // byte 0 of dwords 4j .. 4j+3 (three permutes)
    auto gather0 = [&](const Raw &rw) -> uint4 {
        if constexpr (CH == 1) {
            return make_uint4(rw.q[0].x, rw.q[0].y, rw.q[0].z, rw.q[0].w);
        } else {
            uint32_t d[4 * CH], o[4];
#pragma unroll
            for (int j = 0; j < CH; ++j) { d[4 * j] = rw.q[j].x; d[4 * j + 1] = rw.q[j].y; d[4 * j + 2] = rw.q[j].z; d[4 * j + 3] = rw.q[j].w; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (CH == 3) {
                    const uint32_t t = __builtin_amdgcn_perm(d[3 * j + 1], d[3 * j], 0x0c060300u);       // S1 = d[3j]: bytes 0, 3; S0 = d[3j+1]: byte 2
                    o[j] = __builtin_amdgcn_perm(d[3 * j + 2], t, 0x05020100u);                          // + byte 1 of d[3j+2]
                } else {
                    const uint32_t t1 = __builtin_amdgcn_perm(d[4 * j + 1], d[4 * j], 0x0c0c0400u);
                    const uint32_t t2 = __builtin_amdgcn_perm(d[4 * j + 3], d[4 * j + 2], 0x04000c0cu);
                    o[j] = t1 | t2;
                }
            }
            return make_uint4(o[0], o[1], o[2], o[3]);
        }
    };
    // LV unit of this lane (lanes of the last wave): E words of L for column offset lva, strip lvs
    const bool lv_wave = O::LVS && wave == nwaves - 1;            // scalar
    uint32_t lvo1 = 0, lvo2 = 0, lvsel = 0; int lvdst = 0; const uint8_t *lvbase = nullptr;
    if constexpr (O::LVS) {
        const int lva = lane >> 4, lvs = lane & 15;
        // two-run E words: the staged kernel's packing (unaligned windows at the runs, compacted by G::sel); one-run E words stay in place in their
        // aligned word, masked (the compute lanes mask their R partners the same way: OG::eword)
        const uint32_t sels[4] = {O::single(0) ? O::sel_inplace(0) : G::sel(0), O::single(1) ? O::sel_inplace(1) : G::sel(1),
                                  O::single(2) ? O::sel_inplace(2) : G::sel(2), O::single(3) ? O::sel_inplace(3) : G::sel(3)};
        // a run that is empty reads where the other one does
        const int lo0r[4] = {G::nl(0) ? G::lo0(0) : G::hi0(0), G::nl(1) ? G::lo0(1) : G::hi0(1), G::nl(2) ? G::lo0(2) : G::hi0(2), G::nl(3) ? G::lo0(3) : G::hi0(3)};
        const int lo0s[4] = {O::single(0) ? 4 * O::wl(0) : lo0r[0], O::single(1) ? 4 * O::wl(1) : lo0r[1], O::single(2) ? 4 * O::wl(2) : lo0r[2], O::single(3) ? 4 * O::wl(3) : lo0r[3]};
        const int hi0s[4] = {G::nh(0) && !O::single(0) ? G::hi0(0) : lo0s[0], G::nh(1) && !O::single(1) ? G::hi0(1) : lo0s[1],
                             G::nh(2) && !O::single(2) ? G::hi0(2) : lo0s[2], G::nh(3) && !O::single(3) ? G::hi0(3) : lo0s[3]};
        int P = XB + 16 * lvs;
        lvsel = sels[lva]; lvdst = O::OFF_LV + lva * (4 * NS) + 4 * lvs;
        if constexpr (DIRECT) {
            const int f = P / p.Wp;
            P -= f * p.Wp;
            if (f < nf) lvbase = p.dL + (int64_t)(pz * p.FG + f) * p.in_stride;
        } else {
            P += PADL;
        }
        lvo1 = (uint32_t)(P + lo0s[lva]); lvo2 = (uint32_t)(P + hi0s[lva]);
    }
    // row `row` of the plane -> row slot `slot` of buffer `buf`
    struct Units { Raw w[UPL]; u32x4u v1, v2; };
    auto unit_load = [&](int row) -> Units {
        Units q;
#pragma unroll
        for (int j = 0; j < UPL; ++j) {
#pragma unroll
            for (int k = 0; k < CH; ++k) q.w[j].q[k] = u32x4u{0, 0, 0, 0};
            if (udst[j] >= 0) q.w[j] = fetch(uright[j], ubase[j], uoff[j], row);
        }
        if constexpr (O::LVS) {
            if (lv_wave) { q.v1 = fetch(false, lvbase, lvo1, row).q[0]; q.v2 = fetch(false, lvbase, lvo2, row).q[0]; }
        }
        return q;
    };
    auto unit_store = [&](const Units &q, int buf, int slot) {
#pragma unroll
        for (int j = 0; j < UPL; ++j)
            if (udst[j] >= 0) *(uint4 *)(rows + (buf * 2 + slot) * ROWW + udst[j]) = gather0(q.w[j]);
        if constexpr (O::LVS) {
            if (lv_wave)
                *(uint4 *)(rows + (buf * 2 + slot) * ROWW + lvdst) =
                    make_uint4(__builtin_amdgcn_perm(q.v2.x, q.v1.x, lvsel), __builtin_amdgcn_perm(q.v2.y, q.v1.y, lvsel),
                               __builtin_amdgcn_perm(q.v2.z, q.v1.z, lvsel), __builtin_amdgcn_perm(q.v2.w, q.v1.w, lvsel));
        }
    };

    // ---- compute role ----
    const int kgl = strip_lane_kgl(G::LANES, lane);         // slot within the wave's four (group, class) combinations
    const int sl = strip_lane_strip(G::LANES, lane);        // strip within the block
    const int cq = 4 * wave + kgl;                            // (class, group) combination of this lane
    const int g = cq % NG, c = cq / NG;                       // disparity group (16 k each) and byte-alignment class
    const int kb = 16 * g + c;                                // k_t = kb + 4t
    const int xv = XB + 16 * sl, fr = xv / p.Wp;
    const int xs = xv - fr * p.Wp;
    const bool active = (fr < nf) && (xs < ocols) && (kb < p.D) && (kb <= xs + 15);
    const bool diag = active && (kb + 12 > xs);
    const bool mask_d = !G::POISON_ANY && __any(active && (kb + 12 >= p.D));   // see SG::POISON, SG::KPOISON
    const bool any_diag = G::EPOISON == 0 && __any(diag);     // see SG::EPOISON
    const EmitCtx ec = strip_emit_ctx<WIN>(kb, c, diag, p.D);
    const bool wave_active = __any(active);
    const bool wave_full = __all(active);
    const int sb = sl - g + NG;                               // R slot of (strip, group); reads start one slot lower
    const uint32_t *tL = rows + 4 * sl;
    const uint32_t *tR = rows + O::OFF_RP + c * O::SRP + 4 * (sb - 1);

    uint32_t V[G::VT][NV], E[4][4][2];
    if constexpr (G::PKV) {
        strip_init_sums_pk<WIN>(V, E);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t v0 = (G::HI16 ? (uint32_t)(kb + 4 * t) : 0u) + (kb + 4 * t >= p.D ? (G::HI16 ? G::POISON << 16 : G::POISON) : 0u);
#pragma unroll
            for (int n = 0; n < NV; ++n) V[t][n] = v0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { E[t][i][0] = 0; E[t][i][1] = 0; }
        }
    }
    if constexpr (G::EPOISON != 0) {
        // diagonal lanes (xs == 16 g): slot (t, quad i, column offset a) never competes iff 4 (i - t) + a < c; slots with k >= D
        // are poisoned through V already (both together would overflow the key)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i <= t; ++i)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const bool never = diag && kb + 4 * t < p.D && (i < t || a < c);
                    const uint32_t pe = (G::minus(a) ? (65536u - G::EPOISON) : G::EPOISON) << (16 * (a & 1));
                    E[t][i][a >> 1] += never ? pe : 0u;
                }
    }

    unit_store(unit_load(yb), 0, 0);
    __syncthreads();

    const int nsteps = WIN - 1 + ny;
    for (int r = 0; r < nsteps; ++r) {
        const int bo = (r & 1) * 2 * ROWW;
        const bool emit = r >= WIN - 1;                       // block-uniform
        const int par = r & 1;
        uint32_t *cell = cells + par * 256;
        const bool more = r + 1 < nsteps, more_old = more && r + 1 >= WIN;

        // finalise the previous output row (its atomics completed before the barrier that ended step r-1)
        if (r >= WIN && wave == (r & (nwaves - 1)))
            strip_finalize<WIN>(p, cells + (par ^ 1) * 256, pz * p.FG, fxs[lane], yb + r - WIN, lane, ocols, true);

        // the rows of step r+1 are fetched into registers before the passes of step r and written to the other buffer after them
        Units wn, wo;
        if (more) wn = unit_load(yb + r + 1);
        if (active) strip_pass_otf<WIN, NG_, DIRECT, false>(tL + bo, tR + bo, V, E);
        if (more) unit_store(wn, (r + 1) & 1, 0);
        if (SADBM_OTF_PRIO_PHASE == 5) __builtin_amdgcn_s_setprio(0);
        if (more_old) wo = unit_load(yb + r + 1 - WIN);
        if (active && r >= WIN) strip_pass_otf<WIN, NG_, DIRECT, true>(tL + bo + ROWW, tR + bo + ROWW, V, E);
        if (more_old) unit_store(wo, (r + 1) & 1, 1);

        // Issue priority by phase (s_setprio): the SUB pass at the default, the ADD pass and the key phase raised -- with 4-16 small blocks per
        // CU in different phases, -2.7 ... -5.1 % on every block size from two waves up (profiles/r03o_issue_priority.txt; only the ADD pass
        // or only the key phase raised: -0.4 ... -2 %; passes raised and key phase at the default, the staged kernel's rule: +1.5 ... +2.9 %)
        if (SADBM_OTF_PRIO_PHASE == 5) __builtin_amdgcn_s_setprio(1);
        if (emit && wave_active) {
            uint32_t pub[4] = {KEY_MAX, KEY_MAX, KEY_MAX, KEY_MAX};
            if (mask_d)          strip_emit<WIN, EM_MASK, false>(V, E, active, kgl, ec, pub);
            else if (any_diag)   strip_emit<WIN, EM_DIAG, false>(V, E, active, kgl, ec, pub);
            else if (!wave_full) strip_emit<WIN, EM_FAST, false>(V, E, active, kgl, ec, pub);
            else                 strip_emit<WIN, EM_FAST, true>(V, E, active, kgl, ec, pub);
#pragma unroll
            for (int a = 0; a < 4; ++a) atomicMin(&cell[a * 64 + lane], pub[a]);
        }
        __syncthreads();
    }

    // last output row
    if (ny > 0 && wave == nwaves - 1)
        strip_finalize<WIN>(p, cells + ((nsteps & 1) ^ 1) * 256, pz * p.FG, fxs[lane], yb + nsteps - WIN, lane, ocols, false);

    // the zero frame below the computed rows (SAD.cpp:31,34), written by the blocks of the last band
    if (last_band) {
        const uint32_t fx = fxs[lane];
        const int x0 = (int)(fx & 0xFFFFu);
        if (x0 < p.W) {
            uint8_t *o = p.out + (int64_t)(pz * p.FG + (int)(fx >> 16)) * p.out_stride + x0;
            for (int zr = p.nvalid + wave; zr < p.rows_out; zr += nwaves) {
                uint8_t *q = o + (int64_t)zr * p.out_pitch;
                if (p.dword_store && x0 + 3 < p.W) *(uint32_t *)q = 0u;
                else
                    for (int a = 0; a < 4 && x0 + a < p.W; ++a) q[a] = 0;
            }
        }
    }
}

}  // namespace strip

strip_kernel_t otf_part_1(int win, bool direct, int ch, int *lds_bytes);
strip_kernel_t otf_part_2(int win, bool direct, int ch, int *lds_bytes);
strip_kernel_t otf_part_4(int win, bool direct, int ch, int *lds_bytes);
strip_kernel_t otf_part_8(int win, bool direct, int ch, int *lds_bytes);

namespace {
constexpr int PNG = SADBM_OTF_PART_NG;
// Colour frames in place: BGR, blocks of 128 threads (17-32 disparities: one staging unit per lane, whose three loads stay in flight
// across a pass), windows up to 8 -- where it was measured faster than the prep pass (tools/bgr_ab.py, profiles/r03f_bgr_in_place.txt:
// one 384x288 pair -17 %, 64 x 1080p 7x7 -4.8 %, 16 x 4K 8x8 -4.7 %).  Built and rejected: BGRA (four loads: 10 spilled VGPRs at
// 7x7, +28 %), 256-thread blocks (9x9 / D = 64: 6 spills and no staged E words of L, +16 %; 12x12 +1.5 %), wider windows (6-30 spills).
#ifndef SADBM_OTF_CH_WINS
#define SADBM_OTF_CH_WINS 8
#endif
template <int WIN, bool DIRECT, int CH>
static strip_kernel_t otf_entry()
{
    if constexpr (WIN >= 4 && strip::SG<WIN>::ok && strip::OG<WIN, PNG>::ok && (CH == 1 || (CH == 3 && PNG == 2 && WIN <= SADBM_OTF_CH_WINS)))
        return strip::sad_strip_otf_kernel<WIN, PNG, DIRECT, CH>;
    else return nullptr;
}
template <int WIN, bool DIRECT>
static int otf_lds_one()
{
    if constexpr (WIN >= 4 && strip::SG<WIN>::ok && strip::OG<WIN, PNG>::ok) return (4 * strip::OG<WIN, PNG, DIRECT>::ROWW + 2 * 4 * 64 + 64) * 4;
    else return 0;
}
template <int... I>
strip_kernel_t otf_table(int win, bool direct, int ch, Seq<I...>)
{
    static const strip_kernel_t t[] = { otf_entry<I + 1, false, 1>()... };
    static const strip_kernel_t td[] = { otf_entry<I + 1, true, 1>()... };
    static const strip_kernel_t t3[] = { otf_entry<I + 1, true, 3>()... };
    static const strip_kernel_t t4[] = { otf_entry<I + 1, true, 4>()... };
    return !direct ? t[win - 1] : ch == 1 ? td[win - 1] : ch == 3 ? t3[win - 1] : ch == 4 ? t4[win - 1] : nullptr;
}
template <int... I>
int otf_lds_table(int win, bool direct, Seq<I...>)
{
    static const int t[] = { otf_lds_one<I + 1, false>()... };
    static const int td[] = { otf_lds_one<I + 1, true>()... };
    return direct ? td[win - 1] : t[win - 1];
}
}  // namespace

#define SADBM_CAT2(a, b) a##b
#define SADBM_CAT(a, b) SADBM_CAT2(a, b)
strip_kernel_t SADBM_CAT(otf_part_, SADBM_OTF_PART_NG)(int win, bool direct, int ch, int *lds_bytes)
{
    if (lds_bytes) *lds_bytes = otf_lds_table(win, direct, MakeSeq<MAXWIN>::type());
    return otf_table(win, direct, ch, MakeSeq<MAXWIN>::type());
}

#if SADBM_OTF_PART_NG == 2
strip_kernel_t pick_strip_otf_kernel(int win, int ng, bool direct, int *lds_bytes, int ch)
{
    if (win < 1 || win > MAXWIN) return nullptr;
    if (!direct) ch = 1;                                      // the padded planes hold channel 0 already
    return ng == 1 ? otf_part_1(win, direct, ch, lds_bytes) : ng == 2 ? otf_part_2(win, direct, ch, lds_bytes) : ng == 4 ? otf_part_4(win, direct, ch, lds_bytes)
         : ng == 8 ? otf_part_8(win, direct, ch, lds_bytes) : nullptr;
}
#endif

}  // namespace sadbm_detail
