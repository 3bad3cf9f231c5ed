// sadbm_internal.hpp -- declarations shared by the translation units of libsadbm.so (kernels are compiled in
// separate files so they build in parallel): launch parameter blocks, padding constants, kernel pickers.
#ifndef SADBM_INTERNAL_HPP
#define SADBM_INTERNAL_HPP
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sadbm.h"

// the context's device buffer for filter taps (filters.hip), allocated on first use, freed by sadbm_destroy
extern "C" int sadbm_ctx_taps(sadbm_ctx *ctx, double **taps, int count);

namespace sadbm_detail {

constexpr int PADL = 320;     // >= 256 + 3 + slack; keeps every R read at x0-k-3 >= -PADL addressable
constexpr int PADR = 64;      // >= 4*NW + 8 for the widest window
constexpr int MAXWIN = 25;    // widest window of the strip matchers (and of the L1-operand quad variant)
constexpr int QUAD_MAXWIN = 49; // widest window of sad_quad_lds_kernel (4 * ((win + 6) / 4) <= PADR - 8: a quad's words stay inside the right padding);
                              // beyond it the brute-force kernel takes over -- 300 times slower, so the quad matcher reaches as far as it can
constexpr uint32_t KEY_MAX = 0xFFFFFFFFu;

struct Params {
    const uint8_t *L, *R;     // padded planes, pointing at column 0 of row 0
    int64_t plane_stride;     // bytes between planes of consecutive pairs (batch)
    int pitch;                // plane pitch
    int W;                    // columns
    int nvalid;               // output rows [0,nvalid) are computed (they need input rows < nvalid+win-1)
    int win, D;
    int NQ, NCp;              // quads per block, chunk slots per block (NQ*NCp = blockDim.x)
    int TY;                   // output rows per block
    int LW, RW;               // LDS row segment: L words, R words per shifted copy (sad_quad_lds_kernel)
    int nbx, nby, npairs;     // sad_quad_lds_kernel: column blocks, row bands, pairs of its 1-D launch
    uint8_t *out;
    int64_t out_pitch, out_stride;
    uint32_t flags;
    int dword_store;          // out base and pitch 4-byte aligned
};

namespace strip {
constexpr int NS = 16;                 // strips per block
constexpr int SDIRECT_BIAS = 512;      // >= the longest look-back of a staging unit (16 * 16 + 3 columns, + 16)
constexpr int XBW = NS * 16;           // output columns per block
struct SParams {
    const uint8_t *Lrow0;     // padded L plane of group 0, first byte of row 0 (column -PADL)
    int64_t pair_stride;      // bytes between the L planes of consecutive groups
    uint32_t lr_delta;        // R plane = L plane + lr_delta
    int pitch, W, nvalid, win, D, TY, NG;
    int nbx, nby, ngroups;    // column blocks, row bands, frame groups: the launch is 1-D over their product (padded to 8)
    int FG, Wp, npairs;       // a group = FG frames side by side in one plane, Wp columns apart (Wp % 16 == 0); total frames
    int rows_out;             // output rows [nvalid, rows_out) are zero-filled by the blocks of the last band
    int tail;                 // sad_strip_otf_kernel: the last `tail` (column block, band, group) units of every XCD's chunk run as two half-height blocks
    // direct variant of sad_strip_otf_kernel only: the caller's images (single channel), frames in_stride bytes apart; every
    // byte [0, ext) from a frame's first byte is readable (ext = (rows read - 1) * pitch + cols < 2^31)
    const uint8_t *dL, *dR;
    int64_t in_stride;
    int pitchL, pitchR, extL, extR;
    // direct form of sad_strip_kernel (sdirect != 0): no padded planes either -- Lrow0 is a byte address SDIRECT_BIAS below the lower
    // of the two image batches, staging offsets are 32-bit from there (ll_delta / lr_delta: where the L / R batch starts, bias
    // included), pitch is the images' common row pitch, pair_stride = FG * in_stride.  [loA, loA + lenA) and [loB, loB + lenB) are
    // the byte ranges (offsets from Lrow0) that may be read: the two batches.
    int sdirect, nedge;       // nedge: spare row images in LDS for the rows that cannot be loaded whole (first / last row of a batch)
    uint32_t ll_delta;
    int64_t loA, lenA, loB, lenB;
    uint8_t *out;
    int64_t out_pitch, out_stride;
    uint32_t flags;
    int dword_store;
};
}  // namespace strip

typedef void (*quad_kernel_t)(Params);
typedef void (*strip_kernel_t)(strip::SParams);

// quad_kernels.hip
quad_kernel_t pick_quad_kernel(int win);                 // sad_quad_kernel<win>, 1 <= win <= MAXWIN
quad_kernel_t pick_quad_lds_kernel(int win, int kc);     // sad_quad_lds_kernel<win, kc>, kc in {8, 16}, 1 <= win <= QUAD_MAXWIN
// strip_kernels.hip: nslm = NS + (4, 8 or 16 disparity groups); nullptr where the window has no strip instantiation
strip_kernel_t pick_strip_kernel(int win, int nslm);
// strip_otf_kernels.hip: the on-the-fly form of the strip matcher, ng = 1, 2 or 4 disparity groups per block (64 ng threads); its dynamic LDS size
// direct: the variant that reads the caller's single-channel images itself (SParams::dL ...) instead of padded planes
// ch (direct only): 1, or 3 / 4 = interleaved colour frames whose channel 0 the kernel gathers itself (nullptr where that variant is not built)
strip_kernel_t pick_strip_otf_kernel(int win, int ng, bool direct, int *lds_bytes, int ch = 1);
// host copies of the strip kernel's compile-time geometry: staging passes per row image, dwords per row image
void strip_geometry(int win, int nsl, int *pr, int *roww);

template <int... I> struct Seq {};
template <int N, int... I> struct MakeSeq : MakeSeq<N - 1, N - 1, I...> {};
template <int... I> struct MakeSeq<0, I...> { typedef Seq<I...> type; };

// Makes `device` current for the enclosing scope and puts the caller's device back afterwards: an entry point of this
// library never leaves another device current on the calling thread.
struct DeviceScope {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) { err = hipSetDevice(device); changed = err == hipSuccess && prev >= 0; }
    }
    ~DeviceScope() { if (changed) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// For entry points that walk over several devices: remembers the caller's device and puts it back on exit.
struct DeviceKeeper {
    int prev = -1;
    DeviceKeeper() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceKeeper() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceKeeper(const DeviceKeeper &) = delete;
    DeviceKeeper &operator=(const DeviceKeeper &) = delete;
};

// Streams of the library (contexts, multi-GPU handles) come from a per-device pool and go back to it instead of being destroyed
// (sadbm.hip): hipStreamDestroy of a stream whose last operations were cross-stream waits and asynchronous host copies, right
// after hipStreamSynchronize, crashed inside the runtime's hsa_signal_destroy on MI355X / ROCm 7.0 (profiles/r03b_stream_destroy_crash.txt);
// a pooled stream is idle (the owner synchronises it before the release), non-blocking, and as good as a new one.
hipError_t stream_acquire(int device, hipStream_t *out);    // `device` must be current
void stream_release(int device, hipStream_t s);             // synchronised by the caller; `device` must be current

}  // namespace sadbm_detail
#endif
