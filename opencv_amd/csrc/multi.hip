// multi.hip -- single-process multi-GPU entry points of include/sadbm.h.
//
// The reference (/root/reference/SAD.cpp:24-61) is single-threaded; every output pixel is independent, so the path
// shards two ways, neither with a cost-volume exchange:
//   * ONE frame -> row bands.  The window is anchored top-left (SAD.cpp:39), so band [r0, r1) needs input rows
//     [r0, min(r1 + win - 1, rows)): a downward halo only.  The loop bound of SAD.cpp:34 is evaluated on the frame row
//     (sadbm_compute_band_device), so the zero frame lands in the last band(s).  Every GPU computes its band straight
//     into its slot of its own copy of the full map; one in-place ncclAllGather of u8 bands (equal counts: bands are
//     ceil(rows / ngpu) rows, the map buffer is padded to ngpu * per rows) leaves the map on every GPU.
//   * a batch of pairs -> pair n on GPU n % ngpu, no collective.
// RCCL is loaded with dlopen at sadbm_multi_create (the library a process already holds is reused -- e.g. the one
// PyTorch bundles -- otherwise /opt/rocm/lib/librccl.so.1), so libsadbm.so has no link-time dependency on it and
// single-GPU users never touch it.  SADBM_MULTI_COPY_GATHER replaces the collective by peer copies behind the same
// interface (and allows one device to appear several times: the partition can be rehearsed on one GPU).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <vector>

#include <mutex>

#include "sadbm_internal.hpp"

namespace {

// The handful of RCCL entry points used, resolved at run time.  Types as in <rccl/rccl.h> (ncclComm_t is an opaque
// pointer, ncclResult_t 0 = success, ncclUint8 = 1).
typedef void *nccl_comm_t;
struct Rccl {
    void *handle = nullptr;
    int (*CommInitAll)(nccl_comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
constexpr int NCCL_UINT8 = 1;

// error text -> the buffer behind sadbm_last_hip_error() (sadbm.hip)
extern "C" char *sadbm_internal_error_buffer(void);
#define g_multi_err sadbm_internal_error_buffer()
#define g_multi_err_size 256

bool load_rccl(Rccl &r)
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) { snprintf(g_multi_err, g_multi_err_size, "dlopen librccl: %s", dlerror()); return false; }
#define SADBM_SYM(field, name) *(void **)(&r.field) = dlsym(r.handle, name); if (!r.field) { snprintf(g_multi_err, g_multi_err_size, "librccl lacks %s", name); return false; }
    SADBM_SYM(CommInitAll, "ncclCommInitAll")
    SADBM_SYM(CommDestroy, "ncclCommDestroy")
    SADBM_SYM(AllGather, "ncclAllGather")
    SADBM_SYM(Broadcast, "ncclBroadcast")
    SADBM_SYM(GroupStart, "ncclGroupStart")
    SADBM_SYM(GroupEnd, "ncclGroupEnd")
    SADBM_SYM(GetErrorString, "ncclGetErrorString")
#undef SADBM_SYM
    return true;
}

struct Gpu {
    int device = 0;
    sadbm_ctx *ctx = nullptr;
    hipStream_t st = nullptr, st_comm = nullptr;   // compute / H2D stream; gather stream of the pipelined variant
    hipEvent_t ev_band = nullptr;                  // band (or sub-band) computed
    hipEvent_t ev_gather = nullptr;                // gather finished (comm stream -> compute stream)
    uint8_t *hin = nullptr, *din = nullptr;        // pinned / device staging of the input rows (L then R)
    size_t in_cap = 0, din_cap = 0;
    uint8_t *dmap = nullptr;                       // this GPU's copy of the full map, parts * per rows x cols, pitch = cols
    size_t map_cap = 0;
    uint8_t *hout = nullptr;                       // pinned result staging (first GPU: the map; batch: each GPU's pairs)
    size_t out_cap = 0;
    nccl_comm_t comm = nullptr;
};

}  // namespace

struct sadbm_multi {
    int n = 0;
    uint32_t mflags = 0;
    Rccl rccl;
    std::vector<Gpu> g;
    int map_rows = 0, map_cols = 0;                // geometry of the last sadbm_compute_multi
    std::mutex mu;                                 // calls on one handle are serialised (they share the staging buffers)
};

namespace {

int hip_err(hipError_t e, const char *what)
{
    snprintf(g_multi_err, g_multi_err_size, "%s: %s", what, hipGetErrorString(e));
    return SADBM_E_HIP;
}
#define MHIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_err(e_, #call); } while (0)
#define MNCCL(m, call) do { int r_ = (call); if (r_ != 0) { snprintf(g_multi_err, g_multi_err_size, "%s: %s", #call, (m)->rccl.GetErrorString(r_)); return SADBM_E_RCCL; } } while (0)

int grow_dev(uint8_t **p, size_t *cap, size_t need)
{
    if (*cap >= need) return SADBM_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    if (hipMalloc((void **)p, need + need / 8 + 4096) != hipSuccess) { *p = nullptr; return SADBM_E_ALLOC; }
    *cap = need + need / 8 + 4096;
    return SADBM_OK;
}
int grow_host(uint8_t **p, size_t *cap, size_t need)
{
    if (*cap >= need) return SADBM_OK;
    if (*p) { (void)hipHostFree(*p); *p = nullptr; *cap = 0; }
    if (hipHostMalloc((void **)p, need + need / 8 + 4096, hipHostMallocDefault) != hipSuccess) { *p = nullptr; return SADBM_E_ALLOC; }
    *cap = need + need / 8 + 4096;
    return SADBM_OK;
}

// host-side row copies: sadbm.hip's pool-parallel copy, which also gathers channel 0 of colour rows (ch > 1) so that only
// the plane the matcher reads (SAD.cpp:19-20) is staged and uploaded
extern "C" void sadbm_internal_copy_rows(uint8_t *dst, int64_t dstep, const uint8_t *src, int64_t sstep, size_t cols, int rows, int ch);
void copy_rows(uint8_t *dst, size_t dstep, const uint8_t *src, int64_t sstep, size_t cols, int rows, int ch = 1)
{
    sadbm_internal_copy_rows(dst, (int64_t)dstep, src, sstep, cols, rows, ch);
}

int check_pair(const sadbm_image *l, const sadbm_image *r, int win, int D, uint32_t flags)
{
    if (!l || !r) return SADBM_E_NULL;
    if (l->rows < 0 || l->cols < 0) return SADBM_E_SIZE;
    if (l->channels != 1 && l->channels != 3 && l->channels != 4) return SADBM_E_TYPE;
    if (r->channels != l->channels) return SADBM_E_TYPE;                        // cv::absdiff would throw (SAD.cpp:17)
    if (win < 1 || D < 1 || ((flags & SADBM_RAW_LOC) && D > 256)) return SADBM_E_PARAM;   // as check_params of sadbm.hip
    if (win > 4096 && l->rows > win && l->cols > win) return SADBM_E_PARAM;
    if (r->rows < l->rows || r->cols < l->cols) return SADBM_E_SIZE;            // Mat ROI would throw (SAD.cpp:47)
    if (l->rows == 0 || l->cols == 0) return SADBM_OK;
    if (!l->data || !r->data) return SADBM_E_NULL;
    const int64_t rowb = (int64_t)l->cols * l->channels;
    if (l->step < rowb || r->step < rowb) return SADBM_E_SIZE;
    return SADBM_OK;
}

}  // namespace

extern "C" {

int sadbm_band_plan(int rows, int parts, int win, int index, int *row0, int *row1, int *in1)
{
    if (!row0 || !row1 || !in1) return SADBM_E_NULL;
    if (rows < 0 || parts < 1 || win < 1 || index < 0 || index >= parts) return SADBM_E_PARAM;
    const int per = (rows + parts - 1) / parts;
    int r0 = index * per, r1 = r0 + per;
    if (r0 > rows) r0 = rows;
    if (r1 > rows) r1 = rows;
    int i1 = r1 + win - 1;                                    // downward halo only (SAD.cpp:39: top-left anchor)
    if (i1 > rows) i1 = rows;
    if (r1 == r0) i1 = r0;
    *row0 = r0; *row1 = r1; *in1 = i1;
    return SADBM_OK;
}

void sadbm_multi_destroy(sadbm_multi *m)
{
    if (!m) return;
    sadbm_detail::DeviceKeeper keep;
    for (Gpu &g : m->g) {
        (void)hipSetDevice(g.device);
        if (g.st) (void)hipStreamSynchronize(g.st);
        if (g.st_comm) (void)hipStreamSynchronize(g.st_comm);
    }
    for (Gpu &g : m->g) {
        (void)hipSetDevice(g.device);
        if (g.comm && m->rccl.CommDestroy) (void)m->rccl.CommDestroy(g.comm);
        if (g.ev_band) (void)hipEventDestroy(g.ev_band);
        if (g.ev_gather) (void)hipEventDestroy(g.ev_gather);
        if (g.hin) (void)hipHostFree(g.hin);
        if (g.hout) (void)hipHostFree(g.hout);
        if (g.din) (void)hipFree(g.din);
        if (g.dmap) (void)hipFree(g.dmap);
        sadbm_destroy(g.ctx);
        sadbm_detail::stream_release(g.device, g.st);
        sadbm_detail::stream_release(g.device, g.st_comm);
    }
    // the RCCL handle stays loaded (other users in the process may hold it)
    delete m;
}

int sadbm_multi_create(int ngpu, const int *devices, uint32_t multi_flags, sadbm_multi **out)
{
    if (!out) return SADBM_E_NULL;
    *out = nullptr;
    if (ngpu < 1 || ngpu > 64) return SADBM_E_PARAM;
    int ndev = 0;
    MHIP(hipGetDeviceCount(&ndev));
    sadbm_detail::DeviceKeeper keep;
    std::vector<int> devs(ngpu);
    bool dup = false;
    for (int i = 0; i < ngpu; ++i) {
        devs[i] = devices ? devices[i] : i;
        if (devs[i] < 0 || devs[i] >= ndev) { snprintf(g_multi_err, g_multi_err_size, "device %d of %d", devs[i], ndev); return SADBM_E_HIP; }
        for (int j = 0; j < i; ++j) dup = dup || devs[j] == devs[i];
    }
    const bool copy_gather = (multi_flags & SADBM_MULTI_COPY_GATHER) != 0;
    if (dup && !copy_gather) return SADBM_E_PARAM;            // RCCL refuses one device twice in a communicator
    sadbm_multi *m = new (std::nothrow) sadbm_multi();
    if (!m) return SADBM_E_ALLOC;
    m->n = ngpu; m->mflags = multi_flags;
    m->g.resize(ngpu);
    for (int i = 0; i < ngpu; ++i) {
        Gpu &g = m->g[i];
        g.device = devs[i];
        hipError_t e = hipSetDevice(g.device);
        int rc = e == hipSuccess ? sadbm_create(g.device, &g.ctx) : SADBM_E_HIP;
        if (!rc) e = sadbm_detail::stream_acquire(g.device, &g.st);
        if (!rc && e == hipSuccess) e = sadbm_detail::stream_acquire(g.device, &g.st_comm);
        if (!rc && e == hipSuccess) e = hipEventCreateWithFlags(&g.ev_band, hipEventDisableTiming);
        if (!rc && e == hipSuccess) e = hipEventCreateWithFlags(&g.ev_gather, hipEventDisableTiming);
        if (rc || e != hipSuccess) { if (!rc) rc = hip_err(e, "multi create"); sadbm_multi_destroy(m); return rc; }
    }
    // peer access for the copy gather and for the band scatter of sadbm_compute_multi_device (ignored where it is already on
    // or unsupported: hipMemcpyPeerAsync then stages)
    for (int i = 0; i < ngpu; ++i)
        for (int j = 0; j < ngpu; ++j)
            if (devs[i] != devs[j]) {
                int can = 0;
                if (hipSetDevice(devs[i]) == hipSuccess && hipDeviceCanAccessPeer(&can, devs[i], devs[j]) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(devs[j], 0);
            }
    (void)hipGetLastError();
    if (!copy_gather) {
        if (!load_rccl(m->rccl)) { sadbm_multi_destroy(m); return SADBM_E_RCCL; }
        std::vector<nccl_comm_t> comms(ngpu, nullptr);
        const int r = m->rccl.CommInitAll(comms.data(), ngpu, devs.data());
        if (r != 0) {
            snprintf(g_multi_err, g_multi_err_size, "ncclCommInitAll: %s", m->rccl.GetErrorString(r));
            sadbm_multi_destroy(m);
            return SADBM_E_RCCL;
        }
        for (int i = 0; i < ngpu; ++i) m->g[i].comm = comms[i];
    }
    *out = m;
    return SADBM_OK;
}

}  // extern "C"

namespace {

// Device rows [r0, i1) of GPU i's band (band + downward halo), as the matcher reads them.
struct BandSrc { const uint8_t *l = nullptr, *r = nullptr; int64_t pl = 0, pr = 0; int ch = 1; };

struct BandJob {
    int rows, cols, win, D, n, per, sub_bands, sper;
    uint32_t flags;
    bool copy_gather;
};

// Sub-band s of GPU i: frame rows [r0 + s*sper, ...) computed straight into the band's slot of that GPU's map.
int issue_sub_band(sadbm_multi *m, const BandJob &j, const BandSrc &b, int i, int s)
{
    Gpu &g = m->g[i];
    int r0, r1, i1;
    sadbm_band_plan(j.rows, j.n, j.win, i, &r0, &r1, &i1);
    MHIP(hipSetDevice(g.device));
    uint8_t *slot = g.dmap + (size_t)i * j.per * j.cols;
    const int s0 = r0 + s * j.sper;                           // frame rows [s0, s1) of this sub-band
    int s1 = s0 + j.sper; if (s1 > r1) s1 = r1;
    if (s1 > s0) {
        int si1 = s1 + j.win - 1; if (si1 > j.rows) si1 = j.rows;
        const int rc = sadbm_compute_band_device(g.ctx, b.l + b.pl * (int64_t)(s0 - r0), b.pl, b.r + b.pr * (int64_t)(s0 - r0), b.pr, j.rows, s0,
                                                 si1 - s0, s1 - s0, j.cols, b.ch, j.win, j.D, j.flags, slot + (size_t)(s0 - r0) * j.cols, j.cols, g.st);
        if (rc) return rc;
    }
    MHIP(hipEventRecord(g.ev_band, g.st));
    if (j.sub_bands > 1) MHIP(hipStreamWaitEvent(g.st_comm, g.ev_band, 0));
    return SADBM_OK;
}

// Pipelined gather of sub-band s: every rank broadcasts its piece into the same place of everyone's map, on the comm
// streams, while the compute streams go on with sub-band s+1.
int gather_sub_band(sadbm_multi *m, const BandJob &j, int s)
{
    const int n = j.n;
    const size_t off = (size_t)s * j.sper * j.cols;
    int cnt_rows = j.sper; if (s * j.sper + cnt_rows > j.per) cnt_rows = j.per - s * j.sper;
    if (cnt_rows <= 0) return SADBM_OK;                       // sub_bands * sper may overshoot the band
    const size_t cnt = (size_t)cnt_rows * j.cols;
    if (!j.copy_gather) {
        MNCCL(m, m->rccl.GroupStart());
        for (int root = 0; root < n; ++root)
            for (int i = 0; i < n; ++i) {
                Gpu &g = m->g[i];
                uint8_t *p = g.dmap + (size_t)root * j.per * j.cols + off;
                const int r = m->rccl.Broadcast(p, p, cnt, NCCL_UINT8, root, g.comm, g.st_comm);
                if (r != 0) { (void)m->rccl.GroupEnd(); snprintf(g_multi_err, g_multi_err_size, "ncclBroadcast: %s", m->rccl.GetErrorString(r)); return SADBM_E_RCCL; }
            }
        MNCCL(m, m->rccl.GroupEnd());
    } else {
        for (int i = 0; i < n; ++i) {                         // destination i pulls every other band's piece
            Gpu &g = m->g[i];
            MHIP(hipSetDevice(g.device));
            for (int k = 0; k < n; ++k) {
                if (k == i) continue;
                MHIP(hipStreamWaitEvent(g.st_comm, m->g[k].ev_band, 0));
                const size_t o = (size_t)k * j.per * j.cols + off;
                MHIP(hipMemcpyPeerAsync(g.dmap + o, g.device, m->g[k].dmap + o, m->g[k].device, cnt, g.st_comm));
            }
        }
    }
    return SADBM_OK;
}

// Everything after sub-band 0 of every GPU has been issued: the remaining sub-bands with their gathers, or the ONE in-place
// all-gather of equal-sized u8 bands (un-pipelined form); on return every g.st is ordered after its GPU's complete map.
int finish_bands(sadbm_multi *m, const BandJob &j, const BandSrc *src)
{
    const int n = j.n;
    if (j.sub_bands > 1) {
        int rc = gather_sub_band(m, j, 0);
        if (rc) return rc;
        for (int s = 1; s < j.sub_bands; ++s) {
            for (int i = 0; i < n; ++i)
                if ((rc = issue_sub_band(m, j, src[i], i, s))) return rc;
            if ((rc = gather_sub_band(m, j, s))) return rc;
        }
        for (int i = 0; i < n; ++i) {                         // compute streams rejoin their gather streams
            Gpu &g = m->g[i];
            MHIP(hipSetDevice(g.device));
            MHIP(hipEventRecord(g.ev_gather, g.st_comm));
            MHIP(hipStreamWaitEvent(g.st, g.ev_gather, 0));
        }
        return SADBM_OK;
    }
    if (!j.copy_gather) {
        MNCCL(m, m->rccl.GroupStart());
        for (int i = 0; i < n; ++i) {
            Gpu &g = m->g[i];
            const int r = m->rccl.AllGather(g.dmap + (size_t)i * j.per * j.cols, g.dmap, (size_t)j.per * j.cols, NCCL_UINT8, g.comm, g.st);
            if (r != 0) { (void)m->rccl.GroupEnd(); snprintf(g_multi_err, g_multi_err_size, "ncclAllGather: %s", m->rccl.GetErrorString(r)); return SADBM_E_RCCL; }
        }
        MNCCL(m, m->rccl.GroupEnd());
    } else {
        for (int i = 0; i < n; ++i) {
            Gpu &g = m->g[i];
            MHIP(hipSetDevice(g.device));
            for (int k = 0; k < n; ++k) {
                if (k == i) continue;
                MHIP(hipStreamWaitEvent(g.st, m->g[k].ev_band, 0));
                const size_t o = (size_t)k * j.per * j.cols;
                MHIP(hipMemcpyPeerAsync(g.dmap + o, g.device, m->g[k].dmap + o, m->g[k].device, (size_t)j.per * j.cols, g.st));
            }
        }
    }
    return SADBM_OK;
}

BandJob make_job(const sadbm_multi *m, int rows, int cols, int win, int D, uint32_t flags, int sub_bands)
{
    BandJob j;
    j.rows = rows; j.cols = cols; j.win = win; j.D = D; j.flags = flags; j.n = m->n;
    j.copy_gather = (m->mflags & SADBM_MULTI_COPY_GATHER) != 0;
    j.per = (rows + j.n - 1) / j.n;                           // rows per band; the last band(s) may be short or empty
    if (sub_bands < 1) sub_bands = 1;
    if (sub_bands > j.per) sub_bands = j.per;
    j.sub_bands = sub_bands;
    j.sper = (j.per + sub_bands - 1) / sub_bands;             // rows per sub-band (same on every GPU: equal counts)
    return j;
}

}  // namespace

extern "C" {

int sadbm_compute_multi(sadbm_multi *m, const sadbm_image *left, const sadbm_image *right, int win, int max_disp,
                        uint32_t flags, int sub_bands, sadbm_plane *out)
{
    if (!m) return SADBM_E_NULL;
    int rc = check_pair(left, right, win, max_disp, flags);
    if (rc) return rc;
    const int rows = left->rows, cols = left->cols, ch = left->channels, n = m->n;
    if (out && (out->rows != rows || out->cols != cols || out->step < cols)) return SADBM_E_SIZE;
    if (rows == 0 || cols == 0) return SADBM_OK;
    if (out && !out->data) return SADBM_E_NULL;
    sadbm_detail::DeviceKeeper keep;
    std::lock_guard<std::mutex> guard(m->mu);
    const BandJob j = make_job(m, rows, cols, win, max_disp, flags, sub_bands);
    const int per = j.per;
    const size_t pin = ((size_t)cols + 255) & ~(size_t)255;   // staged rows hold channel 0 only
    const size_t map_bytes = (size_t)n * per * cols;
    std::vector<BandSrc> src(n);

    // 0. every buffer of every GPU first, while nothing is in flight (an earlier sadbm_compute_multi_device on this handle may
    //    still be: drain it): hipFree / hipHostFree wait for their own device only, and an allocation between GPU i's launch and
    //    GPU i+1's staging would serialise exactly what step 1 overlaps
    for (int i = 0; i < n; ++i) {
        Gpu &g = m->g[i];
        MHIP(hipSetDevice(g.device));
        MHIP(hipStreamSynchronize(g.st));
        MHIP(hipStreamSynchronize(g.st_comm));
    }
    for (int i = 0; i < n; ++i) {
        Gpu &g = m->g[i];
        MHIP(hipSetDevice(g.device));
        if ((rc = grow_dev(&g.dmap, &g.map_cap, map_bytes))) return rc;
        if ((rc = grow_host(&g.hin, &g.in_cap, 2 * pin * (size_t)(per + win)))) return rc;
        if ((rc = grow_dev(&g.din, &g.din_cap, 2 * pin * (size_t)(per + win)))) return rc;
        if (out && (rc = grow_host(&g.hout, &g.out_cap, (size_t)per * cols))) return rc;
    }
    // 1. GPU by GPU: band + halo rows -> pinned (channel 0 gathered on the way) -> device, and the first sub-band is launched
    //    at once: GPU i uploads and computes while the host stages band i+1.  Nothing here blocks the host but the copies.
    for (int i = 0; i < n; ++i) {
        Gpu &g = m->g[i];
        int r0, r1, i1;
        sadbm_band_plan(rows, n, win, i, &r0, &r1, &i1);
        const int rin = i1 - r0;
        MHIP(hipSetDevice(g.device));
        uint8_t *slot = g.dmap + (size_t)i * per * cols;
        if (rin > 0) {
            // the previous call's upload from this pinned buffer has completed: every call ends with a stream synchronise
            copy_rows(g.hin, pin, left->data + (int64_t)r0 * left->step, left->step, (size_t)cols, rin, ch);
            MHIP(hipMemcpyAsync(g.din, g.hin, pin * (size_t)rin, hipMemcpyHostToDevice, g.st));
            copy_rows(g.hin + pin * (size_t)rin, pin, right->data + (int64_t)r0 * right->step, right->step, (size_t)cols, rin, ch);
            MHIP(hipMemcpyAsync(g.din + pin * (size_t)rin, g.hin + pin * (size_t)rin, pin * (size_t)rin, hipMemcpyHostToDevice, g.st));
        }
        if (r1 - r0 < per) MHIP(hipMemsetAsync(slot + (size_t)(r1 - r0) * cols, 0, (size_t)(per - (r1 - r0)) * cols, g.st));
        src[i].l = g.din; src[i].r = g.din + pin * (size_t)rin; src[i].pl = src[i].pr = (int64_t)pin; src[i].ch = 1;
        if ((rc = issue_sub_band(m, j, src[i], i, 0))) return rc;
    }
    // 2. remaining sub-bands and the exchange step
    if ((rc = finish_bands(m, j, src.data()))) return rc;

    // 3. result: every GPU returns ITS band over its own PCIe link (the map itself stays on every GPU); then wait for all
    if (out) {
        for (int i = 0; i < n; ++i) {
            Gpu &g = m->g[i];
            int r0, r1, i1;
            sadbm_band_plan(rows, n, win, i, &r0, &r1, &i1);
            if (r1 <= r0) continue;
            MHIP(hipSetDevice(g.device));
            MHIP(hipMemcpyAsync(g.hout, g.dmap + (size_t)i * per * cols, (size_t)(r1 - r0) * cols, hipMemcpyDeviceToHost, g.st));
        }
    }
    for (int i = 0; i < n; ++i) {
        Gpu &g = m->g[i];
        MHIP(hipSetDevice(g.device));
        MHIP(hipStreamSynchronize(g.st));
        if (out) {
            int r0, r1, i1;
            sadbm_band_plan(rows, n, win, i, &r0, &r1, &i1);
            if (r1 > r0) copy_rows(out->data + (int64_t)r0 * out->step, (size_t)out->step, g.hout, cols, (size_t)cols, r1 - r0);
        }
    }
    m->map_rows = rows; m->map_cols = cols;
    return SADBM_OK;
}

int sadbm_compute_multi_device(sadbm_multi *m, int src_index, const uint8_t *dL, int64_t pitchL, const uint8_t *dR, int64_t pitchR,
                               int rows, int cols, int channels, int win, int max_disp, uint32_t flags, int sub_bands,
                               uint8_t *dOut, int64_t out_pitch, void *stream)
{
    if (!m) return SADBM_E_NULL;
    if (src_index < 0 || src_index >= m->n) return SADBM_E_PARAM;
    if (rows < 0 || cols < 0) return SADBM_E_SIZE;
    if (channels != 1 && channels != 3 && channels != 4) return SADBM_E_TYPE;
    if (win < 1 || max_disp < 1 || ((flags & SADBM_RAW_LOC) && max_disp > 256)) return SADBM_E_PARAM;
    if (win > 4096 && rows > win && cols > win) return SADBM_E_PARAM;
    if (rows == 0 || cols == 0) return SADBM_OK;
    if (!dL || !dR) return SADBM_E_NULL;
    const int64_t rowb = (int64_t)cols * channels;
    if (pitchL < rowb || pitchR < rowb || (dOut && out_pitch < cols)) return SADBM_E_SIZE;
    const int n = m->n;
    sadbm_detail::DeviceKeeper keep;
    std::lock_guard<std::mutex> guard(m->mu);
    const BandJob j = make_job(m, rows, cols, win, max_disp, flags, sub_bands);
    const int per = j.per;
    const size_t prow = ((size_t)rowb + 255) & ~(size_t)255;  // pitch of the peer copies (all channels travel; prep takes channel 0)
    const size_t map_bytes = (size_t)n * per * cols;
    Gpu &gs = m->g[src_index];
    hipStream_t cs = (hipStream_t)stream;
    std::vector<BandSrc> src(n);
    int rc;

    // Buffers first.  This form is asynchronous: the previous call's work may still be in flight on every GPU, and a map or
    // staging buffer that has to grow may still be read by ANOTHER GPU's peer copies (hipFree only waits for its own device):
    // when anything grows, every stream of the handle is drained before the first buffer is released.
    auto reads_in_place = [&](int i) { return i == src_index || (m->g[i].device == gs.device && !(m->mflags & SADBM_MULTI_FORCE_PEER_COPY)); };
    const size_t din_bytes = 2 * prow * (size_t)(per + win);
    bool grows = false;
    for (int i = 0; i < n; ++i) grows = grows || m->g[i].map_cap < map_bytes || (!reads_in_place(i) && m->g[i].din_cap < din_bytes);
    if (grows) {
        for (int i = 0; i < n; ++i) {
            Gpu &g = m->g[i];
            MHIP(hipSetDevice(g.device));
            MHIP(hipStreamSynchronize(g.st));
            MHIP(hipStreamSynchronize(g.st_comm));
        }
        for (int i = 0; i < n; ++i) {
            Gpu &g = m->g[i];
            MHIP(hipSetDevice(g.device));
            if ((rc = grow_dev(&g.dmap, &g.map_cap, map_bytes))) return rc;
            if (!reads_in_place(i) && (rc = grow_dev(&g.din, &g.din_cap, din_bytes))) return rc;
        }
    }
    // the inputs are ready when the caller's stream gets here
    MHIP(hipSetDevice(gs.device));
    MHIP(hipEventRecord(gs.ev_gather, cs));
    for (int i = 0; i < n; ++i) {
        Gpu &g = m->g[i];
        int r0, r1, i1;
        sadbm_band_plan(rows, n, win, i, &r0, &r1, &i1);
        const int rin = i1 - r0;
        MHIP(hipSetDevice(g.device));
        MHIP(hipStreamWaitEvent(g.st, gs.ev_gather, 0));
        uint8_t *slot = g.dmap + (size_t)i * per * cols;
        if (r1 - r0 < per) MHIP(hipMemsetAsync(slot + (size_t)(r1 - r0) * cols, 0, (size_t)(per - (r1 - r0)) * cols, g.st));
        if (reads_in_place(i)) {
            // the band lives on this GPU already: the matcher reads the caller's rows in place
            src[i].l = dL + pitchL * (int64_t)r0; src[i].r = dR + pitchR * (int64_t)r0; src[i].pl = pitchL; src[i].pr = pitchR; src[i].ch = channels;
        } else {
            // band + halo rows cross xGMI once, pulled by the GPU that needs them, on its own stream
            if (rin > 0) {
                const uint8_t *sl = dL + pitchL * (int64_t)r0, *sr = dR + pitchR * (int64_t)r0;
                uint8_t *dl = g.din, *dr = g.din + prow * (size_t)rin;
                if ((size_t)pitchL == prow) MHIP(hipMemcpyPeerAsync(dl, g.device, sl, gs.device, prow * (size_t)(rin - 1) + (size_t)rowb, g.st));
                else MHIP(hipMemcpy2DAsync(dl, prow, sl, (size_t)pitchL, (size_t)rowb, (size_t)rin, hipMemcpyDeviceToDevice, g.st));
                if ((size_t)pitchR == prow) MHIP(hipMemcpyPeerAsync(dr, g.device, sr, gs.device, prow * (size_t)(rin - 1) + (size_t)rowb, g.st));
                else MHIP(hipMemcpy2DAsync(dr, prow, sr, (size_t)pitchR, (size_t)rowb, (size_t)rin, hipMemcpyDeviceToDevice, g.st));
            }
            src[i].l = g.din; src[i].r = g.din + prow * (size_t)rin; src[i].pl = src[i].pr = (int64_t)prow; src[i].ch = channels;
        }
        if ((rc = issue_sub_band(m, j, src[i], i, 0))) return rc;
    }
    if ((rc = finish_bands(m, j, src.data()))) return rc;
    // the map -> the caller's buffer on the source GPU (optional), then the caller's stream waits for every GPU: work queued
    // on it after this call sees the output, every GPU's map, and inputs nobody reads any more.  The host does not wait.
    if (dOut) {
        MHIP(hipSetDevice(gs.device));
        MHIP(hipMemcpy2DAsync(dOut, (size_t)out_pitch, gs.dmap, (size_t)cols, (size_t)cols, (size_t)rows, hipMemcpyDeviceToDevice, gs.st));
    }
    for (int i = 0; i < n; ++i) {
        Gpu &g = m->g[i];
        MHIP(hipSetDevice(g.device));
        MHIP(hipEventRecord(g.ev_gather, g.st));
        MHIP(hipSetDevice(gs.device));
        MHIP(hipStreamWaitEvent(cs, g.ev_gather, 0));
    }
    m->map_rows = rows; m->map_cols = cols;
    return SADBM_OK;
}

int sadbm_multi_device_id(const sadbm_multi *m, int index) { return (m && index >= 0 && index < m->n) ? m->g[index].device : -1; }

int sadbm_multi_device_map(sadbm_multi *m, int index, const uint8_t **dmap, int64_t *pitch)
{
    if (!m || !dmap || !pitch) return SADBM_E_NULL;
    std::lock_guard<std::mutex> guard(m->mu);
    if (index < 0 || index >= m->n || !m->g[index].dmap || m->map_rows == 0) return SADBM_E_PARAM;
    *dmap = m->g[index].dmap;
    *pitch = m->map_cols;
    return SADBM_OK;
}

int sadbm_compute_multi_batch(sadbm_multi *m, const sadbm_image *lefts, const sadbm_image *rights, int n_pairs, int win,
                              int max_disp, uint32_t flags, sadbm_plane *outs)
{
    if (!m || !lefts || !rights || !outs) return SADBM_E_NULL;
    if (n_pairs < 0) return SADBM_E_SIZE;
    if (n_pairs == 0) return SADBM_OK;
    const int rows = lefts[0].rows, cols = lefts[0].cols, ch = lefts[0].channels, n = m->n;
    for (int p = 0; p < n_pairs; ++p) {
        int rc = check_pair(&lefts[p], &rights[p], win, max_disp, flags);
        if (rc) return rc;
        if (lefts[p].rows != rows || lefts[p].cols != cols || lefts[p].channels != ch) return SADBM_E_SIZE;   // one geometry per batch
        if (outs[p].rows != rows || outs[p].cols != cols || outs[p].step < cols) return SADBM_E_SIZE;
        if (rows > 0 && cols > 0 && !outs[p].data) return SADBM_E_NULL;
    }
    if (rows == 0 || cols == 0) return SADBM_OK;
    sadbm_detail::DeviceKeeper keep;
    std::lock_guard<std::mutex> guard(m->mu);
    const size_t pin = ((size_t)cols + 255) & ~(size_t)255, pout = pin;   // staged rows hold channel 0 only
    // buffers first, with the handle drained (see sadbm_compute_multi)
    for (int i = 0; i < n; ++i) {
        Gpu &g = m->g[i];
        MHIP(hipSetDevice(g.device));
        MHIP(hipStreamSynchronize(g.st));
        MHIP(hipStreamSynchronize(g.st_comm));
    }
    for (int i = 0; i < n; ++i) {
        Gpu &g = m->g[i];
        const int mine = (n_pairs - i + n - 1) / n;
        if (mine <= 0) continue;
        MHIP(hipSetDevice(g.device));
        int rc;
        const size_t in_need = 2 * pin * rows * (size_t)mine, out_need = pout * rows * (size_t)mine;
        if ((rc = grow_host(&g.hin, &g.in_cap, in_need))) return rc;
        if ((rc = grow_dev(&g.din, &g.din_cap, in_need))) return rc;
        if ((rc = grow_host(&g.hout, &g.out_cap, out_need))) return rc;
        if ((rc = grow_dev(&g.dmap, &g.map_cap, out_need))) return rc;
    }
    // pair p -> GPU p % n, slot p / n of that GPU's staging; one batched launch sequence per GPU, no collective
    for (int i = 0; i < n; ++i) {
        Gpu &g = m->g[i];
        const int mine = (n_pairs - i + n - 1) / n;
        if (mine <= 0) continue;
        MHIP(hipSetDevice(g.device));
        int rc;
        const size_t in_need = 2 * pin * rows * (size_t)mine, out_need = pout * rows * (size_t)mine;
        uint8_t *hL = g.hin, *hR = g.hin + pin * rows * (size_t)mine;
        for (int k = 0; k < mine; ++k) {
            const int p = i + k * n;
            copy_rows(hL + pin * rows * (size_t)k, pin, lefts[p].data, lefts[p].step, (size_t)cols, rows, ch);
            copy_rows(hR + pin * rows * (size_t)k, pin, rights[p].data, rights[p].step, (size_t)cols, rows, ch);
        }
        MHIP(hipMemcpyAsync(g.din, g.hin, in_need, hipMemcpyHostToDevice, g.st));
        rc = sadbm_compute_batch_device(g.ctx, g.din, g.din + pin * rows * (size_t)mine, (int64_t)pin, (int64_t)(pin * rows), mine, rows, cols, 1,
                                        win, max_disp, flags, g.dmap, (int64_t)pout, (int64_t)(pout * rows), g.st);
        if (rc) return rc;
        MHIP(hipMemcpyAsync(g.hout, g.dmap, out_need, hipMemcpyDeviceToHost, g.st));
    }
    for (int i = 0; i < n; ++i) {
        Gpu &g = m->g[i];
        const int mine = (n_pairs - i + n - 1) / n;
        if (mine <= 0) continue;
        MHIP(hipSetDevice(g.device));
        MHIP(hipStreamSynchronize(g.st));
        for (int k = 0; k < mine; ++k) {
            const int p = i + k * n;
            copy_rows(outs[p].data, (size_t)outs[p].step, g.hout + pout * rows * (size_t)k, (int64_t)pout, (size_t)cols, rows);
        }
    }
    m->map_rows = 0;                                          // dmap holds batch outputs now, not a frame map
    return SADBM_OK;
}

}  // extern "C"
