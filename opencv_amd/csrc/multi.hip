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

int check_pair(const sadbm_image *l, const sadbm_image *r, int win, int D)
{
    if (!l || !r) return SADBM_E_NULL;
    if (l->rows < 0 || l->cols < 0) return SADBM_E_SIZE;
    if (l->channels != 1 && l->channels != 3 && l->channels != 4) return SADBM_E_TYPE;
    if (r->channels != l->channels) return SADBM_E_TYPE;                        // cv::absdiff would throw (SAD.cpp:17)
    if (win < 1 || win > 255 || D < 1 || D > 256) return SADBM_E_PARAM;
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
        if (g.st) (void)hipStreamDestroy(g.st);
        if (g.st_comm) (void)hipStreamDestroy(g.st_comm);
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
        if (!rc) e = hipStreamCreateWithFlags(&g.st, hipStreamNonBlocking);
        if (!rc && e == hipSuccess) e = hipStreamCreateWithFlags(&g.st_comm, hipStreamNonBlocking);
        if (!rc && e == hipSuccess) e = hipEventCreateWithFlags(&g.ev_band, hipEventDisableTiming);
        if (!rc && e == hipSuccess) e = hipEventCreateWithFlags(&g.ev_gather, hipEventDisableTiming);
        if (rc || e != hipSuccess) { if (!rc) rc = hip_err(e, "multi create"); sadbm_multi_destroy(m); return rc; }
    }
    if (copy_gather) {
        // peer access for the copies (ignored where it is already on or unsupported: hipMemcpyPeerAsync then stages)
        for (int i = 0; i < ngpu; ++i)
            for (int j = 0; j < ngpu; ++j)
                if (devs[i] != devs[j]) {
                    int can = 0;
                    if (hipSetDevice(devs[i]) == hipSuccess && hipDeviceCanAccessPeer(&can, devs[i], devs[j]) == hipSuccess && can)
                        (void)hipDeviceEnablePeerAccess(devs[j], 0);
                }
        (void)hipGetLastError();
    } else {
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

int sadbm_compute_multi(sadbm_multi *m, const sadbm_image *left, const sadbm_image *right, int win, int max_disp,
                        uint32_t flags, int sub_bands, sadbm_plane *out)
{
    if (!m) return SADBM_E_NULL;
    int rc = check_pair(left, right, win, max_disp);
    if (rc) return rc;
    const int rows = left->rows, cols = left->cols, ch = left->channels, n = m->n;
    if (out && (out->rows != rows || out->cols != cols || out->step < cols)) return SADBM_E_SIZE;
    if (rows == 0 || cols == 0) return SADBM_OK;
    if (out && !out->data) return SADBM_E_NULL;
    if (sub_bands < 1) sub_bands = 1;
    sadbm_detail::DeviceKeeper keep;
    std::lock_guard<std::mutex> guard(m->mu);
    const bool copy_gather = (m->mflags & SADBM_MULTI_COPY_GATHER) != 0;
    const int per = (rows + n - 1) / n;                       // rows per band; the last band(s) may be short or empty
    if (sub_bands > per) sub_bands = per;
    const int sper = (per + sub_bands - 1) / sub_bands;       // rows per sub-band (same on every GPU: equal counts)
    const size_t pin = ((size_t)cols + 255) & ~(size_t)255;   // staged rows hold channel 0 only
    const size_t map_bytes = (size_t)n * per * cols;

    // 1. per GPU: band + halo rows -> pinned -> device; zero the band slot's padding rows (bands past the frame's end)
    for (int i = 0; i < n; ++i) {
        Gpu &g = m->g[i];
        int r0, r1, i1;
        sadbm_band_plan(rows, n, win, i, &r0, &r1, &i1);
        const int rin = i1 - r0;
        MHIP(hipSetDevice(g.device));
        if ((rc = grow_dev(&g.dmap, &g.map_cap, map_bytes))) return rc;
        if ((rc = grow_host(&g.hin, &g.in_cap, 2 * pin * (size_t)(per + win)))) return rc;
        if ((rc = grow_dev(&g.din, &g.din_cap, 2 * pin * (size_t)(per + win)))) return rc;
        if (rin > 0) {
            copy_rows(g.hin, pin, left->data + (int64_t)r0 * left->step, left->step, (size_t)cols, rin, ch);
            copy_rows(g.hin + pin * (size_t)rin, pin, right->data + (int64_t)r0 * right->step, right->step, (size_t)cols, rin, ch);
        }
    }
    // 2. H2D + band compute (sub-band by sub-band), each GPU on its own stream; nothing blocks the host
    for (int s = 0; s < sub_bands; ++s) {
        for (int i = 0; i < n; ++i) {
            Gpu &g = m->g[i];
            int r0, r1, i1;
            sadbm_band_plan(rows, n, win, i, &r0, &r1, &i1);
            const int rin = i1 - r0;
            MHIP(hipSetDevice(g.device));
            uint8_t *slot = g.dmap + (size_t)i * per * cols;
            if (s == 0) {
                if (rin > 0) MHIP(hipMemcpyAsync(g.din, g.hin, 2 * pin * (size_t)rin, hipMemcpyHostToDevice, g.st));
                if (r1 - r0 < per) MHIP(hipMemsetAsync(slot + (size_t)(r1 - r0) * cols, 0, (size_t)(per - (r1 - r0)) * cols, g.st));
            }
            const int s0 = r0 + s * sper;                     // frame rows [s0, s1) of this sub-band
            int s1 = s0 + sper; if (s1 > r1) s1 = r1;
            if (s1 > s0) {
                int si1 = s1 + win - 1; if (si1 > rows) si1 = rows;
                const uint8_t *bl = g.din + pin * (size_t)(s0 - r0), *br = g.din + pin * (size_t)rin + pin * (size_t)(s0 - r0);
                rc = sadbm_compute_band_device(g.ctx, bl, (int64_t)pin, br, (int64_t)pin, rows, s0, si1 - s0, s1 - s0, cols, 1, win,
                                               max_disp, flags, slot + (size_t)(s0 - r0) * cols, cols, g.st);
                if (rc) return rc;
            }
            MHIP(hipEventRecord(g.ev_band, g.st));
            if (sub_bands > 1) MHIP(hipStreamWaitEvent(g.st_comm, g.ev_band, 0));
        }
        if (sub_bands == 1) break;
        // pipelined gather of sub-band s: every rank broadcasts its piece into the same place of everyone's map,
        // on the comm streams, while the compute streams go on with sub-band s+1
        const size_t off = (size_t)s * sper * cols;
        int cnt_rows = sper; if (s * sper + cnt_rows > per) cnt_rows = per - s * sper;
        if (cnt_rows <= 0) continue;                          // sub_bands * sper may overshoot the band
        const size_t cnt = (size_t)cnt_rows * cols;
        if (!copy_gather) {
            MNCCL(m, m->rccl.GroupStart());
            for (int root = 0; root < n; ++root)
                for (int i = 0; i < n; ++i) {
                    Gpu &g = m->g[i];
                    uint8_t *p = g.dmap + (size_t)root * per * cols + off;
                    const int r = m->rccl.Broadcast(p, p, cnt, NCCL_UINT8, root, g.comm, g.st_comm);
                    if (r != 0) { (void)m->rccl.GroupEnd(); snprintf(g_multi_err, g_multi_err_size, "ncclBroadcast: %s", m->rccl.GetErrorString(r)); return SADBM_E_RCCL; }
                }
            MNCCL(m, m->rccl.GroupEnd());
        } else {
            for (int i = 0; i < n; ++i) {                     // destination i pulls every other band's piece
                Gpu &g = m->g[i];
                MHIP(hipSetDevice(g.device));
                for (int j = 0; j < n; ++j) {
                    if (j == i) continue;
                    MHIP(hipStreamWaitEvent(g.st_comm, m->g[j].ev_band, 0));
                    const size_t o = (size_t)j * per * cols + off;
                    MHIP(hipMemcpyPeerAsync(g.dmap + o, g.device, m->g[j].dmap + o, m->g[j].device, cnt, g.st_comm));
                }
            }
        }
    }

    // 3. the exchange step (un-pipelined form): ONE in-place all-gather of equal-sized u8 bands
    if (sub_bands == 1) {
        if (!copy_gather) {
            MNCCL(m, m->rccl.GroupStart());
            for (int i = 0; i < n; ++i) {
                Gpu &g = m->g[i];
                const int r = m->rccl.AllGather(g.dmap + (size_t)i * per * cols, g.dmap, (size_t)per * cols, NCCL_UINT8, g.comm, g.st);
                if (r != 0) { (void)m->rccl.GroupEnd(); snprintf(g_multi_err, g_multi_err_size, "ncclAllGather: %s", m->rccl.GetErrorString(r)); return SADBM_E_RCCL; }
            }
            MNCCL(m, m->rccl.GroupEnd());
        } else {
            for (int i = 0; i < n; ++i) {
                Gpu &g = m->g[i];
                MHIP(hipSetDevice(g.device));
                for (int j = 0; j < n; ++j) {
                    if (j == i) continue;
                    MHIP(hipStreamWaitEvent(g.st, m->g[j].ev_band, 0));
                    const size_t o = (size_t)j * per * cols;
                    MHIP(hipMemcpyPeerAsync(g.dmap + o, g.device, m->g[j].dmap + o, m->g[j].device, (size_t)per * cols, g.st));
                }
            }
        }
    } else {
        for (int i = 0; i < n; ++i) {                         // compute streams rejoin their gather streams
            Gpu &g = m->g[i];
            MHIP(hipSetDevice(g.device));
            MHIP(hipEventRecord(g.ev_gather, g.st_comm));
            MHIP(hipStreamWaitEvent(g.st, g.ev_gather, 0));
        }
    }

    // 4. result: the first GPU's copy of the map -> host (optional); then wait for every GPU
    Gpu &g0 = m->g[0];
    if (out) {
        MHIP(hipSetDevice(g0.device));
        if ((rc = grow_host(&g0.hout, &g0.out_cap, (size_t)rows * cols))) return rc;
        MHIP(hipMemcpyAsync(g0.hout, g0.dmap, (size_t)rows * cols, hipMemcpyDeviceToHost, g0.st));
    }
    for (int i = 0; i < n; ++i) {
        MHIP(hipSetDevice(m->g[i].device));
        MHIP(hipStreamSynchronize(m->g[i].st));
    }
    if (out) copy_rows(out->data, (size_t)out->step, g0.hout, cols, (size_t)cols, rows);
    m->map_rows = rows; m->map_cols = cols;
    return SADBM_OK;
}

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
        int rc = check_pair(&lefts[p], &rights[p], win, max_disp);
        if (rc) return rc;
        if (lefts[p].rows != rows || lefts[p].cols != cols || lefts[p].channels != ch) return SADBM_E_SIZE;   // one geometry per batch
        if (outs[p].rows != rows || outs[p].cols != cols || outs[p].step < cols) return SADBM_E_SIZE;
        if (rows > 0 && cols > 0 && !outs[p].data) return SADBM_E_NULL;
    }
    if (rows == 0 || cols == 0) return SADBM_OK;
    sadbm_detail::DeviceKeeper keep;
    std::lock_guard<std::mutex> guard(m->mu);
    const size_t pin = ((size_t)cols + 255) & ~(size_t)255, pout = pin;   // staged rows hold channel 0 only
    // pair p -> GPU p % n, slot p / n of that GPU's staging; one batched launch sequence per GPU, no collective
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
