/*
 * sadbm.h -- C ABI of libsadbm: SAD stereo block matching on AMD Instinct MI355X (gfx950).
 *
 * Drop-in boundary for the one hot path of the reference project: the free function
 *     Mat getDisparity(Mat &left, Mat &right)          /root/reference/SAD.cpp:24-61
 * with its two file-scope parameters  int winSize = 7  (SAD.cpp:9)  and  int DSR = 30  (SAD.cpp:11)
 * and its helper  int sub_kernel(Mat&, Mat&)           (SAD.cpp:13-22).
 *
 * The reference has no FFI of its own; a cv::Mat is (data, rows, cols, channels(), step[0]) and that
 * is exactly what sadbm_image carries, so a maintainer binds this header from SAD.cpp with a
 * six-line adapter (see INTEGRATION.md and include/sadbm.hpp).
 *
 * Semantics reproduced bit-exactly:
 *   cost uses channel 0 only (SAD.cpp:19-20); window anchored top-left (SAD.cpp:39,47);
 *   output rows/cols >= rows-win / cols-win stay 0 (SAD.cpp:31,34,36); disparities with j-k<0 score
 *   0 (SAD.cpp:40,44-45); first minimum wins (SAD.cpp:54); value = (uchar)(loc*16) (SAD.cpp:57).
 *
 * Plain pointers and sizes only; no exceptions cross this boundary; all functions return 0 or a
 * negative SADBM_E_* code.  There is NO CPU fallback: without a HIP device every compute entry
 * point fails with SADBM_E_HIP.
 */
#ifndef SADBM_H
#define SADBM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SADBM_VERSION 200

/* flags */
#define SADBM_RAW_LOC        1u   /* write loc (0..max_disp-1) instead of (loc*16)&255; a byte, so it needs max_disp<=256 */
#define SADBM_FORCE_GENERIC  2u   /* use the brute-force HIP kernel (any win); for cross-checks */
#define SADBM_KERNEL_GLOBAL  4u   /* use the matcher variant that reads operands through L1 instead of LDS (A/B) */
#define SADBM_KERNEL_STRIP   8u   /* force the strip matcher (16 columns x 4 disparities per thread; 4 <= win <= 25) */
#define SADBM_KERNEL_QUAD   16u   /* force the quad matcher (4 columns x 8/16 disparities per thread) */

/* error codes */
#define SADBM_OK        0
#define SADBM_E_NULL   -1   /* null pointer argument */
#define SADBM_E_TYPE   -2   /* channels not in {1,3,4} or left/right channel counts differ
                               (the reference would throw inside cv::absdiff, SAD.cpp:17) */
#define SADBM_E_SIZE   -3   /* right smaller than left (the reference would throw in Mat::operator()(Rect),
                               SAD.cpp:47), negative sizes, or a pitch smaller than a row */
#define SADBM_E_PARAM  -4   /* win<1 or max_disp<1 (the reference would loop over nothing / throw); SADBM_RAW_LOC with
                               max_disp>256 (loc does not fit the output byte); win>4096 on a frame larger than that.
                               winSize and DSR are plain ints in the reference (SAD.cpp:9,11) and it accepts any value:
                               so does this library, in the reference's encoding -- DSR>256 and windows of 256..4096
                               columns run slower matchers */
#define SADBM_E_HIP    -5   /* HIP runtime error (sadbm_last_hip_error() has the text) */
#define SADBM_E_ALLOC  -6   /* device memory allocation failed */
#define SADBM_E_BUSY   -7   /* pipeline slot still in flight (wait on its ticket first) / unknown ticket */
#define SADBM_E_LIMIT  -8   /* sadbm_compute_device_static: the frame needs more scratch than the static buffer holds */
#define SADBM_E_RCCL   -9   /* sadbm_multi_*: librccl could not be loaded, or an RCCL call failed
                               (sadbm_last_hip_error() has the text) */

/* Host-memory image view; maps 1:1 onto cv::Mat {data, rows, cols, channels(), step[0]} (CV_8U only).
 * Replaces the `Mat &left, Mat &right` arguments of getDisparity (SAD.cpp:24). */
typedef struct {
    const uint8_t *data;
    int32_t rows, cols, channels;
    int64_t step;                 /* bytes between rows */
} sadbm_image;

/* Caller-allocated CV_8UC1 output; replaces `Mat disparity(row, col, CV_8UC1, Scalar(0))` (SAD.cpp:31). */
typedef struct {
    uint8_t *data;
    int32_t rows, cols;
    int64_t step;
} sadbm_plane;

typedef struct sadbm_ctx sadbm_ctx;   /* per (host thread, device, stream) state: scratch planes + a stream.  Calls on one
                                         context are serialised; a call on another stream than the previous one first waits
                                         for that one's work (the scratch is shared). */

int         sadbm_version(void);
const char *sadbm_strerror(int code);
const char *sadbm_last_hip_error(void);              /* thread-local text of the last SADBM_E_HIP */

/* Context life cycle.  A context owns device scratch (padded single-channel planes) that grows on
 * demand; use one context per stream.  No reference counterpart (the reference keeps its state in
 * file-scope globals, SAD.cpp:9-11, and is not re-entrant). */
int  sadbm_create(int device, sadbm_ctx **out);
void sadbm_destroy(sadbm_ctx *ctx);
int  sadbm_ctx_device(const sadbm_ctx *ctx);            /* the device the context was created on, -1 for NULL */

/* getDisparity(left, right) with explicit winSize / DSR (SAD.cpp:24, :9, :11).  Host buffers in, host
 * buffer out, synchronous.  Every byte of `out` (rows x cols of `left`) is written.  Uses a lazily
 * created thread-local context on the calling thread's current device (hipSetDevice; device 0 by default).
 * No entry point of this library changes the calling thread's current device. */
int sadbm_get_disparity(const sadbm_image *left, const sadbm_image *right, int win, int max_disp,
                        uint32_t flags, sadbm_plane *out);

/* Releases the calling thread's lazily created context now (it is also released when the thread exits). */
void sadbm_thread_release(void);

/* Same, on an explicit context (any device). */
int sadbm_get_disparity_ctx(sadbm_ctx *ctx, const sadbm_image *left, const sadbm_image *right,
                            int win, int max_disp, uint32_t flags, sadbm_plane *out);

/* Device-resident getDisparity: dL/dR/dOut are device pointers on ctx's device; asynchronous on
 * `stream` (a hipStream_t passed as void*; NULL = the device's default stream, as in HIP).  channels in {1,3,4}
 * (channel 0 is used, SAD.cpp:19-20).  dOut must not overlap dL or dR (the reference writes a fresh Mat, SAD.cpp:31): the
 * matcher may read the inputs in place while it writes the output.  Only the bytes of the images themselves are read
 * (rows * pitch is not assumed readable beyond the last row's last column). */
int sadbm_compute_device(sadbm_ctx *ctx, const uint8_t *dL, int64_t pitchL, const uint8_t *dR,
                         int64_t pitchR, int rows, int cols, int channels, int win, int max_disp,
                         uint32_t flags, uint8_t *dOut, int64_t out_pitch, void *stream);

/* Row band of a larger frame (single-frame multi-GPU split).  The device
 * buffers hold input rows [row0, row0+rows_in) of a frame that has `frame_rows` rows in total; output
 * rows [row0, row0+rows_out) are written to dOut (row 0 of dOut = frame row row0).  The loop bound of
 * SAD.cpp:34 is evaluated on the frame row index, so rows >= frame_rows-win are zero.  Requires
 * rows_in >= min(rows_out + win - 1, frame_rows - row0). */
int sadbm_compute_band_device(sadbm_ctx *ctx, const uint8_t *dL, int64_t pitchL, const uint8_t *dR,
                              int64_t pitchR, int frame_rows, int row0, int rows_in, int rows_out,
                              int cols, int channels, int win, int max_disp, uint32_t flags,
                              uint8_t *dOut, int64_t out_pitch, void *stream);

/* N equally sized pairs (batch data parallelism): pair n lives at
 * dL + n*pair_stride_in, dR + n*pair_stride_in, dOut + n*pair_stride_out.  One launch sequence for
 * the whole batch. */
int sadbm_compute_batch_device(sadbm_ctx *ctx, const uint8_t *dL, const uint8_t *dR, int64_t pitch_in,
                               int64_t pair_stride_in, int n_pairs, int rows, int cols, int channels,
                               int win, int max_disp, uint32_t flags, uint8_t *dOut,
                               int64_t out_pitch, int64_t pair_stride_out, void *stream);

/* Streaming pipeline: `depth` slots, each with pinned host buffers, device buffers,
 * its own stream and scratch, so H2D copy, matcher and D2H copy of consecutive frames overlap.  The reference
 * processes one pair per program run (SAD.cpp:63-82) and has no counterpart; this is the caller loop a video
 * user of getDisparity() would write.  All frames of a pipeline share rows/cols/channels/win/max_disp/flags.
 *   slot = ticket % depth.  A slot may be refilled only after sadbm_pipeline_wait() on its previous ticket.
 *   sadbm_pipeline_inputs : pinned input buffers of the next slot (fill them in place: zero-copy submit)
 *   sadbm_pipeline_submit : enqueue H2D + getDisparity + D2H for that slot, returns its ticket at once;
 *                           left/right may be NULL (buffers already filled) or images to copy in first
 *   sadbm_pipeline_wait   : block until the ticket is done; *result points at the pinned rows x cols output
 *                           (valid until the slot is submitted again); out, if not NULL, also receives a copy */
typedef struct sadbm_pipeline sadbm_pipeline;
int  sadbm_pipeline_create(int device, int depth, int rows, int cols, int channels, int win, int max_disp,
                           uint32_t flags, sadbm_pipeline **out);
void sadbm_pipeline_destroy(sadbm_pipeline *p);
int  sadbm_pipeline_inputs(sadbm_pipeline *p, uint8_t **left, uint8_t **right, int64_t *step);
int  sadbm_pipeline_submit(sadbm_pipeline *p, const sadbm_image *left, const sadbm_image *right, int64_t *ticket);
int  sadbm_pipeline_wait(sadbm_pipeline *p, int64_t ticket, const uint8_t **result, int64_t *result_step,
                         sadbm_plane *out);

/* Single-process multi-GPU (no reference counterpart: SAD.cpp is single-threaded).  A
 * sadbm_multi owns, per GPU: a context, a stream, pinned and device staging and -- unless SADBM_MULTI_COPY_GATHER is
 * given -- one rank of a single-process RCCL communicator (ncclCommInitAll; librccl is loaded at run time, so
 * libsadbm.so itself does not link it).  `devices` may be NULL (= 0 .. ngpu-1).
 *   sadbm_compute_multi       ONE frame split into ngpu row bands with a downward halo of win-1 rows (the window is
 *                             anchored top-left, SAD.cpp:39); every GPU computes its band into its slot of its own copy
 *                             of the full map and ONE all-gather of u8 bands (ncclAllGather, in place) leaves the full
 *                             rows x cols map on every GPU.  sub_bands > 1 splits each band further and gathers the
 *                             finished sub-band (grouped ncclBroadcast) while the next one is computed.  Band i is
 *                             staged (channel 0 gathered), uploaded and launched before band i+1 is staged.  `out` (may
 *                             be NULL) receives every band from the GPU that computed it, over that GPU's own link;
 *                             sadbm_multi_device_map gives each GPU's copy of the whole map.
 *   sadbm_compute_multi_device the same for a frame that is ALREADY in the memory of one of the handle's GPUs (index
 *                             src_index of `devices`; what SURVEY 8e calls "device-resident on one GPU"): that GPU reads
 *                             its band in place, every other GPU pulls its band + halo rows with one peer copy per image on
 *                             its own stream, then bands and gather as above.  dOut (may be NULL) = a rows x cols buffer on
 *                             the source GPU that receives the map.  ASYNCHRONOUS like sadbm_compute_device: the inputs are
 *                             read once `stream` (a stream of the source GPU, NULL = its default stream) reaches the call,
 *                             and `stream` waits for every GPU before it goes on -- work queued on it afterwards sees dOut
 *                             and every GPU's map; the host does not wait.  No reference counterpart (SAD.cpp:24,76: one
 *                             caller, one thread, host Mats).
 *   sadbm_compute_multi_batch n_pairs independent pairs, pair n on GPU n % ngpu, no collective (batch data
 *                             parallelism).
 * Calls on one sadbm_multi are serialised (they share its staging buffers); none changes the caller's current device.
 * SADBM_MULTI_COPY_GATHER: assemble the map with peer copies (hipMemcpyPeerAsync) instead of RCCL -- the same
 * interface without the collective; it also allows one device to be listed several times (rehearsal of the
 * partition on a single GPU). */
#define SADBM_MULTI_COPY_GATHER 1u
/* SADBM_MULTI_FORCE_PEER_COPY: sadbm_compute_multi_device treats every list entry other than src_index as a remote GPU even
 * when it names the source device again, i.e. copies its band -- rehearsal of the scatter on a single GPU. */
#define SADBM_MULTI_FORCE_PEER_COPY 2u
typedef struct sadbm_multi sadbm_multi;
int  sadbm_multi_create(int ngpu, const int *devices, uint32_t multi_flags, sadbm_multi **out);
void sadbm_multi_destroy(sadbm_multi *m);
int  sadbm_compute_multi(sadbm_multi *m, const sadbm_image *left, const sadbm_image *right, int win, int max_disp,
                         uint32_t flags, int sub_bands, sadbm_plane *out);
int  sadbm_compute_multi_device(sadbm_multi *m, int src_index, const uint8_t *dL, int64_t pitchL, const uint8_t *dR, int64_t pitchR,
                                int rows, int cols, int channels, int win, int max_disp, uint32_t flags, int sub_bands,
                                uint8_t *dOut, int64_t out_pitch, void *stream);
int  sadbm_multi_device_map(sadbm_multi *m, int index, const uint8_t **dmap, int64_t *pitch);
int  sadbm_multi_device_id(const sadbm_multi *m, int index);     /* the HIP device behind list entry `index`, -1 if out of range */
int  sadbm_compute_multi_batch(sadbm_multi *m, const sadbm_image *lefts, const sadbm_image *rights, int n_pairs, int win,
                               int max_disp, uint32_t flags, sadbm_plane *outs);
/* Row partition used by sadbm_compute_multi (pure host arithmetic): band `index` of `parts` covers output rows
 * [*row0, *row1) and needs input rows [*row0, *in1); rows per band = ceil(rows / parts). */
int  sadbm_band_plan(int rows, int parts, int win, int index, int *row0, int *row1, int *in1);

/* Device-resident getDisparity WITHOUT any allocation.  No counterpart in the reference; it is for callers that cannot
 * allocate: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same arguments and
 * results as sadbm_compute_device, but no context: the scratch (the two padded single-channel planes the matchers
 * read) is a static device array of SADBM_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the
 * process ends.  Frames whose planes need more return SADBM_E_LIMIT (one 3840x2160 pair fits).  Calls no hipMalloc /
 * hipFree / hipStreamCreate / hipEventCreate; the first call on a device resolves the array's address.  All calls on
 * one GPU share that array: calls that overlap on one GPU (two streams, two threads) race -- serialise them. */
#define SADBM_STATIC_SCRATCH_BYTES (19u << 20)
int sadbm_compute_device_static(int device, const uint8_t *dL, int64_t pitchL, const uint8_t *dR, int64_t pitchR,
                                int rows, int cols, int channels, int win, int max_disp, uint32_t flags,
                                uint8_t *dOut, int64_t out_pitch, void *stream);

/* Introspection for benchmarks (no reference counterpart): name of the matcher kernel the last call
 * on this ctx launched, and HIP-event timing of the matcher launches between profile_begin and
 * profile_end (events are recorded on the launch stream, nothing synchronises until profile_end). */
const char *sadbm_last_kernel_name(const sadbm_ctx *ctx);
int sadbm_profile_begin(sadbm_ctx *ctx, int max_launches);
int sadbm_profile_end(sadbm_ctx *ctx, int *n_launches, float *matcher_ms_total, float *prep_ms_total);
/* What a call on n_pairs whole frames (contiguous, device-resident) would launch, without touching a GPU (no reference
 * counterpart; for integrators sizing their batches and for the CPU-only tests of the launch policy): one line of
 * key=value fields -- matcher family, kernel, threads per block, frame groups (FG frames side by side, Wp columns apart),
 * 256-column blocks per group row (bx), rows per block (TY), bands (nby), blocks, half-height tail units per XCD, dynamic
 * LDS bytes, scratch bytes the context would hold, whether the plane-preparation pass runs.  cus = compute units of the
 * device (<= 0: 256, an MI355X).  Returns SADBM_OK or the SADBM_E_* code the call itself would return for these arguments. */
int sadbm_plan_describe(int rows, int cols, int channels, int n_pairs, int win, int max_disp, uint32_t flags, int cus,
                        char *text, int text_len);
/* The tuning knobs of INTEGRATION.md are read from the environment once, at the first call; this re-reads them. */
void sadbm_reload_env(void);

#ifdef __cplusplus
}
#endif
#endif /* SADBM_H */
