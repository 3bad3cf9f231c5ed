/*
 * sadbm_filters.h -- C ABI of the Gaussian low/high-pass path of the reference's two filter programs
 * (SURVEY.md section 8f, row f4), exported by the same libsadbm.so.
 *
 * Replaces, with explicit arguments instead of #defines and file-scope arrays:
 *   GetgaussianKernel()                          /root/reference/gaus_filter.cpp:15-40, hybrid_image.cpp:15-40
 *   filter2D(src, dst, src.depth(), kern)        gaus_filter.cpp:60, hybrid_image.cpp:124,129
 *   addWeighted(src, 1, low, -1, 0, high)        gaus_filter.cpp:62, hybrid_image.cpp:133
 *   src_image2 += Scalar::all(12)                hybrid_image.cpp:131
 *   add(mat1, mat2, alpha)                       hybrid_image.cpp:78-100
 *
 * filter2D is defined here as: correlation (no kernel flip), anchor at the kernel centre (krows/2, kcols/2),
 * BORDER_REFLECT_101, float64 accumulation in row-major tap order without FMA contraction,
 * saturate_cast<uchar>(cvRound(sum)) -- the definition of oracle/filter_oracle.c, to which the HIP kernel is
 * bit-exact.  The reference's own arithmetic (OpenCV 4.1.0's DFT-based float path for kernels this large) is not
 * available here: parity with a real OpenCV build is unpinned and expected to differ by +-1 LSB on few pixels.
 * All pointers of the *_device functions are device pointers on ctx's device; calls are asynchronous on `stream`.
 * Return values: 0 or SADBM_E_* (sadbm.h).  No CPU fallback.
 */
#ifndef SADBM_FILTERS_H
#define SADBM_FILTERS_H

#include "sadbm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host: the size x size kernel of GetgaussianKernel(), including its integer-divided exponent
 * exp(-((di*di+dj*dj) / (2*sigma*sigma))) (int division) and the normalisation by the sum. */
int sadbm_gaussian_kernel(int size, int sigma, double *out);

/* filter2D on CV_8UC{1,3,4}; kernel = krows x kcols float64 in HOST memory (copied to the device per call,
 * krows*kcols <= 4096). */
int sadbm_filter2d_device(sadbm_ctx *ctx, const uint8_t *dSrc, int64_t src_pitch, int rows, int cols, int channels,
                          const double *kernel, int krows, int kcols, uint8_t *dDst, int64_t dst_pitch, void *stream);

/* dst = saturate(saturate(src + add) - low): addWeighted(src,1,low,-1,0) with the hybrid program's `+= 12` folded in
 * (add = 0 for gaus_filter.cpp).  rowbytes = cols * channels. */
int sadbm_highpass_device(sadbm_ctx *ctx, const uint8_t *dSrc, int64_t src_pitch, const uint8_t *dLow, int64_t low_pitch,
                          int rows, int rowbytes, int add, uint8_t *dDst, int64_t dst_pitch, void *stream);

/* add(mat1, mat2, alpha) of hybrid_image.cpp:78-100: interior pixels and the first three channels get
 * (int)(alpha*m1 + (1-alpha)*m2) clamped to 0..255, everything else copies mat1.  channels >= 3. */
int sadbm_blend_device(sadbm_ctx *ctx, const uint8_t *dM1, int64_t pitch1, const uint8_t *dM2, int64_t pitch2, int rows,
                       int cols, int channels, double alpha, uint8_t *dDst, int64_t dst_pitch, void *stream);

/* The two filter programs end to end on HOST buffers (device 0; H2D, kernels, D2H, synchronous) -- what their main()
 * functions compute between imread and imshow.  Output buffers are rows x cols x channels of the source, may be NULL
 * where a result is not wanted.
 *   sadbm_gaus_filter_host   : gaus_filter.cpp:57-62   low = filter2D(src, G(size, sigma)); high = addWeighted(src,1,low,-1,0)
 *   sadbm_hybrid_image_host  : hybrid_image.cpp:120-138 low = filter2D(img1, G); tmp = filter2D(img2, G);
 *                              high = addWeighted(img2 + add, 1, tmp, -1, 0); hybrid = add(low, high, alpha)
 * The reference's values: size 35 / sigma 3 (gaus_filter.cpp:10-11); size 31 / sigma 10, add 12, alpha 0.6
 * (hybrid_image.cpp:10-11,131,138). */
int sadbm_gaus_filter_host(const sadbm_image *src, int size, int sigma, uint8_t *low, int64_t low_step, uint8_t *high,
                           int64_t high_step);
int sadbm_hybrid_image_host(const sadbm_image *img1, const sadbm_image *img2, int size, int sigma, int add, double alpha,
                            uint8_t *low, int64_t low_step, uint8_t *high, int64_t high_step, uint8_t *hybrid,
                            int64_t hybrid_step);

#ifdef __cplusplus
}
#endif
#endif /* SADBM_FILTERS_H */
