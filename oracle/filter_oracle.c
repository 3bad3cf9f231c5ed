/*
 * filter_oracle.c -- CPU restatement of the Gaussian low/high-pass programs of the reference
 * (SURVEY.md section 8f, row f4).  TEST INFRASTRUCTURE ONLY: imported by tests/ and the benchmark's
 * CPU leg, never by the product.
 *
 * Follows
 *   GetgaussianKernel()            /root/reference/gaus_filter.cpp:15-40, hybrid_image.cpp:15-40
 *   filter2D(src, dst, depth, kern) gaus_filter.cpp:60, hybrid_image.cpp:124,129
 *   addWeighted(src,1,low,-1,0,high) gaus_filter.cpp:62, hybrid_image.cpp:133
 *   src_image2 += Scalar::all(12)  hybrid_image.cpp:131
 *   add(mat1, mat2, alpha)         hybrid_image.cpp:78-100
 *
 * PARITY UNPINNED: the arithmetic of filter2D lives in OpenCV 4.1.0 (README.md:38), which is neither
 * vendored nor installed, and the reference holds no expected outputs.  For kernels this large OpenCV's
 * filter2D switches to a DFT-based float path, whose last-bit rounding is not reproducible without the
 * library; this restatement is the documented definition instead -- correlation (no kernel flip) with the
 * anchor at the kernel centre, BORDER_REFLECT_101 borders, double accumulation in row-major tap order,
 * saturate_cast<uchar>(cvRound(sum)) -- and the HIP kernel is held bit-exact to IT.  Expect +-1 LSB against
 * a real OpenCV build on a small fraction of pixels.
 */
#include <math.h>
#include <stdint.h>

/* gaus_filter.cpp:22-38.  Note the reference's integer arithmetic: -(di*di+dj*dj) / (2*sigma*sigma) is an
 * int division (truncating toward zero) BEFORE exp(); sigma and size are ints (#define). */
int filter_oracle_gaussian_kernel(int size, int sigma, double *out)
{
    if (size < 1 || sigma < 1 || !out) return -1;
    const int center = size / 2;
    double sum = 0;
    for (int i = 0; i < size; ++i)
        for (int j = 0; j < size; ++j) {
            const int q = -((i - center) * (i - center) + (j - center) * (j - center)) / (2 * sigma * sigma);
            out[i * size + j] = exp((double)q);
            sum += out[i * size + j];
        }
    for (int i = 0; i < size * size; ++i) out[i] /= sum;
    return 0;
}

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

static uint8_t sat_round(double v)
{
    const double r = nearbyint(v);          /* cvRound: round half to even */
    return r < 0 ? 0 : r > 255 ? 255 : (uint8_t)r;
}

/* filter2D on CV_8UC{ch}: dst(y,x,c) = sat(round(sum_{i,j} k[i][j] * src(y+i-ay, x+j-ax, c))), anchor = centre. */
int filter_oracle_filter2d(const uint8_t *src, int64_t step, int rows, int cols, int ch, const double *kern, int krows,
                           int kcols, uint8_t *dst, int64_t dstep)
{
    if (!src || !kern || !dst || rows < 1 || cols < 1 || ch < 1 || krows < 1 || kcols < 1) return -1;
    const int ay = krows / 2, ax = kcols / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
            for (int c = 0; c < ch; ++c) {
                double acc = 0;
                for (int i = 0; i < krows; ++i) {
                    const uint8_t *row = src + (int64_t)reflect101(y + i - ay, rows) * step;
                    for (int j = 0; j < kcols; ++j) {
                        const double prod = kern[i * kcols + j] * (double)row[(int64_t)reflect101(x + j - ax, cols) * ch + c];
                        acc = acc + prod;           /* separate multiply and add: no FMA contraction (-ffp-contract=off) */
                    }
                }
                dst[(int64_t)y * dstep + (int64_t)x * ch + c] = sat_round(acc);
            }
    return 0;
}

/* addWeighted(a, 1, b, -1, 0): saturate_cast<uchar>(a - b) on 8-bit data = max(a-b, 0).  `add` = the constant
 * the hybrid program adds to the source first (hybrid_image.cpp:131: += Scalar::all(12), saturating). */
int filter_oracle_highpass(const uint8_t *a, int64_t astep, const uint8_t *low, int64_t lstep, int rows, int rowbytes, int add,
                           uint8_t *dst, int64_t dstep)
{
    if (!a || !low || !dst) return -1;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < rowbytes; ++x) {
            int s = a[(int64_t)y * astep + x] + add;
            if (s > 255) s = 255;
            int d = s - low[(int64_t)y * lstep + x];
            dst[(int64_t)y * dstep + x] = (uint8_t)(d < 0 ? 0 : d);
        }
    return 0;
}

/* add(mat1, mat2, alpha), hybrid_image.cpp:78-100: interior pixels only (1..rows-2, 1..cols-2), first 3 channels,
 * temp = (int)(alpha*a + (1-alpha)*b) truncated toward zero, clamped; everything else keeps mat1. */
int filter_oracle_blend(const uint8_t *m1, int64_t step1, const uint8_t *m2, int64_t step2, int rows, int cols, int ch, double alpha,
                        uint8_t *dst, int64_t dstep)
{
    if (!m1 || !m2 || !dst || ch < 3) return -1;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols * ch; ++x) dst[(int64_t)y * dstep + x] = m1[(int64_t)y * step1 + x];
    for (int k = 0; k < 3; ++k)
        for (int i = 1; i < rows - 1; ++i)
            for (int j = 1; j < cols - 1; ++j) {
                int temp = (int)(alpha * m1[(int64_t)i * step1 + (int64_t)j * ch + k] + (1.0 - alpha) * m2[(int64_t)i * step2 + (int64_t)j * ch + k]);
                if (temp > 255) temp = 255; else if (temp < 0) temp = 0;
                dst[(int64_t)i * dstep + (int64_t)j * ch + k] = (uint8_t)temp;
            }
    return 0;
}
