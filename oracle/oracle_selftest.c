/* oracle_selftest.c -- runs the CPU restatements (sad_oracle.c, filter_oracle.c) on small cases as one executable, so
 * that `make -C oracle asan` can put them under AddressSanitizer + UBSan (TEST INFRASTRUCTURE ONLY).
 * Cases: the three formulations of getDisparity (literal / literal_mt / box; SAD.cpp:24-61) agree on exact-size heap
 * buffers for ordinary, ragged and degenerate shapes (H <= win, W <= win, W < D, 1x1), 1/3/4 channels and padded
 * strides; bands concatenate to the full frame; known answers K1 (identical images -> 0) and K2 (constants -> j+1
 * ramp); the Gaussian kernel, filter2D, highpass and blend run on odd shapes.  Exit 0 = all agreed and no report. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int sad_oracle_literal(const uint8_t *, int64_t, const uint8_t *, int64_t, int, int, int, int, int, uint32_t, uint8_t *, int64_t);
int sad_oracle_literal_mt(const uint8_t *, int64_t, const uint8_t *, int64_t, int, int, int, int, int, uint32_t, uint8_t *, int64_t, int);
int sad_oracle_box(const uint8_t *, int64_t, const uint8_t *, int64_t, int, int, int, int, int, uint32_t, uint8_t *, int64_t, int);
int sad_oracle_band(const uint8_t *, int64_t, const uint8_t *, int64_t, int, int, int, int, int, int, int, int, uint32_t, uint8_t *, int64_t);
int sad_oracle_rds(uint64_t, int, int, int, int, int, uint8_t *, uint8_t *);
int filter_oracle_gaussian_kernel(int, int, double *);
int filter_oracle_filter2d(const uint8_t *, int64_t, int, int, int, const double *, int, int, uint8_t *, int64_t);
int filter_oracle_highpass(const uint8_t *, int64_t, const uint8_t *, int64_t, int, int, int, uint8_t *, int64_t);
int filter_oracle_blend(const uint8_t *, int64_t, const uint8_t *, int64_t, int, int, int, double, uint8_t *, int64_t);

static uint32_t rng_state = 12345;
static uint8_t rnd(void) { rng_state = rng_state * 1664525u + 1013904223u; return (uint8_t)(rng_state >> 24); }
static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { ++fails; printf("FAIL %s:%d ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static void sad_case(int H, int W, int ch, int win, int D, int pad)
{
    const int64_t step = (int64_t)W * ch + pad;
    const size_t n = (size_t)(H > 0 ? H : 1) * (size_t)(step > 0 ? step : 1);
    uint8_t *L = malloc(n), *R = malloc(n);
    uint8_t *a = malloc((size_t)H * W + 1), *b = malloc((size_t)H * W + 1), *c = malloc((size_t)H * W + 1);
    for (size_t i = 0; i < n; ++i) { L[i] = rnd(); R[i] = (i % 7 == 0) ? rnd() : L[(i + 3 * ch) % n]; }
    const int ra = sad_oracle_literal(L, step, R, step, H, W, ch, win, D, 1, a, W);
    const int rb = sad_oracle_literal_mt(L, step, R, step, H, W, ch, win, D, 1, b, W, 3);
    const int rc = sad_oracle_box(L, step, R, step, H, W, ch, win, D, 1, c, W, 2);
    CHECK(ra == 0 && rb == 0 && rc == 0, "rc %d %d %d (H%d W%d ch%d w%d D%d)", ra, rb, rc, H, W, ch, win, D);
    CHECK(!memcmp(a, b, (size_t)H * W) && !memcmp(a, c, (size_t)H * W), "formulations differ H%d W%d ch%d w%d D%d", H, W, ch, win, D);
    /* bands with a downward halo concatenate to the frame */
    for (int parts = 2; parts <= 3 && H > 0; ++parts) {
        const int per = (H + parts - 1) / parts;
        for (int p = 0; p < parts; ++p) {
            int r0 = p * per, r1 = r0 + per;
            if (r0 > H) r0 = H;
            if (r1 > H) r1 = H;
            if (r1 == r0) continue;
            int i1 = r1 + win - 1;
            if (i1 > H) i1 = H;
            uint8_t *o = malloc((size_t)(r1 - r0) * W + 1);
            const int rr = sad_oracle_band(L + r0 * step, step, R + r0 * step, step, H, r0, i1 - r0, r1 - r0, W, ch, win, D, 1, o, W);
            CHECK(rr == 0 && !memcmp(o, a + (size_t)r0 * W, (size_t)(r1 - r0) * W), "band %d/%d differs H%d W%d w%d D%d", p, parts, H, W, win, D);
            free(o);
        }
    }
    free(L); free(R); free(a); free(b); free(c);
}

int main(void)
{
    const int shapes[][2] = {{23, 41}, {9, 300}, {30, 9}, {8, 8}, {7, 50}, {50, 7}, {1, 1}, {33, 129}, {16, 17}};
    const int params[][2] = {{7, 30}, {3, 64}, {9, 256}, {15, 128}, {1, 1}, {21, 40}};
    for (unsigned s = 0; s < sizeof(shapes) / sizeof(shapes[0]); ++s)
        for (unsigned p = 0; p < sizeof(params) / sizeof(params[0]); ++p)
            for (int ch = 1; ch <= 4; ch += (ch == 1 ? 2 : 1))
                sad_case(shapes[s][0], shapes[s][1], ch, params[p][0], params[p][1], (s + p) % 2 ? 5 : 0);
    /* K1 / K2 of SURVEY section 4 */
    {
        enum { H = 20, W = 60 };
        uint8_t *L = malloc(H * W), *R = malloc(H * W), *o = malloc(H * W);
        for (int i = 0; i < H * W; ++i) L[i] = rnd();
        sad_oracle_literal(L, W, L, W, H, W, 1, 7, 30, 1, o, W);
        for (int i = 0; i < H * W; ++i) CHECK(o[i] == 0, "K1 at %d", i);
        memset(L, 10, H * W); memset(R, 200, H * W);
        sad_oracle_box(L, W, R, W, H, W, 1, 5, 16, 1, o, W, 1);
        for (int y = 0; y < H - 5; ++y)
            for (int x = 0; x < W - 5; ++x) CHECK(o[y * W + x] == (x < 15 ? x + 1 : 0), "K2 at %d,%d: %d", y, x, o[y * W + x]);
        free(L); free(R); free(o);
    }
    {
        enum { W = 64, H = 48 };
        uint8_t *L = malloc(W * H), *R = malloc(W * H);
        CHECK(sad_oracle_rds(1, W, H, 2, 9, 0, L, R) == 0, "rds");
        CHECK(sad_oracle_rds(1, W, H, 2, 9, 1, L, R) == 0, "rds binary");
        free(L); free(R);
    }
    /* filters on odd shapes: every read of the reflect-101 border stays inside the heap blocks */
    for (int ch = 1; ch <= 4; ch += (ch == 1 ? 2 : 1)) {
        const int dims[][2] = {{1, 1}, {2, 37}, {19, 23}, {40, 3}};
        for (unsigned d = 0; d < 4; ++d) {
            const int H = dims[d][0], W = dims[d][1];
            const int64_t step = (int64_t)W * ch;
            uint8_t *s = malloc((size_t)H * step), *lo = malloc((size_t)H * step), *hi = malloc((size_t)H * step), *bl = malloc((size_t)H * step);
            for (int64_t i = 0; i < H * step; ++i) s[i] = rnd();
            const int sizes[][2] = {{35, 3}, {31, 10}, {5, 1}, {1, 1}};
            for (unsigned k = 0; k < 4; ++k) {
                const int sz = sizes[k][0];
                double *kern = malloc(sizeof(double) * sz * sz);
                CHECK(filter_oracle_gaussian_kernel(sz, sizes[k][1], kern) == 0, "kernel");
                CHECK(filter_oracle_filter2d(s, step, H, W, ch, kern, sz, sz, lo, step) == 0, "filter2d");
                free(kern);
            }
            CHECK(filter_oracle_highpass(s, step, lo, step, H, (int)step, 12, hi, step) == 0, "highpass");
            if (ch >= 3) CHECK(filter_oracle_blend(lo, step, hi, step, H, W, ch, 0.6, bl, step) == 0, "blend");
            free(s); free(lo); free(hi); free(bl);
        }
    }
    printf(fails ? "oracle selftest: %d FAILED\n" : "oracle selftest: ok%.0d\n", fails);
    return fails ? 1 : 0;
}
