/*
 * sad_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the shipped library.
 *
 * A CPU restatement of the SAD stereo block matcher in the reference's SAD.cpp, used as the
 * checker for the HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * Nothing under opencv_amd/ may call, link or load this file.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference has no tests, golden outputs or printed
 * numbers for this path, and its arithmetic lives in OpenCV 4.1.0 (absdiff / sum / minMaxLoc),
 * which is neither vendored in /root/reference nor installed here, so SAD.cpp cannot be built.
 * This restatement follows SAD.cpp line by line plus OpenCV's documented semantics of those three
 * calls; it is pinned by hand-derivable known-answer tests and by the restatement-derived
 * SHA-256 vectors recorded in SURVEY.md section 4 (tests/test_oracle.py).
 *
 * Reference lines restated (all /root/reference/SAD.cpp):
 *   sub_kernel      :13-22  absdiff -> sum -> channel 0 -> float -> int
 *   getDisparity    :24-61  loop bounds (:34,:36), ROI anchoring (:39,:47), zero-initialised
 *                           float cost row (:40), j-k>=0 guard (:44-45), first-minimum minMaxLoc
 *                           (:54), uchar = loc*16 wrap (:57), zero-filled output (:31)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SAD_ORACLE_RAW_LOC 1u

/* SAD.cpp:13-22.  absdiff on CV_8U is |a-b| per element per channel; sum() returns per-channel
 * totals as doubles; only Add[0] (channel 0) is kept, stored through a float, returned as int. */
static int oracle_sub_kernel(const uint8_t *kl, int64_t lstep, const uint8_t *kr, int64_t rstep,
                             int win, int ch)
{
    double add0 = 0.0;
    for (int dy = 0; dy < win; ++dy) {
        const uint8_t *pl = kl + (int64_t)dy * lstep;
        const uint8_t *pr = kr + (int64_t)dy * rstep;
        for (int dx = 0; dx < win; ++dx) {
            int a = pl[(int64_t)dx * ch], b = pr[(int64_t)dx * ch];
            add0 += (double)(a > b ? a - b : b - a);
        }
    }
    float sub_Sum = (float)add0;          /* SAD.cpp:20 (global float sub_Sum) */
    return (int)sub_Sum;                  /* SAD.cpp:21 */
}

/* SAD.cpp:24-61, literal.  Rows [row0,row1) of the output only (row0=0,row1=rows for the whole
 * frame); the i < rows - win test always uses the global row index, which is what a row-band
 * split must reproduce. Output rows are written at out + (i-row0)*ostep. */
static void oracle_rows(const uint8_t *L, int64_t lstep, const uint8_t *R, int64_t rstep,
                        int rows, int cols, int ch, int win, int dsr, uint32_t flags,
                        uint8_t *out, int64_t ostep, int row0, int row1)
{
    float *temp = (float *)malloc(sizeof(float) * (size_t)(dsr > 0 ? dsr : 1));
    for (int i = row0; i < row1; ++i) {
        uint8_t *orow = out + (int64_t)(i - row0) * ostep;
        memset(orow, 0, (size_t)cols);                       /* SAD.cpp:31 Scalar(0) */
        if (i >= rows - win) continue;                       /* SAD.cpp:34 */
        for (int j = 0; j < cols - win; ++j) {               /* SAD.cpp:36 */
            const uint8_t *kl = L + (int64_t)i * lstep + (int64_t)j * ch;      /* Rect(j,i,w,w) :39 */
            for (int k = 0; k < dsr; ++k) temp[k] = 0.0f;    /* SAD.cpp:40 */
            for (int k = 0; k < dsr; ++k) {                  /* SAD.cpp:42 */
                int y = j - k;
                if (y >= 0) {                                /* SAD.cpp:44-45 */
                    const uint8_t *kr = R + (int64_t)i * rstep + (int64_t)y * ch;   /* :47 */
                    temp[k] = (float)oracle_sub_kernel(kl, lstep, kr, rstep, win, ch); /* :48 */
                }
            }
            /* minMaxLoc (:54): first minimum in index order (update only on strict <). */
            int loc = 0; float mn = temp[0];
            for (int k = 1; k < dsr; ++k) if (temp[k] < mn) { mn = temp[k]; loc = k; }
            orow[j] = (flags & SAD_ORACLE_RAW_LOC) ? (uint8_t)loc : (uint8_t)(loc * 16);  /* :57 */
        }
    }
    free(temp);
}

static int oracle_check(const uint8_t *L, const uint8_t *R, int rows, int cols, int ch, int win,
                        int dsr, uint32_t flags, const uint8_t *out)
{
    if (!L || !R || !out) return -1;
    if (rows < 0 || cols < 0) return -3;
    if (ch != 1 && ch != 3 && ch != 4) return -2;
    if (win < 1 || dsr < 1) return -4;
    if ((flags & SAD_ORACLE_RAW_LOC) && dsr > 256) return -4;
    return 0;
}

/* Literal restatement, single thread (the reference is single-threaded). */
int sad_oracle_literal(const uint8_t *L, int64_t lstep, const uint8_t *R, int64_t rstep,
                       int rows, int cols, int ch, int win, int dsr, uint32_t flags,
                       uint8_t *out, int64_t ostep)
{
    int e = oracle_check(L, R, rows, cols, ch, win, dsr, flags, out);
    if (e) return e;
    oracle_rows(L, lstep, R, rstep, rows, cols, ch, win, dsr, flags, out, ostep, 0, rows);
    return 0;
}

/* Same, output rows [row0,row1) only (band split check; out has row1-row0 rows). */
int sad_oracle_literal_rows(const uint8_t *L, int64_t lstep, const uint8_t *R, int64_t rstep,
                            int rows, int cols, int ch, int win, int dsr, uint32_t flags,
                            uint8_t *out, int64_t ostep, int row0, int row1)
{
    int e = oracle_check(L, R, rows, cols, ch, win, dsr, flags, out);
    if (e) return e;
    if (row0 < 0 || row1 > rows || row0 > row1) return -3;
    oracle_rows(L, lstep, R, rstep, rows, cols, ch, win, dsr, flags, out, ostep, row0, row1);
    return 0;
}

/* Row band of a larger frame, mirroring sadbm_compute_band_device: the buffers hold input rows
 * [row0, row0+rows_in) of a frame with frame_rows rows; output rows [row0, row0+rows_out) are
 * produced with SAD.cpp:34's bound evaluated on the frame row index. */
int sad_oracle_band(const uint8_t *L, int64_t lstep, const uint8_t *R, int64_t rstep, int frame_rows,
                    int row0, int rows_in, int rows_out, int cols, int ch, int win, int dsr,
                    uint32_t flags, uint8_t *out, int64_t ostep)
{
    int e = oracle_check(L, R, frame_rows, cols, ch, win, dsr, flags, out);
    if (e) return e;
    if (row0 < 0 || rows_in < 0 || rows_out < 0 || row0 + rows_out > frame_rows || row0 + rows_in > frame_rows) return -3;
    int nvalid = frame_rows - win - row0;
    if (nvalid > rows_out) nvalid = rows_out;
    if (nvalid < 0) nvalid = 0;
    if (nvalid > 0 && rows_in < nvalid + win - 1) return -3;
    /* run the literal loop on a virtual image of nvalid+win rows: its own bound i < rows-win then
     * admits exactly the nvalid rows; row nvalid+win-1 is never read (SAD.cpp:34 leaves one spare row). */
    for (int i = 0; i < rows_out; ++i) memset(out + (int64_t)i * ostep, 0, (size_t)cols);
    if (nvalid > 0)
        oracle_rows(L, lstep, R, rstep, nvalid + win, cols, ch, win, dsr, flags, out, ostep, 0, nvalid);
    return 0;
}

/* Literal restatement with the output rows spread over `threads` OpenMP threads. */
int sad_oracle_literal_mt(const uint8_t *L, int64_t lstep, const uint8_t *R, int64_t rstep,
                          int rows, int cols, int ch, int win, int dsr, uint32_t flags,
                          uint8_t *out, int64_t ostep, int threads)
{
    int e = oracle_check(L, R, rows, cols, ch, win, dsr, flags, out);
    if (e) return e;
    if (threads < 1) threads = 1;
    int chunk = 4;
    int nchunks = (rows + chunk - 1) / chunk;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int c = 0; c < nchunks; ++c) {
        int r0 = c * chunk, r1 = r0 + chunk > rows ? rows : r0 + chunk;
        oracle_rows(L, lstep, R, rstep, rows, cols, ch, win, dsr, flags,
                    out + (int64_t)r0 * ostep, ostep, r0, r1);
    }
    return 0;
}

/*
 * Independent second formulation (used to check full-size frames in seconds): per disparity k,
 * column sums of |L-R| over `win` rows slid down the image, then a sliding horizontal sum; the
 * first-minimum rule is applied as a strict-< update in ascending k; disparities with j-k<0 keep
 * cost 0 exactly as the zero-initialised row in SAD.cpp:40 does.  Integer arithmetic throughout
 * (the float round trip of SAD.cpp:20-21 is exact below 2^24, i.e. for win <= 256).
 */
int sad_oracle_box(const uint8_t *L, int64_t lstep, const uint8_t *R, int64_t rstep,
                   int rows, int cols, int ch, int win, int dsr, uint32_t flags,
                   uint8_t *out, int64_t ostep, int threads)
{
    int e = oracle_check(L, R, rows, cols, ch, win, dsr, flags, out);
    if (e) return e;
    if (threads < 1) threads = 1;
    for (int i = 0; i < rows; ++i) memset(out + (int64_t)i * ostep, 0, (size_t)cols);
    int orows = rows - win, ocols = cols - win;
    if (orows <= 0 || ocols <= 0) return 0;
    int band = 32;
    int nb = (orows + band - 1) / band;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int b = 0; b < nb; ++b) {
        int i0 = b * band, i1 = i0 + band > orows ? orows : i0 + band;
        int32_t *col = (int32_t *)malloc(sizeof(int32_t) * (size_t)cols);
        int32_t *best = (int32_t *)malloc(sizeof(int32_t) * (size_t)ocols * (size_t)(i1 - i0));
        uint8_t *bloc = (uint8_t *)malloc((size_t)ocols * (size_t)(i1 - i0) * 2);
        uint16_t *bl = (uint16_t *)bloc;
        for (int k = 0; k < dsr; ++k) {
            /* column sums over rows i0..i0+win-1 for columns x >= k (x-k >= 0) */
            for (int x = 0; x < cols; ++x) col[x] = 0;
            for (int dy = 0; dy < win; ++dy) {
                const uint8_t *pl = L + (int64_t)(i0 + dy) * lstep, *pr = R + (int64_t)(i0 + dy) * rstep;
                for (int x = k; x < cols - 1; ++x) {
                    int a = pl[(int64_t)x * ch], c = pr[(int64_t)(x - k) * ch];
                    col[x] += a > c ? a - c : c - a;
                }
            }
            for (int i = i0; i < i1; ++i) {
                if (i > i0) {
                    const uint8_t *pa = L + (int64_t)(i + win - 1) * lstep, *ra = R + (int64_t)(i + win - 1) * rstep;
                    const uint8_t *pd = L + (int64_t)(i - 1) * lstep, *rd = R + (int64_t)(i - 1) * rstep;
                    for (int x = k; x < cols - 1; ++x) {
                        int a = pa[(int64_t)x * ch], c = ra[(int64_t)(x - k) * ch];
                        int d = pd[(int64_t)x * ch], f = rd[(int64_t)(x - k) * ch];
                        col[x] += (a > c ? a - c : c - a) - (d > f ? d - f : f - d);
                    }
                }
                int32_t *bb = best + (size_t)(i - i0) * ocols;
                uint16_t *ll = bl + (size_t)(i - i0) * ocols;
                int32_t run = 0;
                int have = 0;
                for (int j = 0; j < ocols; ++j) {
                    int32_t cost;
                    if (j < k) cost = 0;                      /* invalid slot keeps its 0 (SAD.cpp:40,44) */
                    else {
                        if (!have) { run = 0; for (int dx = 0; dx < win; ++dx) run += col[j + dx]; have = 1; }
                        else run += col[j + win - 1] - col[j - 1];
                        cost = run;
                    }
                    if (k == 0 || cost < bb[j]) { bb[j] = cost; ll[j] = (uint16_t)k; }
                }
            }
        }
        for (int i = i0; i < i1; ++i)
            for (int j = 0; j < ocols; ++j) {
                int loc = bl[(size_t)(i - i0) * ocols + j];
                out[(int64_t)i * ostep + j] = (flags & SAD_ORACLE_RAW_LOC) ? (uint8_t)loc : (uint8_t)(loc * 16);
            }
        free(col); free(best); free(bloc);
    }
    return 0;
}

/*
 * Deterministic random-dot stereogram generator rds(seed, W, H, d_bg, d_fg, binary) as specified
 * in SURVEY.md section 8(d) d2 (splitmix64; left = first W*H draws >> 56, noise = next W*H draws).
 */
static inline uint64_t splitmix64_next(uint64_t *s)
{
    *s += 0x9E3779B97F4A7C15ull;
    uint64_t z = *s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

uint64_t sad_oracle_splitmix64(uint64_t seed, int n)   /* n-th output (1-based) for the KAT */
{
    uint64_t s = seed, v = 0;
    for (int i = 0; i < n; ++i) v = splitmix64_next(&s);
    return v;
}

int sad_oracle_rds(uint64_t seed, int W, int H, int d_bg, int d_fg, int binary, uint8_t *Lo, uint8_t *Ro)
{
    if (!Lo || !Ro || W <= 0 || H <= 0) return -1;
    uint64_t s = seed;
    size_t n = (size_t)W * (size_t)H;
    uint8_t *N = (uint8_t *)malloc(n);
    for (size_t i = 0; i < n; ++i) Lo[i] = (uint8_t)(splitmix64_next(&s) >> 56);
    for (size_t i = 0; i < n; ++i) N[i] = (uint8_t)(splitmix64_next(&s) >> 56);
    if (binary) for (size_t i = 0; i < n; ++i) { Lo[i] = (Lo[i] & 0x80) ? 255 : 0; N[i] = (N[i] & 0x80) ? 255 : 0; }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int d = (y >= H / 4 && y < 3 * H / 4 && x >= W / 4 && x < 3 * W / 4) ? d_fg : d_bg;
            Ro[(size_t)y * W + x] = (x + d < W) ? Lo[(size_t)y * W + x + d] : N[(size_t)y * W + x];
        }
    free(N);
    return 0;
}
