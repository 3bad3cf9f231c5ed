// hostcopy_selftest.cpp -- CPU self-test of hostcopy.hpp, built with -fsanitize=address,undefined (make asan): the SSSE3 / AVX2
// channel-0 gathers and the pool-parallel row copy on exact-size heap buffers (any read or write past an end is caught),
// odd widths, 3 and 4 channels, tails shorter than a vector, pitched sources, copies large enough to be split over the pool,
// and calls from several threads at once.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "hostcopy.hpp"

using namespace sadbm_hostcopy;

static int fails = 0;
static uint32_t rng = 88172645u;
static uint8_t rnd() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return (uint8_t)(rng >> 9); }

static void one(int rows, int cols, int ch, int spad, int dpad)
{
    const size_t sstep = (size_t)cols * ch + spad, dstep = (size_t)cols + dpad;
    // exact-size buffers: the last row carries no padding
    const size_t sbytes = rows ? sstep * (rows - 1) + (size_t)cols * ch : 0, dbytes = rows ? dstep * (rows - 1) + cols : 0;
    uint8_t *src = (uint8_t *)malloc(sbytes ? sbytes : 1), *dst = (uint8_t *)malloc(dbytes ? dbytes : 1);
    for (size_t i = 0; i < sbytes; ++i) src[i] = rnd();
    for (size_t i = 0; i < dbytes; ++i) dst[i] = 0xA5;
    copy_rows(dst, (int64_t)dstep, src, (int64_t)sstep, (size_t)cols, rows, ch);
    for (int y = 0; y < rows; ++y) {
        for (int x = 0; x < cols; ++x)
            if (dst[y * dstep + x] != src[y * sstep + (size_t)x * ch]) { if (!fails++) printf("FAIL rows %d cols %d ch %d at (%d,%d)\n", rows, cols, ch, y, x); }
        if (y + 1 < rows)
            for (int x = cols; x < (int)dstep; ++x)
                if (dst[y * dstep + x] != 0xA5) { if (!fails++) printf("FAIL: padding written, rows %d cols %d ch %d row %d\n", rows, cols, ch, y); }
    }
    free(src); free(dst);
}

int main()
{
    for (int ch : {1, 3, 4})
        for (int cols : {1, 2, 5, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 100, 257, 384, 1001})
            for (int rows : {1, 2, 7})
                for (int spad : {0, 5}) one(rows, cols, ch, spad, spad ? 3 : 0);
    // large enough for the pool (>= 512 KB): split over the workers, uneven last part
    one(1081, 1920, 3, 0, 0);
    one(723, 1283, 4, 7, 1);
    one(2161, 3840, 1, 0, 0);
    // several callers at once (the pool runs one parallel copy at a time, the others queue)
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t) th.emplace_back([t] { for (int i = 0; i < 3; ++i) one(700 + 13 * t, 1500 + i, 1 + 2 * (t & 1), t, 0); });
    for (auto &x : th) x.join();
    printf(fails ? "hostcopy selftest: %d FAILED\n" : "hostcopy selftest: ok\n", fails);
    return fails ? 1 : 0;
}
