// abi_consumer.cpp -- a plain C++ program (g++, no HIP headers, no Python) that uses libsadbm the way a C++ caller of the
// reference's getDisparity() would: through include/sadbm.hpp / sadbm.h only.  Built by `make -C opencv_amd/csrc abi_consumer`
// and run on the GPU box by tests/test_cpp_consumer.py.  It checks the entry points against each other and against the
// known answers that need no oracle (K1 identical images, K2 constant images, K5 zero frame):
//   sadbm::getDisparity(MatView)  ==  sadbm_compute_multi (1 GPU through RCCL)  ==  sadbm_compute_multi (3 "devices" on GPU 0,
//   peer-copy gather, pipelined sub-bands)  ==  sadbm_compute_multi_device (the frame already in GPU memory, band scatter by
//   peer copies)  ==  sadbm_compute_multi_batch  ==  the streaming pipeline.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sadbm.hpp"

// the four HIP runtime calls a caller with device-resident frames makes itself (declared by hand: no HIP headers needed;
// libamdhip64 is what libsadbm.so links anyway)
extern "C" {
int hipMalloc(void **ptr, size_t size);
int hipFree(void *ptr);
int hipMemcpy(void *dst, const void *src, size_t size, int kind);   // kind 1 = host to device, 2 = device to host
int hipDeviceSynchronize(void);
}

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { ++fails; printf("FAIL line %d: ", __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static uint32_t rng = 2463534242u;
static uint8_t rnd() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return (uint8_t)(rng >> 11); }

// `abi_consumer --latency`: the synchronous host call at the C boundary on the reference's own configuration
// (384x288 BGR pair, winSize 7, DSR 30: SAD.cpp:9-11,65-66) and on larger frames -- no Python in the loop.
static int latency()
{
    const int cases[][5] = {{384, 288, 3, 7, 30}, {1920, 1080, 1, 9, 64}, {1920, 1080, 3, 9, 64}, {3840, 2160, 1, 15, 128}, {3840, 2160, 3, 15, 128}};
    for (const auto &c : cases) {
        const int W = c[0], H = c[1], ch = c[2], win = c[3], D = c[4];
        sadbm::Image L(H, W, ch), R(H, W, ch), out(H, W, 1);
        for (size_t i = 0; i < L.pix.size(); ++i) { L.pix[i] = rnd(); R.pix[i] = rnd(); }
        sadbm_image li{L.pix.data(), H, W, ch, (int64_t)W * ch}, ri{R.pix.data(), H, W, ch, (int64_t)W * ch};
        sadbm_plane op{out.pix.data(), H, W, W};
        for (int i = 0; i < 5; ++i) if (sadbm_get_disparity(&li, &ri, win, D, 0, &op)) { printf("FAIL: %s\n", sadbm_last_hip_error()); return 1; }
        const int n = W > 2000 ? 50 : 300;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) sadbm_get_disparity(&li, &ri, win, D, 0, &op);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        printf("sadbm_get_disparity %4dx%4d x%d win %2d D %3d: %8.1f us per call\n", W, H, ch, win, D, us);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "--latency")) return latency();
    const int H = 203, W = 517, win = 9, D = 48;
    sadbm::Image L(H, W, 3), R(H, W, 3);                        // BGR, as cv::imread hands the reference (SAD.cpp:65-66)
    for (size_t i = 0; i < L.pix.size(); ++i) L.pix[i] = rnd();
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c) R.ptr(y)[3 * x + c] = x + 11 < W ? L.ptr(y)[3 * (x + 11) + c] : rnd();   // planted disparity 11
    try {
        const sadbm::Image ref = sadbm::getDisparity(L.view(), R.view(), win, D, SADBM_RAW_LOC);
        // K4: the planted shift is found exactly in the interior; K5: zero frame; K3-like: j+1 ramp at the left border
        CHECK(ref.ptr(50)[200] == 11 && ref.ptr(100)[400] == 11, "planted disparity: %d %d", ref.ptr(50)[200], ref.ptr(100)[400]);
        for (int x = 0; x < W; ++x) CHECK(ref.ptr(H - 1)[x] == 0 && ref.ptr(H - win)[x] == 0, "zero rows at %d", x);
        for (int y = 0; y < H; ++y) CHECK(ref.ptr(y)[W - 1] == 0 && ref.ptr(y)[W - win] == 0, "zero cols at %d", y);
        for (int x = 0; x < 11; ++x) CHECK(ref.ptr(7)[x] == x + 1, "left ramp at %d: %d", x, ref.ptr(7)[x]);
        // K1: identical images -> loc 0 everywhere
        const sadbm::Image same = sadbm::getDisparity(L.view(), L.view(), 7, 30, SADBM_RAW_LOC);
        for (size_t i = 0; i < same.pix.size(); ++i) if (same.pix[i]) { CHECK(false, "K1 at %zu", i); break; }
        // default encoding = (loc * 16) & 255 (SAD.cpp:57)
        const sadbm::Image enc = sadbm::getDisparity(L.view(), R.view(), win, D);
        for (size_t i = 0; i < enc.pix.size(); ++i) if (enc.pix[i] != (uint8_t)(ref.pix[i] * 16)) { CHECK(false, "encode at %zu", i); break; }

        sadbm_image li{L.pix.data(), H, W, 3, (int64_t)W * 3}, ri{R.pix.data(), H, W, 3, (int64_t)W * 3};
        // single-process multi-GPU entry points
        for (int variant = 0; variant < 3; ++variant) {
            const int n = variant == 0 ? 1 : 3;
            const int devs[3] = {0, 0, 0};
            sadbm_multi *m = nullptr;
            int rc = sadbm_multi_create(n, devs, variant == 0 ? 0u : variant == 1 ? SADBM_MULTI_COPY_GATHER : (SADBM_MULTI_COPY_GATHER | SADBM_MULTI_FORCE_PEER_COPY), &m);
            CHECK(rc == 0, "multi_create variant %d: %s (%s)", variant, sadbm_strerror(rc), sadbm_last_hip_error());
            if (rc) continue;
            sadbm::Image out(H, W, 1);
            sadbm_plane op{out.pix.data(), H, W, W};
            rc = sadbm_compute_multi(m, &li, &ri, win, D, SADBM_RAW_LOC, variant == 2 ? 5 : 1, &op);
            CHECK(rc == 0, "compute_multi variant %d: %s", variant, sadbm_strerror(rc));
            CHECK(out.pix == ref.pix, "compute_multi variant %d differs from getDisparity", variant);
            // the same frame, already resident on GPU 0 (BGR, pitch = row bytes): entry 0 reads in place, the others pull their bands
            {
                const size_t fb = (size_t)H * W * 3;
                void *dl = nullptr, *dr = nullptr, *dout = nullptr;
                CHECK(hipMalloc(&dl, fb) == 0 && hipMalloc(&dr, fb) == 0 && hipMalloc(&dout, (size_t)H * W) == 0, "hipMalloc");
                CHECK(hipMemcpy(dl, L.pix.data(), fb, 1) == 0 && hipMemcpy(dr, R.pix.data(), fb, 1) == 0, "hipMemcpy H2D");
                rc = sadbm_compute_multi_device(m, n - 1, (const uint8_t *)dl, (int64_t)W * 3, (const uint8_t *)dr, (int64_t)W * 3, H, W, 3, win, D,
                                                SADBM_RAW_LOC, variant == 2 ? 4 : 1, (uint8_t *)dout, W, nullptr);
                CHECK(rc == 0, "compute_multi_device variant %d: %s (%s)", variant, sadbm_strerror(rc), sadbm_last_hip_error());
                CHECK(hipDeviceSynchronize() == 0, "sync");       // the call is asynchronous
                sadbm::Image dev_out(H, W, 1);
                CHECK(hipMemcpy(dev_out.pix.data(), dout, (size_t)H * W, 2) == 0, "hipMemcpy D2H");
                CHECK(dev_out.pix == ref.pix, "compute_multi_device variant %d differs from getDisparity", variant);
                hipFree(dl); hipFree(dr); hipFree(dout);
            }
            // batch: 5 pairs (the same pair five times, one with L and R swapped) dealt round-robin
            std::vector<sadbm_image> ls(5, li), rs(5, ri);
            ls[3] = ri; rs[3] = li;
            std::vector<sadbm::Image> outs(5, sadbm::Image(H, W, 1));
            std::vector<sadbm_plane> ops;
            for (auto &o : outs) ops.push_back(sadbm_plane{o.pix.data(), H, W, W});
            rc = sadbm_compute_multi_batch(m, ls.data(), rs.data(), 5, win, D, SADBM_RAW_LOC, ops.data());
            CHECK(rc == 0, "compute_multi_batch variant %d: %s", variant, sadbm_strerror(rc));
            const sadbm::Image swapped = sadbm::getDisparity(R.view(), L.view(), win, D, SADBM_RAW_LOC);
            for (int i = 0; i < 5; ++i) CHECK(outs[i].pix == (i == 3 ? swapped.pix : ref.pix), "batch pair %d (variant %d)", i, variant);
            sadbm_multi_destroy(m);
        }
        // streaming pipeline: 6 frames through 3 slots
        sadbm_pipeline *p = nullptr;
        int rc = sadbm_pipeline_create(0, 3, H, W, 3, win, D, SADBM_RAW_LOC, &p);
        CHECK(rc == 0, "pipeline_create: %s", sadbm_strerror(rc));
        if (!rc) {
            int64_t t[6];
            for (int i = 0; i < 6; ++i) {
                if (i >= 3) {
                    const uint8_t *res; int64_t step;
                    CHECK(sadbm_pipeline_wait(p, t[i - 3], &res, &step, nullptr) == 0, "wait %d", i - 3);
                    for (int y = 0; y < H; ++y) if (memcmp(res + y * step, ref.ptr(y), W)) { CHECK(false, "pipeline frame %d row %d", i - 3, y); break; }
                }
                CHECK(sadbm_pipeline_submit(p, &li, &ri, &t[i]) == 0, "submit %d", i);
            }
            for (int i = 3; i < 6; ++i) CHECK(sadbm_pipeline_wait(p, t[i], nullptr, nullptr, nullptr) == 0, "wait %d", i);
            sadbm_pipeline_destroy(p);
        }
        // error mapping: right smaller than left -> SADBM_E_SIZE (the reference throws in Mat::operator()(Rect), SAD.cpp:47)
        sadbm::MatView small = R.view();
        small.rows = H - 1;
        try { sadbm::getDisparity(L.view(), small); CHECK(false, "no error for a smaller right image"); }
        catch (const sadbm::Error &e) { CHECK(e.code == SADBM_E_SIZE, "code %d", e.code); }
    } catch (const sadbm::Error &e) {
        printf("FAIL: %s\n", e.what());
        return 1;
    }
    sadbm_thread_release();
    printf(fails ? "abi_consumer: %d FAILED\n" : "abi_consumer: ok%.0d\n", fails);
    return fails ? 1 : 0;
}
