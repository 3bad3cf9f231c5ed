// Single-call cost at the C boundary (no Python, no torch): sadbm_compute_device on one device-resident pair, timed with
// clock_gettime round (a) the call alone, (b) the call + hipStreamSynchronize, (c) n calls back to back + one synchronise.
// The reference's own configuration first: one 384x288 BGR pair, winSize 7, DSR 30 (SAD.cpp:9-11,65-66).
//   hipcc -O2 --offload-arch=gfx950 -I include tools/clat.cpp -L opencv_amd/csrc -lsadbm -Wl,-rpath,'$ORIGIN/../opencv_amd/csrc' -o tools/clat
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <vector>
#include "sadbm.h"

static double now_us()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void nop_kernel() {}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main()
{
    struct Case { const char *name; int W, H, ch, win, D; };
    const Case cases[] = {{"384x288 BGR  w7 D30 (the reference's call)", 384, 288, 3, 7, 30}, {"384x288 grey w7 D30", 384, 288, 1, 7, 30},
                          {"1920x1080 grey w7 D30", 1920, 1080, 1, 7, 30}, {"1920x1080 grey w9 D64", 1920, 1080, 1, 9, 64},
                          {"3840x2160 grey w15 D128", 3840, 2160, 1, 15, 128}};
    CK(hipSetDevice(0));
    sadbm_ctx *ctx = nullptr;
    if (sadbm_create(0, &ctx) != 0) { fprintf(stderr, "sadbm_create failed\n"); return 2; }
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    {   // yardstick: what one launch of an empty kernel + synchronise costs on this box (the floor of any single-launch call)
        std::vector<double> t(300);
        for (int i = 0; i < 320; ++i) {
            const double t0 = now_us();
            hipLaunchKernelGGL(nop_kernel, dim3(1), dim3(64), 0, st);
            CK(hipStreamSynchronize(st));
            if (i >= 20) t[i - 20] = now_us() - t0;
        }
        printf("%-44s call+sync median %7.1f us (min %7.1f)\n", "empty kernel, 1 block (yardstick)", median(t), *std::min_element(t.begin(), t.end()));
    }
    for (const Case &c : cases) {
        const size_t nb = (size_t)c.W * c.H * c.ch;
        std::vector<uint8_t> hl(nb), hr(nb);
        uint32_t s = 12345u;
        for (size_t i = 0; i < nb; ++i) { s = s * 1664525u + 1013904223u; hl[i] = (uint8_t)(s >> 24); }
        for (int y = 0; y < c.H; ++y)                                          // right image: the left one shifted by 9 columns
            for (int x = 0; x < c.W; ++x)
                for (int k = 0; k < c.ch; ++k)
                    hr[((size_t)y * c.W + x) * c.ch + k] = hl[((size_t)y * c.W + std::min(x + 9, c.W - 1)) * c.ch + k];
        uint8_t *dl, *dr, *dout;
        CK(hipMalloc((void **)&dl, nb)); CK(hipMalloc((void **)&dr, nb)); CK(hipMalloc((void **)&dout, (size_t)c.W * c.H));
        CK(hipMemcpy(dl, hl.data(), nb, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr.data(), nb, hipMemcpyHostToDevice));
        auto call = [&]() { return sadbm_compute_device(ctx, dl, (int64_t)c.W * c.ch, dr, (int64_t)c.W * c.ch, c.H, c.W, c.ch, c.win, c.D, 0, dout, c.W, (void *)st); };
        for (int i = 0; i < 20; ++i)
            if (call() != 0) { fprintf(stderr, "call failed: %s\n", sadbm_last_hip_error()); return 2; }
        CK(hipStreamSynchronize(st));
        const int n = 200;
        std::vector<double> t_call(n), t_sync(n);
        for (int i = 0; i < n; ++i) {
            const double t0 = now_us();
            if (call() != 0) return 2;
            const double t1 = now_us();
            CK(hipStreamSynchronize(st));
            const double t2 = now_us();
            t_call[i] = t1 - t0; t_sync[i] = t2 - t0;
        }
        const double t0 = now_us();
        for (int i = 0; i < n; ++i)
            if (call() != 0) return 2;
        CK(hipStreamSynchronize(st));
        const double b2b = (now_us() - t0) / n;
        int nl = 0; float mm = 0, pm = 0;
        sadbm_profile_begin(ctx, n);
        for (int i = 0; i < n; ++i)
            if (call() != 0) return 2;
        CK(hipStreamSynchronize(st));
        sadbm_profile_end(ctx, &nl, &mm, &pm);
        std::vector<uint8_t> ho((size_t)c.W * c.H);
        CK(hipMemcpy(ho.data(), dout, ho.size(), hipMemcpyDeviceToHost));
        unsigned long sum = 0;
        for (uint8_t v : ho) sum += v;
        printf("%-44s call %6.1f us   call+sync median %7.1f us (min %7.1f)   back-to-back %7.1f us/call   kernel (events) %7.1f us  prep %4.1f us   %s  [sum %lu]\n",
               c.name, median(t_call), median(t_sync), *std::min_element(t_sync.begin(), t_sync.end()), b2b, nl ? mm / nl * 1e3 : 0.0, nl ? pm / nl * 1e3 : 0.0,
               sadbm_last_kernel_name(ctx), sum);
        fflush(stdout);
        CK(hipFree(dl)); CK(hipFree(dr)); CK(hipFree(dout));
    }
    sadbm_destroy(ctx);
    return 0;
}
