// gfx950 v_permlane16_swap / v_permlane32_swap: lane semantics and issue cost (run on the GPU box).
// Build: hipcc -O3 --offload-arch=gfx950 ubench3.hip -o ubench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef uint32_t u32;

__global__ void probe(u32* o)
{
    int l = threadIdx.x;
    u32 a = 1000 + l, b = 2000 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[l] = r[0]; o[64 + l] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[128 + l] = q[0]; o[192 + l] = q[1];
}

constexpr int NCH = 8, ITERS = 2048;
template <int MODE>   // 0: v_min_u32 only, 1: permlane32_swap + min, 2: permlane16_swap + min, 3: quad_perm dpp min x2
__global__ void __launch_bounds__(256) rate(u32* out, u32 seed)
{
    u32 t = threadIdx.x + blockIdx.x * blockDim.x;
    u32 x[NCH], y[NCH];
    for (int i = 0; i < NCH; ++i) { x[i] = t * 2654435761u + i * seed; y[i] = (t ^ seed) * 40503u + i; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (MODE == 0) { x[i] = min(x[i], y[i] + it); }
            if (MODE == 1) { auto r = __builtin_amdgcn_permlane32_swap(x[i], y[i], false, false); x[i] = min(r[0], r[1]); y[i] = r[1] + it; }
            if (MODE == 2) { auto r = __builtin_amdgcn_permlane16_swap(x[i], y[i], false, false); x[i] = min(r[0], r[1]); y[i] = r[1] + it; }
            if (MODE == 3) {
                x[i] = min(x[i], (u32)__builtin_amdgcn_mov_dpp((int)x[i], 0xB1, 0xF, 0xF, true));
                x[i] = min(x[i] + it, (u32)__builtin_amdgcn_mov_dpp((int)y[i], 0x4E, 0xF, 0xF, true));
            }
        }
    }
    u32 r = 0;
    for (int i = 0; i < NCH; ++i) r ^= x[i] ^ y[i];
    out[t] = r;
}

template <int MODE> int run(const char* name, u32* d, int ops_per_iter, double clk_ghz, int cus)
{
    const int blocks = cus * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double waveinst = (double)blocks * 4 * ITERS * NCH * ops_per_iter;
    double cyc = ms * 1e-3 * clk_ghz * 1e9 * cus * 4 / waveinst;
    printf("%-34s %8.3f ms  %6.2f cyc per counted op per SIMD (%d ops/iter counted)\n", name, ms, cyc, ops_per_iter);
    return 0;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const double clk = p.clockRate * 1e-6; const int cus = p.multiProcessorCount;
    u32* d; CK(hipMalloc(&d, cus * 8 * 256 * 4 + 1024));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    u32 h[256]; CK(hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost));
    const char* names[4] = {"permlane32_swap vdst'", "permlane32_swap src0'", "permlane16_swap vdst'", "permlane16_swap src0'"};
    for (int j = 0; j < 4; ++j) { printf("%s (lanes 0,8,..,56): ", names[j]); for (int i = 0; i < 64; i += 8) printf("%u ", h[j * 64 + i]); printf("\n"); }
    run<0>("v_add + v_min (2 VALU)", d, 2, clk, cus);
    run<1>("permlane32_swap + min + add (3)", d, 3, clk, cus);
    run<2>("permlane16_swap + min + add (3)", d, 3, clk, cus);
    run<3>("2 x (dpp quad_perm min) + add (3)", d, 3, clk, cus);
    return 0;
}
