// The wave64 inclusive prefix sum of sad_wide_kernel (DPP row shifts + gfx9 row broadcasts) against a host loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}
__global__ void probe(const uint32_t *in, uint32_t *out) { out[threadIdx.x] = wave_scan_add(in[threadIdx.x]); }
int main()
{
    uint32_t h[64], r[64], *d, *o;
    for (int i = 0; i < 64; ++i) h[i] = (uint32_t)(i * 2654435761u) >> 12;
    if (hipMalloc(&d, 256) != hipSuccess || hipMalloc(&o, 256) != hipSuccess) return 1;
    if (hipMemcpy(d, h, 256, hipMemcpyHostToDevice) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
    if (hipMemcpy(r, o, 256, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    uint32_t acc = 0; int bad = 0;
    for (int i = 0; i < 64; ++i) { acc += h[i]; if (r[i] != acc) { if (!bad) printf("lane %d: got %u want %u\n", i, r[i], acc); ++bad; } }
    printf("wave scan: %d wrong lanes\n", bad);
    return bad != 0;
}
