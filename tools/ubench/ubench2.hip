// Mixed-instruction issue-rate probe: how many cycles per VALU instruction for the matcher's mix
// (v_sad_hi_u8 VOP3 + v_and/v_add/v_sub/v_min VOP2) at 1, 2, 4, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef uint32_t u32;
constexpr int ITERS = 2048;

// MODE 0: only sad (16 independent chains)   1: only VOP2 add (16 chains)   2: sad + and (1:1)
// MODE 3: sad + and + add (1:1:1)            4: matcher-like: per 5 sads, 3 ands, 2 add/sub, 1 min
// MODE 5: VOP2 and+add+min only, 3 chains deep
template <int MODE>
__global__ void __launch_bounds__(256) kmix(u32* out, u32 seed) {
  extern __shared__ u32 lds[];
  u32 t = threadIdx.x + blockIdx.x * blockDim.x;
  u32 acc[16], b[16];
  for (int i = 0; i < 16; ++i) { acc[i] = t * 2654435761u + i * seed; b[i] = (t ^ seed) * 40503u + i; }
  u32 a = seed | 0x01010101u, m = seed ^ 0x00FFFFFFu;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_sad_hi_u8(a, b[i], acc[i]);
      if (MODE == 1) acc[i] = acc[i] + b[i];
      if (MODE == 2) { u32 x = b[i] & acc[i]; acc[i] = __builtin_amdgcn_sad_hi_u8(a, x, acc[i]); }
      if (MODE == 3) { u32 x = b[i] & acc[i]; acc[i] = __builtin_amdgcn_sad_hi_u8(a, x, acc[i]) + b[(i + 1) & 15]; }
      if (MODE == 4) {
        u32 x0 = b[i] & m, x1 = acc[i] & m, x2 = b[(i + 3) & 15] & acc[i];
        u32 s = __builtin_amdgcn_sad_hi_u8(a, b[i], 0);
        s = __builtin_amdgcn_sad_hi_u8(a, x0, s);
        u32 r = __builtin_amdgcn_sad_hi_u8(b[(i + 1) & 15], x1, acc[i]);
        r = __builtin_amdgcn_sad_hi_u8(a, x2, r);
        u32 o = __builtin_amdgcn_sad_hi_u8(b[(i + 2) & 15], x0, s);
        r = r + s; r = r - o;
        acc[i] = r; b[i] = min(b[i], r);
      }
      if (MODE == 5) { u32 x = b[i] & acc[i]; x = x + b[(i + 1) & 15]; acc[i] = min(acc[i], x) ^ it; }
    }
  }
  u32 r = 0;
  for (int i = 0; i < 16; ++i) r ^= acc[i] ^ b[i];
  out[t] = r + (lds ? 0 : 1);
}

template <int MODE>
int run(const char* name, double valu_per_iter, u32* d, int ncu, double clk) {
  for (int occ = 1; occ <= 8; occ *= 2) {
    size_t lds = (160 * 1024) / occ - 1024;           // limits resident blocks per CU to `occ`
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)kmix<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int nblk = ncu * occ * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kmix<MODE>, dim3(nblk), dim3(256), lds, 0, d, 1234u); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(kmix<MODE>, dim3(nblk), dim3(256), lds, 0, d, 99u + r);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double winst = valu_per_iter * 16 * ITERS * nblk * 4.0;
    double cyc = best * 1e-3 * clk * 1e9 * ncu * 4 / winst;
    printf("%-34s occ=%d waves/SIMD  %8.3f ms  %5.2f cyc/VALU-inst/SIMD (@%.1f GHz nominal)\n", name, occ, best, cyc, clk);
  }
  return 0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount; double clk = p.clockRate * 1e-6;
  u32* d; CK(hipMalloc(&d, (size_t)ncu * 8 * 4 * 256 * 4));
  if (run<0>("sad_hi only", 1, d, ncu, clk)) return 1;
  if (run<1>("v_add_u32 only", 1, d, ncu, clk)) return 1;
  if (run<2>("and + sad (dep)", 2, d, ncu, clk)) return 1;
  if (run<3>("and + sad + add (dep)", 3, d, ncu, clk)) return 1;
  if (run<4>("matcher mix 5 sad:3 and:2 add:1 min", 11, d, ncu, clk)) return 1;
  if (run<5>("and + add + min + xor (VOP2 only)", 4, d, ncu, clk)) return 1;
  return 0;
}
