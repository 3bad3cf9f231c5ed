// Instruction-rate microbenchmarks for the integer/SAD ops the stereo matcher is built from.
// Build: hipcc -O3 --offload-arch=gfx950 ubench.hip -o ubench ; run on the GPU box.
// Reports wave-instructions per cycle-equivalent relative to v_add_u32 (measured in the same run).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef uint32_t u32; typedef uint64_t u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int NCH = 8;      // independent chains per thread
constexpr int ITERS = 2048; // loop trips

struct OpAdd   { static constexpr const char* name = "v_add_u32";        __device__ static u32 f(u32 a, u32 b, u32 c) { return a + (b ^ c); } static constexpr int n = 2; };
struct OpAdd1  { static constexpr const char* name = "v_add_u32(1op)";   __device__ static u32 f(u32 a, u32 b, u32 c) { u32 r; asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(b), "v"(c)); return r; } static constexpr int n = 1; };
struct OpSad   { static constexpr const char* name = "v_sad_u8";         __device__ static u32 f(u32 a, u32 b, u32 c) { return __builtin_amdgcn_sad_u8(a, b, c); } static constexpr int n = 1; };
struct OpSadHi { static constexpr const char* name = "v_sad_hi_u8";      __device__ static u32 f(u32 a, u32 b, u32 c) { return __builtin_amdgcn_sad_hi_u8(a, b, c); } static constexpr int n = 1; };
struct OpSad16 { static constexpr const char* name = "v_sad_u16";        __device__ static u32 f(u32 a, u32 b, u32 c) { return __builtin_amdgcn_sad_u16(a, b, c); } static constexpr int n = 1; };
struct OpMsad  { static constexpr const char* name = "v_msad_u8";        __device__ static u32 f(u32 a, u32 b, u32 c) { return __builtin_amdgcn_msad_u8(a, b, c); } static constexpr int n = 1; };
struct OpMin3  { static constexpr const char* name = "v_min3_u32";       __device__ static u32 f(u32 a, u32 b, u32 c) { u32 r; asm volatile("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; } static constexpr int n = 1; };
struct OpLshlOr{ static constexpr const char* name = "v_lshl_or_b32";    __device__ static u32 f(u32 a, u32 b, u32 c) { u32 r; asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(r) : "v"(c), "v"(b)); return r; } static constexpr int n = 1; };
struct OpAndOr { static constexpr const char* name = "v_and_or_b32";     __device__ static u32 f(u32 a, u32 b, u32 c) { u32 r; asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(c), "v"(a), "v"(b)); return r; } static constexpr int n = 1; };
struct OpAlign { static constexpr const char* name = "v_alignbyte_b32";  __device__ static u32 f(u32 a, u32 b, u32 c) { return __builtin_amdgcn_alignbyte(c, b, 1); } static constexpr int n = 1; };
struct OpPerm  { static constexpr const char* name = "v_perm_b32";       __device__ static u32 f(u32 a, u32 b, u32 c) { return __builtin_amdgcn_perm(c, b, a); } static constexpr int n = 1; };
struct OpPkAdd { static constexpr const char* name = "v_pk_add_u16";     __device__ static u32 f(u32 a, u32 b, u32 c) { u32 r; asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(c), "v"(b)); return r; } static constexpr int n = 1; };
struct OpPkSub { static constexpr const char* name = "v_pk_sub_u16";     __device__ static u32 f(u32 a, u32 b, u32 c) { u32 r; asm volatile("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(c), "v"(b)); return r; } static constexpr int n = 1; };
struct OpPkMin { static constexpr const char* name = "v_pk_min_u16";     __device__ static u32 f(u32 a, u32 b, u32 c) { u32 r; asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(c), "v"(b)); return r; } static constexpr int n = 1; };
struct OpMadU16{ static constexpr const char* name = "v_mad_u32_u16";    __device__ static u32 f(u32 a, u32 b, u32 c) { u32 r; asm volatile("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(c), "v"(a), "v"(b)); return r; } static constexpr int n = 1; };
struct OpSdwaSub{ static constexpr const char* name = "v_sub_u32_sdwa";  __device__ static u32 f(u32 a, u32 b, u32 c) { u32 r = a; asm volatile("v_sub_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(r) : "v"(c), "v"(b)); return r; } static constexpr int n = 1; };
struct OpMinSdwa{ static constexpr const char* name = "v_min_u32_sdwa";  __device__ static u32 f(u32 a, u32 b, u32 c) { u32 r; asm volatile("v_min_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(r) : "v"(c), "v"(b)); return r; } static constexpr int n = 1; };
struct OpDpp   { static constexpr const char* name = "v_mov_b32_dpp(row_shr:1)"; __device__ static u32 f(u32 a, u32 b, u32 c) { return (u32)__builtin_amdgcn_update_dpp((int)a, (int)c, 0x111, 0xf, 0xf, false); } static constexpr int n = 1; };
struct OpAddDpp{ static constexpr const char* name = "v_add_u32_dpp(row_shr:1)"; __device__ static u32 f(u32 a, u32 b, u32 c) { u32 r; asm volatile("v_add_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(c), "v"(b)); return r; } static constexpr int n = 1; };

template <class Op>
__global__ void __launch_bounds__(256) k32(u32* out, u32 seed) {
  u32 t = threadIdx.x + blockIdx.x * blockDim.x;
  u32 acc[NCH], b[NCH];
  for (int i = 0; i < NCH; ++i) { acc[i] = t * 2654435761u + i * seed; b[i] = (t ^ seed) * 40503u + i; }
  u32 a = seed | 0x01010101u;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NCH; ++i) acc[i] = Op::f(a, b[i], acc[i]);
  }
  u32 r = 0;
  for (int i = 0; i < NCH; ++i) r ^= acc[i];
  out[t] = r;
}

// 64-bit quad SADs: src0 = u64 (sliding 8 bytes), src1 = u32, acc = u64 (4 x u16)
struct OpQsad  { static constexpr const char* name = "v_qsad_pk_u16_u8";  __device__ static u64 f(u64 s0, u32 s1, u64 acc) { return __builtin_amdgcn_qsad_pk_u16_u8(s0, s1, acc); } };
struct OpMqsad { static constexpr const char* name = "v_mqsad_pk_u16_u8"; __device__ static u64 f(u64 s0, u32 s1, u64 acc) { return __builtin_amdgcn_mqsad_pk_u16_u8(s0, s1, acc); } };

template <class Op>
__global__ void __launch_bounds__(256) k64(u64* out, u32 seed) {
  u32 t = threadIdx.x + blockIdx.x * blockDim.x;
  u64 acc[NCH]; u64 s0[NCH]; u32 s1[NCH];
  for (int i = 0; i < NCH; ++i) { acc[i] = 0; s0[i] = (u64)(t * 2654435761u + i) * 0x9E3779B97F4A7C15ull + seed; s1[i] = (t ^ seed) * 40503u + i * 77u + 0x01010101u; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NCH; ++i) acc[i] = Op::f(s0[i], s1[i], acc[i]);
  }
  u64 r = 0;
  for (int i = 0; i < NCH; ++i) r ^= acc[i];
  out[t] = r;
}

__global__ void __launch_bounds__(256) kmq32(u32x4* out, u32 seed) {
  u32 t = threadIdx.x + blockIdx.x * blockDim.x;
  u32x4 acc[4]; u64 s0[4]; u32 s1[4];
  for (int i = 0; i < 4; ++i) { acc[i] = (u32x4){0,0,0,0}; s0[i] = (u64)(t * 2654435761u + i) * 0x9E3779B97F4A7C15ull + seed; s1[i] = (t ^ seed) * 40503u + i * 77u + 0x01010101u; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mqsad_u32_u8(s0[i], s1[i], acc[i]);
  }
  u32x4 r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  out[t] = r;
}

// LDS read rates incl. unaligned b32 reads. Each thread reads ITERS*16 words at byte offset (lane*4 + misalign).
template <int MIS, int W>   // W: 4 -> ds_read_b32, 8 -> b64, 16 -> b128
__global__ void __launch_bounds__(256) klds(u32* out, u32 seed) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768 + 64];
  for (int i = threadIdx.x; i < (32768 + 64) / 4; i += 256) ((u32*)lds)[i] = i * 2654435761u + seed;
  __syncthreads();
  u32 base = (threadIdx.x * W + MIS) & 8191;
  u32 r = 0;
  for (int it = 0; it < ITERS / 4; ++it) {
    u32 off0 = base + ((it & 7) << 10);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (W == 4) { u32 v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(off0), "i"(u * 1024)); asm volatile("s_waitcnt lgkmcnt(8)"); r ^= v; }
      if (W == 8) { u64 v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(off0), "i"(u * 1024)); asm volatile("s_waitcnt lgkmcnt(8)"); r ^= (u32)v ^ (u32)(v >> 32); }
      if (W == 16) { u32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(off0), "i"(u * 1024)); asm volatile("s_waitcnt lgkmcnt(8)"); r ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  out[threadIdx.x + blockIdx.x * 256] = r;
}

// Functional probe: semantics of qsad / mqsad / unaligned ds_read on known bytes.
__global__ void kprobe(u64* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64];
  if (threadIdx.x == 0) for (int i = 0; i < 64; ++i) lds[i] = (unsigned char)(i * 3 + 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 s0 = 0x0807060504030201ull;      // bytes 1..8 (byte0 = 1)
    u32 s1 = 0x0A000A0Au;                // bytes 10,10,0,10
    out[0] = __builtin_amdgcn_qsad_pk_u16_u8(s0, s1, 0x0001000100010001ull);
    out[1] = __builtin_amdgcn_mqsad_pk_u16_u8(s0, s1, 0x0001000100010001ull);
    out[2] = __builtin_amdgcn_qsad_pk_u16_u8(s0, s1, 0xFFF0FFF0FFF0FFF0ull);  // wrap or clamp?
    u32 v1, v2, v3; u32 a1 = 1, a2 = 2, a3 = 3;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v1) : "v"(a1));
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v2) : "v"(a2));
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v3) : "v"(a3));
    out[3] = v1; out[4] = v2; out[5] = v3;
    u64 w; u32 a5 = 5;
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(a5));
    out[6] = w;
  }
}

template <class K, class T>
static int run(const char* name, K kern, T* dout, int nblk, double ops_per_thread, double* base_rate, int ncu, double clk_ghz) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), 0, 0, dout, 12345u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), 0, 0, dout, 12345u + r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  double waveinst = ops_per_thread * nblk * 4.0;          // wave-instructions
  double rate = waveinst / (best * 1e-3);                 // wave-instr / s (whole chip)
  double per_simd_cyc = (best * 1e-3) * clk_ghz * 1e9 * ncu * 4 / waveinst;  // cycles per wave-instr per SIMD at nominal clk
  if (base_rate && *base_rate == 0) *base_rate = rate;
  printf("%-28s %8.3f ms  %8.2f Gwaveinst/s  %6.2f cyc/inst/SIMD(@%.1fGHz)  rel=%.3f\n", name, best, rate * 1e-9, per_simd_cyc, clk_ghz,
         base_rate ? rate / *base_rate : 1.0);
  return 0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount; double clk = p.clockRate * 1e-6;
  printf("device %s CUs=%d clk=%.2f GHz\n", p.name, ncu, clk);
  int nblk = ncu * 8;   // 8 blocks of 4 waves per CU = 8 waves/SIMD
  void* d; CK(hipMalloc(&d, (size_t)nblk * 256 * 16));
  u64 hp[8] = {0}; u64* dp; CK(hipMalloc(&dp, 64));
  hipLaunchKernelGGL(kprobe, dim3(1), dim3(64), 0, 0, dp); CK(hipMemcpy(hp, dp, 64, hipMemcpyDeviceToHost));
  printf("probe qsad   = %016llx\nprobe mqsad  = %016llx\nprobe qsad wrap = %016llx\n", (unsigned long long)hp[0], (unsigned long long)hp[1], (unsigned long long)hp[2]);
  printf("probe ds_read_b32 @1,2,3 = %08llx %08llx %08llx (bytes are i*3+1)\nprobe ds_read_b64 @5 = %016llx\n",
         (unsigned long long)hp[3], (unsigned long long)hp[4], (unsigned long long)hp[5], (unsigned long long)hp[6]);
  double base = 0;
  double n32 = (double)ITERS * 4 * NCH;
#define R32(Op) if (run(Op::name, k32<Op>, (u32*)d, nblk, n32 * Op::n, &base, ncu, clk)) return 1;
  R32(OpAdd1) R32(OpAdd) R32(OpSad) R32(OpSadHi) R32(OpSad16) R32(OpMsad) R32(OpMin3) R32(OpLshlOr) R32(OpAndOr) R32(OpAlign) R32(OpPerm)
  R32(OpPkAdd) R32(OpPkSub) R32(OpPkMin) R32(OpMadU16) R32(OpSdwaSub) R32(OpMinSdwa) R32(OpDpp) R32(OpAddDpp)
  if (run(OpQsad::name, k64<OpQsad>, (u64*)d, nblk, n32, &base, ncu, clk)) return 1;
  if (run(OpMqsad::name, k64<OpMqsad>, (u64*)d, nblk, n32, &base, ncu, clk)) return 1;
  if (run("v_mqsad_u32_u8", kmq32, (u32x4*)d, nblk, (double)ITERS * 32, &base, ncu, clk)) return 1;
  double nl = (double)(ITERS / 4) * 16;
  if (run("ds_read_b32 aligned", klds<0, 4>, (u32*)d, nblk, nl, &base, ncu, clk)) return 1;
  if (run("ds_read_b32 mis+1", klds<1, 4>, (u32*)d, nblk, nl, &base, ncu, clk)) return 1;
  if (run("ds_read_b32 mis+2", klds<2, 4>, (u32*)d, nblk, nl, &base, ncu, clk)) return 1;
  if (run("ds_read_b64 aligned", klds<0, 8>, (u32*)d, nblk, nl, &base, ncu, clk)) return 1;
  if (run("ds_read_b64 mis+4", klds<4, 8>, (u32*)d, nblk, nl, &base, ncu, clk)) return 1;
  if (run("ds_read_b64 mis+1", klds<1, 8>, (u32*)d, nblk, nl, &base, ncu, clk)) return 1;
  if (run("ds_read_b128 aligned", klds<0, 16>, (u32*)d, nblk, nl, &base, ncu, clk)) return 1;
  if (run("ds_read_b128 mis+4", klds<4, 16>, (u32*)d, nblk, nl, &base, ncu, clk)) return 1;
  if (run("ds_read_b128 mis+8", klds<8, 16>, (u32*)d, nblk, nl, &base, ncu, clk)) return 1;
  return 0;
}
