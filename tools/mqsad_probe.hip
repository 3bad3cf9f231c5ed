// v_mqsad_pk_u16_u8 as sad_wide_kernel uses it: reference operand = one byte (the other three zero = masked out).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const uint32_t *in, uint64_t *out)
{
    const uint32_t src = in[0], ref = in[1];
    out[0] = __builtin_amdgcn_mqsad_pk_u16_u8((uint64_t)src, ref, 0ull);
    out[1] = __builtin_amdgcn_mqsad_pk_u16_u8((uint64_t)src | 0xAABBCCDD00000000ull, ref, 0x0001000200030004ull);
    out[2] = __builtin_amdgcn_mqsad_pk_u16_u8((uint64_t)src, 0u, 0ull);
    out[3] = __builtin_amdgcn_mqsad_pk_u16_u8((uint64_t)(~src), 255u, 0ull);
}
int main()
{
    uint32_t h[2] = {0x40302010u, 0x25u}, *d; uint64_t *o, r[4];
    if (hipMalloc(&d, 8) != hipSuccess || hipMalloc(&o, 32) != hipSuccess) return 1;
    hipMemcpy(d, h, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, d, o);
    if (hipMemcpy(r, o, 32, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    // expected: fields |0x10-0x25|,|0x20-0x25|,|0x30-0x25|,|0x40-0x25| = 0x15,0x05,0x0b,0x1b from the low field up
    for (int i = 0; i < 4; ++i) printf("%d: %016llx\n", i, (unsigned long long)r[i]);
    return 0;
}
