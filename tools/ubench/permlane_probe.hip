#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(uint32_t* o)
{
    int l = threadIdx.x;
    uint32_t a = 1000 + l, b = 2000 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[l] = r[0]; o[64 + l] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[128 + l] = q[0]; o[192 + l] = q[1];
}
int main(){ uint32_t* d; hipMalloc(&d, 1024); k<<<1,64>>>(d); uint32_t h[256]; hipMemcpy(h,d,1024,hipMemcpyDeviceToHost);
 for(int j=0;j<4;j++){ for(int i=0;i<64;i+=8) printf("%u ", h[j*64+i]); printf("\n"); } return 0; }
