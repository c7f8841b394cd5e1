#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out)
{
    unsigned x = threadIdx.x;
    u2 a = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    u2 b = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    out[threadIdx.x] = a[0]; out[64 + threadIdx.x] = a[1]; out[128 + threadIdx.x] = b[0]; out[192 + threadIdx.x] = b[1];
}
int main()
{
    unsigned* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[4] = {"p32[0]", "p32[1]", "p16[0]", "p16[1]"};
    for (int q = 0; q < 4; ++q) { printf("%s:", nm[q]); for (int i = 0; i < 64; ++i) printf(" %u", h[64 * q + i]); printf("\n"); }
    return 0;
}
