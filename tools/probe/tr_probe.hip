// Probe of ds_read_b64_tr_b16 (gfx950): lane l reads the 4 x b16 chunk at LDS element offset 4*l of an image holding
// s[i] = i, so every returned value v tells its origin: source lane v / 4, element v % 4.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(v4s* o) {
    __shared__ __attribute__((aligned(16))) short s[256];
    for (int i = threadIdx.x; i < 256; i += 64) s[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    o[l] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(s + 4 * l));
}
int main() {
    v4s* d;
    hipMalloc(&d, 64 * sizeof(v4s));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    v4s h[64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf("  (L%2d,e%d)", h[l][e] / 4, h[l][e] % 4);
        printf("\n");
    }
    return 0;
}
