// Shared device/host helpers for the CDNA4 (gfx950) kernels of libcdetr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CDETR_OK 0
#define CDETR_ERR_ARG (-1)
#define CDETR_ERR_LAUNCH (-2)
#define CDETR_ERR_UNSUPPORTED (-3)

// thread-local last-error string (cdetr_last_error)
void cdetr_set_error(const char* fmt, ...);

#define CDETR_CHECK_ARG(cond, ...)          \
    do {                                    \
        if (!(cond)) {                      \
            cdetr_set_error(__VA_ARGS__);   \
            return CDETR_ERR_ARG;           \
        }                                   \
    } while (0)

static inline int cdetr_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        cdetr_set_error("%s: %s", what, hipGetErrorString(e));
        return CDETR_ERR_LAUNCH;
    }
    return CDETR_OK;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
