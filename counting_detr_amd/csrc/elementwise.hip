// elementwise.hip -- HBM-bound fused elementwise kernels of the train step (float4 per lane, grid-stride).
//
// cdetr_sumsq       : partial sums of squares of the flat gradient arena (global grad-norm for clip_grad_norm_).
// cdetr_adamw_step  : clip (x coef) + AdamW (decoupled weight decay, bias correction) over the flat parameter / gradient
//                     / moment arenas in ONE pass: 4 reads + 3 writes per element instead of ~12 torch passes
//                     (A2/engine.py:54-57, A2/main.py:186: torch.optim.AdamW defaults betas (0.9, 0.999), eps 1e-8).
//                     Step count, learning-rate scale and the clip coefficient live in device memory (graph replay safe).
// cdetr_relu_mask   : dz = (y > 0) ? dy * scale : 0   (ReLU backward for the linear layers, one pass).
#include "../../include/cdetr_hip.h"
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
    float s = 0.f;
    const long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = g4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    s = wave_sum(s);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// state[0] = step count t (float, incremented here by block 0), state[1] = lr scale (StepLR factor),
// sumsq[0] = sum of squares of the gradient (from sumsq_kernel); outputs total_norm to state[2].
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const float* __restrict__ lr, long n,
                                                    const float* __restrict__ sumsq, float* __restrict__ state, float max_norm,
                                                    float beta1, float beta2, float eps, float wd, float grad_div) {
    const float t = state[0] + 1.f;
    const float lr_scale = state[1];
    const float total_norm = sqrtf(sumsq[0]) * grad_div;
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(max_norm / (total_norm + 1e-6f), 1.f);
    coef *= grad_div;                      // grad_div = 1 / world_size folds the data-parallel average into the same pass
    const float bc1 = 1.f - powf(beta1, t);
    const float bc2 = 1.f - powf(beta2, t);
    const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    const long n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    const float4* l4 = reinterpret_cast<const float4*>(lr);
    auto upd = [&](float& pp, float gg, float& mm, float& vv, float l) {
        gg *= coef;
        l *= lr_scale;
        pp *= 1.f - l * wd;
        mm = beta1 * mm + (1.f - beta1) * gg;
        vv = beta2 * vv + (1.f - beta2) * gg * gg;
        const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        pp -= (l / bc1) * (mm / denom);
    };
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i], ll = l4[i];
        upd(pp.x, gg.x, mm.x, vv.x, ll.x);
        upd(pp.y, gg.y, mm.y, vv.y, ll.y);
        upd(pp.z, gg.z, mm.z, vv.z, ll.z);
        upd(pp.w, gg.w, mm.w, vv.w, ll.w);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) upd(p[i], g[i], m[i], v[i], lr[i]);
    // every block has read state[0] / sumsq before block 0 writes them only if ... they are written by a SEPARATE tiny
    // kernel (adamw_finish) launched after this one -- see cdetr_adamw_step.
}

__global__ void adamw_finish_kernel(const float* __restrict__ sumsq, float* __restrict__ state, float grad_div) {
    state[2] = sqrtf(sumsq[0]) * grad_div;   // total gradient norm (of the averaged gradient), for logging
    state[0] += 1.f;
}

__global__ __launch_bounds__(256) void relu_mask_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                        float* __restrict__ dz, long n, float scale) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(y)[i];
        float4 b = reinterpret_cast<const float4*>(dy)[i];
        b.x = a.x > 0.f ? b.x * scale : 0.f;
        b.y = a.y > 0.f ? b.y * scale : 0.f;
        b.z = a.z > 0.f ? b.z * scale : 0.f;
        b.w = a.w > 0.f ? b.w * scale : 0.f;
        reinterpret_cast<float4*>(dz)[i] = b;
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) dz[i] = y[i] > 0.f ? dy[i] * scale : 0.f;
}

inline int grid_for(long n4) {
    long b = (n4 + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int cdetr_sumsq(const float* g, int64_t n, float* out, void* stream) {
    CDETR_CHECK_ARG(g && out && n >= 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0, "cdetr_sumsq: bad args");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), st);
    if (e != hipSuccess) { cdetr_set_error("cdetr_sumsq: memset: %s", hipGetErrorString(e)); return CDETR_ERR_LAUNCH; }
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, st, g, (long)n, out);
    return cdetr_launch_status("cdetr_sumsq");
}

extern "C" int cdetr_adamw_step(float* p, const float* g, float* m, float* v, const float* lr, int64_t n, const float* sumsq,
                                float* state, float max_norm, float beta1, float beta2, float eps, float weight_decay,
                                float grad_div, void* stream) {
    CDETR_CHECK_ARG(p && g && m && v && lr && sumsq && state && n >= 0, "cdetr_adamw_step: null pointer");
    CDETR_CHECK_ARG(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                      reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(lr)) & 15) == 0,
                    "cdetr_adamw_step: arenas must be 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, st, p, g, m, v, lr, (long)n, sumsq, state, max_norm,
                       beta1, beta2, eps, weight_decay, grad_div);
    hipLaunchKernelGGL(adamw_finish_kernel, dim3(1), dim3(1), 0, st, sumsq, state, grad_div);
    return cdetr_launch_status("cdetr_adamw_step");
}

extern "C" int cdetr_relu_mask(const float* y, const float* dy, float* dz, int64_t n, float scale, void* stream) {
    CDETR_CHECK_ARG(y && dy && dz && n >= 0, "cdetr_relu_mask: bad args");
    CDETR_CHECK_ARG(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dz)) & 15) == 0,
                    "cdetr_relu_mask: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(relu_mask_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y, dy, dz,
                       (long)n, scale);
    return cdetr_launch_status("cdetr_relu_mask");
}
