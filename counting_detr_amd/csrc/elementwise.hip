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
#include <stdlib.h>

namespace {

// Bit-reproducible: every block parks its partial sum in ws[block]; a one-block second kernel adds the partials in index
// order.  (An atomicAdd per block would make the clip coefficient -- and with it the parameters -- depend on the arrival
// order: data-parallel ranks holding the same reduced gradient would drift apart by an ulp per step.  A last-arriving-block
// epilogue inside the same kernel needs a device-scope release per block: measured 74 us against 46 + 3 us for two launches.)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ ws) {
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
    if (threadIdx.x == 0) ws[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ ws, int nblocks, float* __restrict__ out) {
    float t = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) t += ws[i];
    t = wave_sum(t);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (part[0] + part[1]) + (part[2] + part[3]);
}

// state[0] = step count t (float, incremented here by block 0), state[1] = lr scale (StepLR factor),
// sumsq[0] = sum of squares of the gradient (from sumsq_kernel); outputs total_norm to state[2].
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const float* __restrict__ lr, long n,
                                                    const float* __restrict__ sumsq, float* __restrict__ state, float max_norm,
                                                    float beta1, float beta2, float eps, float wd, float grad_div, float lr0, float lr1,
                                                    long lr_split) {
    const float t = state[0] + 1.f;
    const float lr_scale = state[1];
    const float total_norm = sqrtf(sumsq[0]) * grad_div;
    // a non-finite gradient (NaN / Inf loss, A2/engine.py:44-49) must not reach the parameters or the moments: the whole update is
    // skipped (grid-uniform branch) and adamw_finish_kernel latches the event in state[3] for the host to read when it next looks
    if (!(fabsf(total_norm) <= 3.0e38f)) return;
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
    // lr == NULL: two learning rates split at element lr_split (the arena is ordered [everything else | backbone], A2/main.py:157-183):
    // one 150 MB stream less than the per-element table (lr_split is a multiple of 4 or the table is used)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        float4 ll;
        if (lr) ll = l4[i];
        else { const float l = (4 * i < lr_split) ? lr0 : lr1; ll = make_float4(l, l, l, l); }
        upd(pp.x, gg.x, mm.x, vv.x, ll.x);
        upd(pp.y, gg.y, mm.y, vv.y, ll.y);
        upd(pp.z, gg.z, mm.z, vv.z, ll.z);
        upd(pp.w, gg.w, mm.w, vv.w, ll.w);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) upd(p[i], g[i], m[i], v[i], lr ? lr[i] : (i < lr_split ? lr0 : lr1));
    // every block has read state[0] / sumsq before block 0 writes them only if ... they are written by a SEPARATE tiny
    // kernel (adamw_finish) launched after this one -- see cdetr_adamw_step.
}

__global__ void adamw_finish_kernel(const float* __restrict__ sumsq, float* __restrict__ state, float grad_div) {
    const float total_norm = sqrtf(sumsq[0]) * grad_div;
    state[2] = total_norm;                   // total gradient norm (of the averaged gradient), for logging
    if (fabsf(total_norm) <= 3.0e38f) state[0] += 1.f;
    else state[3] += 1.f;                    // skipped (non-finite) steps since the host last cleared it
}

__global__ __launch_bounds__(256) void relu_mask_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                        float* __restrict__ dz, __bf16* __restrict__ dz16, long n, float scale) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(y)[i];
        float4 b = reinterpret_cast<const float4*>(dy)[i];
        b.x = a.x > 0.f ? b.x * scale : 0.f;
        b.y = a.y > 0.f ? b.y * scale : 0.f;
        b.z = a.z > 0.f ? b.z * scale : 0.f;
        b.w = a.w > 0.f ? b.w * scale : 0.f;
        reinterpret_cast<float4*>(dz)[i] = b;
        if (dz16) {      // bf16 twin for the plain-bf16 weight gradient that consumes dz
            const f32x2 lo = {b.x, b.y}, hi = {b.z, b.w};
            uint2 t;
            t.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
            t.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
            reinterpret_cast<uint2*>(dz16)[i] = t;
        }
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
            const float v = y[i] > 0.f ? dy[i] * scale : 0.f;
            dz[i] = v;
            if (dz16) dz16[i] = (__bf16)v;
        }
}

inline int grid_for(long n4) {
    long b = (n4 + 255) / 256;
    static_assert(CDETR_SUMSQ_MAX_BLOCKS == 2048, "cdetr_sumsq's workspace holds one partial per block of this grid");
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}


// ---------------------------------------------------------------------------------------------------- weight mirrors
// Wt[c][tap][o] = W[o][tap][c] * scale[o] for a table of weights in ONE launch: block -> item by binary search over the
// tile prefix sums, 32x32 tile through LDS (row stride 33: conflict-free both ways), coalesced 128-byte rows in and out.
__global__ __launch_bounds__(256) void weight_mirror_kernel(const cdetr_mirror_item* __restrict__ items, const int n_items) {
    __shared__ float tile[32][33];
    const int b = blockIdx.x;
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].tile0 <= b) lo = mid; else hi = mid - 1;
    }
    const cdetr_mirror_item it = items[lo];
    const int tilesC = (it.C + 31) >> 5, tilesR = (it.R + 31) >> 5;
    int t = b - it.tile0;
    const int tap = t / (tilesR * tilesC);
    t -= tap * tilesR * tilesC;
    const int r0 = (t / tilesC) * 32, c0 = (t % tilesC) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    __bf16* sp = reinterpret_cast<__bf16*>(it.dst_split);
    // the four rows of a thread are fetched as one batch of unconditional loads (clamped address, masked use): a load under a
    // per-element condition is its own basic block and costs one memory round trip each
    float vin[4], sin_[4];
    const int cc = min(c0 + tx, it.C - 1);
#pragma unroll
    for (int p = 0; p < 4; ++p) vin[p] = it.src[((long)min(r0 + ty + 8 * p, it.R - 1) * it.taps + tap) * it.C + cc];
    if (it.scale) {
#pragma unroll
        for (int p = 0; p < 4; ++p) sin_[p] = it.scale[min(r0 + ty + 8 * p, it.R - 1)];
    } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) sin_[p] = 1.f;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = r0 + ty + 8 * p, c = c0 + tx;
        const float v = (r < it.R && c < it.C) ? (it.scale ? vin[p] * sin_[p] : vin[p]) : 0.f;
        tile[ty + 8 * p][tx] = v;
        if (!it.transpose && sp && r < it.R && c < it.C) {      // forward image: row r, k = tap*C + c (C % 32 == 0: one group per tile row)
            const __bf16 h = (__bf16)v;
            __bf16* g = sp + (long)r * it.taps * it.C * 2 + (long)((tap * it.C + c0) >> 5) * 64;
            g[tx] = h;
            g[32 + tx] = (__bf16)(v - (float)h);
        }
    }
    if (!it.transpose) return;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = c0 + ty + 8 * p, r = r0 + tx;
        if (r < it.R && c < it.C) {
            const float v = tile[tx][ty + 8 * p];
            if (it.dst) it.dst[((long)c * it.taps + tap) * it.R + r] = v;
            if (it.dst_hi) reinterpret_cast<__bf16*>(it.dst_hi)[((long)c * it.taps + tap) * it.R + r] = (__bf16)v;
            if (sp) {                                            // transposed image: row c, k = tap*R + r (R % 32 == 0)
                const __bf16 h = (__bf16)v;
                __bf16* g = sp + (long)c * it.taps * it.R * 2 + (long)((tap * it.R + r0) >> 5) * 64;
                g[tx] = h;
                g[32 + tx] = (__bf16)(v - (float)h);
            }
        }
    }
}
// The FORWARD images alone (transpose = 0: W * scale pre-split into groups [hi 32 | lo 32], same row-major order as W): a pure stream -- every
// thread turns 4 consecutive weights (16-byte load; C % 32 == 0, so they share a row and a group) into 4 hi + 4 lo values (two 8-byte stores).  The
// tiled kernel above did this with 4-byte loads and 2-byte stores (90 us for the 38 M weights of the model, in the step's critical path); item.tile0
// counts blocks of 4096 weights here.  Bit-identical images.
constexpr int WIMG_BLOCK = 4096;
__global__ __launch_bounds__(256) void weight_image_kernel(const cdetr_mirror_item* __restrict__ items, const int n_items) {
    const int b = blockIdx.x;
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].tile0 <= b) lo = mid; else hi = mid - 1;
    }
    const cdetr_mirror_item it = items[lo];
    const long K = (long)it.taps * it.C, total = (long)it.R * K;
    const long e0 = (long)(b - it.tile0) * WIMG_BLOCK + threadIdx.x * 4;
    __bf16* sp = reinterpret_cast<__bf16*>(it.dst_split);
    const bool al = (reinterpret_cast<uintptr_t>(it.src) & 15) == 0;       // block-uniform
    float4 v[4];
    float sc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                                // clamped, unconditional: the four loads (and scales) of a thread are in flight together
        const long e = min(e0 + u * 1024, total - 4);
        if (al) v[u] = *reinterpret_cast<const float4*>(it.src + e);
        else v[u] = make_float4(it.src[e], it.src[e + 1], it.src[e + 2], it.src[e + 3]);      // (a parameter view at an odd offset of its arena)
        sc[u] = it.scale ? it.scale[e / K] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long e = e0 + u * 1024;
        if (e >= total) continue;
        const float x[4] = {it.scale ? v[u].x * sc[u] : v[u].x, it.scale ? v[u].y * sc[u] : v[u].y, it.scale ? v[u].z * sc[u] : v[u].z,
                            it.scale ? v[u].w * sc[u] : v[u].w};
        __bf16 h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = (__bf16)x[j];
            l[j] = (__bf16)(x[j] - (float)h[j]);
        }
        __bf16* g = sp + (e >> 5) * 64 + (e & 31);
        *reinterpret_cast<uint2*>(g) = *reinterpret_cast<const uint2*>(h);
        *reinterpret_cast<uint2*>(g + 32) = *reinterpret_cast<const uint2*>(l);
    }
}

// ------------------------------------------------------------------------------------------------ GroupNorm on NHWC rows
// nn.GroupNorm(G, C) of A2/models/anchor_detr.py:86-92 on the NHWC activation [B][P = h*w][C]: a group is CG = C / G consecutive
// channels of every pixel of one image.  One workgroup per (image, group): pass 1 sums x and x^2 over its P x CG slab (16-byte loads
// when CG % 4 == 0; the slab is a few tens of KB and stays in L2), pass 2 normalises.  Replaces permute -> contiguous -> native
// group_norm -> permute -> contiguous (and the same in backward): 2 launches instead of ~10.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {        // NT threads; result in every thread
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}
constexpr int GN_NT = 1024;      // forward: 16 waves per (image, group) slab (32-byte runs at a 1 KB stride, latency bound): 22 -> 17 us
constexpr int GN_NT_BWD = 256;   // backward: 1024 threads measured slower (39 vs 36 us: the per-channel LDS atomics)

__global__ __launch_bounds__(GN_NT) void gn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int P, int C, int G, float eps) {
    __shared__ float red[GN_NT / 64];
    const int CG = C / G, q4 = CG >> 2;                       // float4 per pixel of the group
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const float* xb = x + (long)b * P * C + g * CG;
    float* yb = y + (long)b * P * C + g * CG;
    const int n4 = P * q4;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < n4; i += GN_NT) {
        const int p = i / q4, c4 = i - p * q4;
        const float4 v = *reinterpret_cast<const float4*>(xb + (long)p * C + c4 * 4);
        s1 += (v.x + v.y) + (v.z + v.w);
    }
    const float inv = 1.f / ((float)P * CG);
    const float mu = block_sum<GN_NT>(s1, red) * inv;
    for (int i = threadIdx.x; i < n4; i += GN_NT) {             // centred second moment: no cancellation when |mean| >> std (slab is L2-hot)
        const int p = i / q4, c4 = i - p * q4;
        const float4 v = *reinterpret_cast<const float4*>(xb + (long)p * C + c4 * 4);
        const float a = v.x - mu, b2 = v.y - mu, c2 = v.z - mu, e2 = v.w - mu;
        s2 += (a * a + b2 * b2) + (c2 * c2 + e2 * e2);
    }
    const float var = block_sum<GN_NT>(s2, red) * inv;
    const float rs = rsqrtf(var + eps);
    if (threadIdx.x == 0) { mean[blockIdx.x] = mu; rstd[blockIdx.x] = rs; }
    for (int i = threadIdx.x; i < n4; i += GN_NT) {
        const int p = i / q4, c4 = i - p * q4;
        const float4 v = *reinterpret_cast<const float4*>(xb + (long)p * C + c4 * 4);
        const float4 gm = *reinterpret_cast<const float4*>(gamma + g * CG + c4 * 4);
        const float4 bt = *reinterpret_cast<const float4*>(beta + g * CG + c4 * 4);
        float4 o;
        o.x = (v.x - mu) * rs * gm.x + bt.x; o.y = (v.y - mu) * rs * gm.y + bt.y;
        o.z = (v.z - mu) * rs * gm.z + bt.z; o.w = (v.w - mu) * rs * gm.w + bt.w;
        *reinterpret_cast<float4*>(yb + (long)p * C + c4 * 4) = o;
    }
}

// dx = rstd * (dy*gamma - mean_g(dy*gamma) - xhat * mean_g(dy*gamma*xhat)); dgamma[c] += sum dy*xhat, dbeta[c] += sum dy (atomics:
// one workgroup per (image, group), the channel sums of a group come from B workgroups).
__global__ __launch_bounds__(GN_NT_BWD) void gn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, float* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int P, int C, int G) {
    __shared__ float red[GN_NT_BWD / 64];
    __shared__ float cacc[2][64];                              // per-channel dgamma / dbeta partials of the group (CG <= 64)
    const int CG = C / G, q4 = CG >> 2;
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const float* xb = x + (long)b * P * C + g * CG;
    const float* db = dy + (long)b * P * C + g * CG;
    float* ob = dx + (long)b * P * C + g * CG;
    const float mu = mean[blockIdx.x], rs = rstd[blockIdx.x];
    const int n4 = P * q4;
    if (threadIdx.x < 64) { cacc[0][threadIdx.x] = 0.f; cacc[1][threadIdx.x] = 0.f; }
    // this thread always visits the same channel quad when GN_NT_BWD % q4 == 0 (q4 = 1, 2, 4, ...): keep its channel sums in registers
    const bool fixed = (GN_NT_BWD % q4) == 0;
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
    float s1 = 0.f, s2 = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < n4; i += GN_NT_BWD) {
        const int p = i / q4, c4 = i - p * q4;
        const float4 xv = *reinterpret_cast<const float4*>(xb + (long)p * C + c4 * 4);
        const float4 dv = *reinterpret_cast<const float4*>(db + (long)p * C + c4 * 4);
        const float4 gm = *reinterpret_cast<const float4*>(gamma + g * CG + c4 * 4);
        const float4 xh = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        const float4 dg = make_float4(dv.x * gm.x, dv.y * gm.y, dv.z * gm.z, dv.w * gm.w);
        s1 += (dg.x + dg.y) + (dg.z + dg.w);
        s2 += (dg.x * xh.x + dg.y * xh.y) + (dg.z * xh.z + dg.w * xh.w);
        if (fixed) {
            ag.x += dv.x * xh.x; ag.y += dv.y * xh.y; ag.z += dv.z * xh.z; ag.w += dv.w * xh.w;
            ab.x += dv.x; ab.y += dv.y; ab.z += dv.z; ab.w += dv.w;
        } else {
            atomicAdd(&cacc[0][c4 * 4 + 0], dv.x * xh.x); atomicAdd(&cacc[0][c4 * 4 + 1], dv.y * xh.y);
            atomicAdd(&cacc[0][c4 * 4 + 2], dv.z * xh.z); atomicAdd(&cacc[0][c4 * 4 + 3], dv.w * xh.w);
            atomicAdd(&cacc[1][c4 * 4 + 0], dv.x); atomicAdd(&cacc[1][c4 * 4 + 1], dv.y);
            atomicAdd(&cacc[1][c4 * 4 + 2], dv.z); atomicAdd(&cacc[1][c4 * 4 + 3], dv.w);
        }
    }
    if (fixed) {
        const int c4 = threadIdx.x % q4;
        atomicAdd(&cacc[0][c4 * 4 + 0], ag.x); atomicAdd(&cacc[0][c4 * 4 + 1], ag.y);
        atomicAdd(&cacc[0][c4 * 4 + 2], ag.z); atomicAdd(&cacc[0][c4 * 4 + 3], ag.w);
        atomicAdd(&cacc[1][c4 * 4 + 0], ab.x); atomicAdd(&cacc[1][c4 * 4 + 1], ab.y);
        atomicAdd(&cacc[1][c4 * 4 + 2], ab.z); atomicAdd(&cacc[1][c4 * 4 + 3], ab.w);
    }
    const float inv = 1.f / ((float)P * CG);
    const float m1 = block_sum<GN_NT_BWD>(s1, red) * inv, m2 = block_sum<GN_NT_BWD>(s2, red) * inv;      // (the barriers inside also publish cacc)
    for (int i = threadIdx.x; i < n4; i += GN_NT_BWD) {
        const int p = i / q4, c4 = i - p * q4;
        const float4 xv = *reinterpret_cast<const float4*>(xb + (long)p * C + c4 * 4);
        const float4 dv = *reinterpret_cast<const float4*>(db + (long)p * C + c4 * 4);
        const float4 gm = *reinterpret_cast<const float4*>(gamma + g * CG + c4 * 4);
        float4 o;
        o.x = rs * (dv.x * gm.x - m1 - (xv.x - mu) * rs * m2); o.y = rs * (dv.y * gm.y - m1 - (xv.y - mu) * rs * m2);
        o.z = rs * (dv.z * gm.z - m1 - (xv.z - mu) * rs * m2); o.w = rs * (dv.w * gm.w - m1 - (xv.w - mu) * rs * m2);
        *reinterpret_cast<float4*>(ob + (long)p * C + c4 * 4) = o;
    }
    if (threadIdx.x < CG) {
        atomicAdd(dgamma + g * CG + threadIdx.x, cacc[0][threadIdx.x]);
        atomicAdd(dbeta + g * CG + threadIdx.x, cacc[1][threadIdx.x]);
    }
}


// ---- GroupNorm with the pixels of an image SPLIT over workgroups (C == 256: one wave = one full 1 KB pixel row) -------------------------
// The one-workgroup-per-(image, group) kernels above read 32-byte runs at a 1 KB stride on 64 workgroups (18 / 36 us for the projection's
// 2 x 2500 x 256 map, on the step's main chain).  Here a workgroup owns GN_CH consecutive pixels x all channels (coalesced rows, B * P / GN_CH
// workgroups), and the group statistics cross workgroups through a small workspace: launch 1 leaves per-(image, chunk, group) partials
// -- forward: (count, mean, centred second moment) of the chunk, merged with Chan's update in chunk order, so the variance stays a centred one;
// backward: the two plain sums -- launch 2 merges them (every workgroup for itself, same order: same result everywhere) and applies.
constexpr int GN_CH = 32;         // pixels per workgroup: 4 waves x 8 rows, all 8 row loads of a wave in flight together

__device__ __forceinline__ float gn_group_total(float v, const int cgq, float (*red)[64], const int w, const int lane) {
    for (int m = 1; m < cgq; m <<= 1) v += __shfl_xor(v, m);       // the lanes (channel quads) of one group
    __syncthreads();
    red[w][lane] = v;
    __syncthreads();
    const int l0 = lane & ~(cgq - 1);
    return (red[0][l0] + red[1][l0]) + (red[2][l0] + red[3][l0]);   // the four waves (pixel rows), fixed order
}

__global__ __launch_bounds__(256) void gn_split_stats_kernel(const float* __restrict__ x, float* __restrict__ ws, int P, int G, int nchunk) {
    constexpr int C = 256;
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.y, ch = blockIdx.x;
    const int p0 = ch * GN_CH, pn = min(GN_CH, P - p0);
    const int cgq = (C / G) >> 2;
    const float* xb = x + ((long)b * P + p0) * C + lane * 4;
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(xb + (long)min(w + 4 * j, pn - 1) * C);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (w + 4 * j < pn) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float n = (float)pn * (float)(C / G);
    const float mu = gn_group_total(s, cgq, red, w, lane) / n;
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (w + 4 * j < pn) {
            const float a = v[j].x - mu, b2 = v[j].y - mu, c2 = v[j].z - mu, e2 = v[j].w - mu;
            m2 += (a * a + b2 * b2) + (c2 * c2 + e2 * e2);
        }
    const float M2 = gn_group_total(m2, cgq, red, w, lane);
    if (w == 0 && (lane & (cgq - 1)) == 0) {
        float* o = ws + (((long)b * nchunk + ch) * G + lane / cgq) * 3;
        o[0] = n; o[1] = mu; o[2] = M2;
    }
}

__global__ __launch_bounds__(256) void gn_split_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ ws, float* __restrict__ y, float* __restrict__ mean,
                                                             float* __restrict__ rstd, int P, int G, int nchunk, float eps) {
    constexpr int C = 256;
    __shared__ float comb[8][64][3];
    __shared__ float stat[64][2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.y, ch = blockIdx.x;
    const int p0 = ch * GN_CH, pn = min(GN_CH, P - p0);
    const int cgq = (C / G) >> 2;
    const float* xb = x + ((long)b * P + p0) * C + lane * 4;
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(xb + (long)min(w + 4 * j, pn - 1) * C);
    const float4 gm = *reinterpret_cast<const float4*>(gamma + lane * 4), bt = *reinterpret_cast<const float4*>(beta + lane * 4);
    // merge the chunks of this image: NS = 256 / G sub-sequences (chunk s, s + NS, ...), then the NS partial results in order
    const int NS = 256 / G, g = threadIdx.x % G, sub = threadIdx.x / G;
    float n = 0.f, mu = 0.f, M2 = 0.f;
    for (int c = sub; c < nchunk; c += NS) {
        const float* pw = ws + (((long)b * nchunk + c) * G + g) * 3;
        const float nc = pw[0], mc = pw[1], Mc = pw[2];
        const float nt = n + nc, d = mc - mu;
        mu += d * (nc / nt);
        M2 += Mc + d * d * (n * nc / nt);
        n = nt;
    }
    comb[sub][g][0] = n; comb[sub][g][1] = mu; comb[sub][g][2] = M2;
    __syncthreads();
    if (threadIdx.x < G) {
        float tn = 0.f, tm = 0.f, tM = 0.f;
        for (int s2 = 0; s2 < NS; ++s2) {
            const float nc = comb[s2][g][0], mc = comb[s2][g][1], Mc = comb[s2][g][2];
            if (nc > 0.f) {
                const float nt = tn + nc, d = mc - tm;
                tm += d * (nc / nt);
                tM += Mc + d * d * (tn * nc / nt);
                tn = nt;
            }
        }
        const float rs = rsqrtf(tM / tn + eps);
        stat[g][0] = tm; stat[g][1] = rs;
        if (ch == 0) { mean[b * G + g] = tm; rstd[b * G + g] = rs; }
    }
    __syncthreads();
    const float m = stat[lane / cgq][0], rs = stat[lane / cgq][1];
    float* yb = y + ((long)b * P + p0) * C + lane * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (w + 4 * j < pn) {
            float4 o;
            o.x = (v[j].x - m) * rs * gm.x + bt.x; o.y = (v[j].y - m) * rs * gm.y + bt.y;
            o.z = (v[j].z - m) * rs * gm.z + bt.z; o.w = (v[j].w - m) * rs * gm.w + bt.w;
            *reinterpret_cast<float4*>(yb + (long)(w + 4 * j) * C) = o;
        }
}

__global__ __launch_bounds__(256) void gn_split_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ gamma, float* __restrict__ ws,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, int P, int G, int nchunk) {
    constexpr int C = 256;
    __shared__ float red[4][64];
    __shared__ float cred[4][64][8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.y, ch = blockIdx.x;
    const int p0 = ch * GN_CH, pn = min(GN_CH, P - p0);
    const int cgq = (C / G) >> 2;
    const long base = ((long)b * P + p0) * C + lane * 4;
    float4 xv[8], dv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long o = base + (long)min(w + 4 * j, pn - 1) * C;
        xv[j] = *reinterpret_cast<const float4*>(x + o);
        dv[j] = *reinterpret_cast<const float4*>(dy + o);
    }
    const float4 gm = *reinterpret_cast<const float4*>(gamma + lane * 4);
    const float mu = mean[b * G + lane / cgq], rs = rstd[b * G + lane / cgq];
    float s1 = 0.f, s2 = 0.f;
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (w + 4 * j < pn) {
            const float4 xh = make_float4((xv[j].x - mu) * rs, (xv[j].y - mu) * rs, (xv[j].z - mu) * rs, (xv[j].w - mu) * rs);
            const float4 dg = make_float4(dv[j].x * gm.x, dv[j].y * gm.y, dv[j].z * gm.z, dv[j].w * gm.w);
            s1 += (dg.x + dg.y) + (dg.z + dg.w);
            s2 += (dg.x * xh.x + dg.y * xh.y) + (dg.z * xh.z + dg.w * xh.w);
            ag.x += dv[j].x * xh.x; ag.y += dv[j].y * xh.y; ag.z += dv[j].z * xh.z; ag.w += dv[j].w * xh.w;
            ab.x += dv[j].x; ab.y += dv[j].y; ab.z += dv[j].z; ab.w += dv[j].w;
        }
    const float S1 = gn_group_total(s1, cgq, red, w, lane);
    const float S2 = gn_group_total(s2, cgq, red, w, lane);
    if (w == 0 && (lane & (cgq - 1)) == 0) {
        float* o = ws + (((long)b * nchunk + ch) * G + lane / cgq) * 2;
        o[0] = S1; o[1] = S2;
    }
    cred[w][lane][0] = ag.x; cred[w][lane][1] = ag.y; cred[w][lane][2] = ag.z; cred[w][lane][3] = ag.w;
    cred[w][lane][4] = ab.x; cred[w][lane][5] = ab.y; cred[w][lane][6] = ab.z; cred[w][lane][7] = ab.w;
    __syncthreads();
    // 512 channel sums of the chunk: thread t adds the four waves' partials of (quad t / 8... ) -- two per thread
    for (int e = threadIdx.x; e < 512; e += 256) {
        const int qd = e >> 3, k = e & 7;
        const float t = (cred[0][qd][k] + cred[1][qd][k]) + (cred[2][qd][k] + cred[3][qd][k]);
        atomicAdd((k < 4 ? dgamma : dbeta) + qd * 4 + (k & 3), t);
    }
}

__global__ __launch_bounds__(256) void gn_split_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ ws,
                                                                 float* __restrict__ dx, int P, int G, int nchunk) {
    constexpr int C = 256;
    __shared__ float comb[8][64][2];
    __shared__ float stat[64][2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.y, ch = blockIdx.x;
    const int p0 = ch * GN_CH, pn = min(GN_CH, P - p0);
    const int cgq = (C / G) >> 2;
    const long base = ((long)b * P + p0) * C + lane * 4;
    float4 xv[8], dv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long o = base + (long)min(w + 4 * j, pn - 1) * C;
        xv[j] = *reinterpret_cast<const float4*>(x + o);
        dv[j] = *reinterpret_cast<const float4*>(dy + o);
    }
    const float4 gm = *reinterpret_cast<const float4*>(gamma + lane * 4);
    const int NS = 256 / G, g = threadIdx.x % G, sub = threadIdx.x / G;
    float a1 = 0.f, a2 = 0.f;
    for (int c = sub; c < nchunk; c += NS) {
        const float* pw = ws + (((long)b * nchunk + c) * G + g) * 2;
        a1 += pw[0]; a2 += pw[1];
    }
    comb[sub][g][0] = a1; comb[sub][g][1] = a2;
    __syncthreads();
    if (threadIdx.x < G) {
        float t1 = 0.f, t2 = 0.f;
        for (int s2 = 0; s2 < NS; ++s2) { t1 += comb[s2][g][0]; t2 += comb[s2][g][1]; }
        const float inv = 1.f / ((float)P * (float)(C / G));
        stat[g][0] = t1 * inv; stat[g][1] = t2 * inv;
    }
    __syncthreads();
    const float m1 = stat[lane / cgq][0], m2 = stat[lane / cgq][1];
    const float mu = mean[b * G + lane / cgq], rs = rstd[b * G + lane / cgq];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (w + 4 * j < pn) {
            float4 o;
            o.x = rs * (dv[j].x * gm.x - m1 - (xv[j].x - mu) * rs * m2); o.y = rs * (dv[j].y * gm.y - m1 - (xv[j].y - mu) * rs * m2);
            o.z = rs * (dv[j].z * gm.z - m1 - (xv[j].z - mu) * rs * m2); o.w = rs * (dv[j].w * gm.w - m1 - (xv[j].w - mu) * rs * m2);
            *reinterpret_cast<float4*>(dx + base + (long)(w + 4 * j) * C) = o;
        }
}

}  // namespace

extern "C" int cdetr_sumsq(const float* g, int64_t n, float* out, float* workspace, void* stream) {
    CDETR_CHECK_ARG(g && out && workspace && n >= 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0, "cdetr_sumsq: bad args");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = grid_for(n >> 2);
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, st, g, (long)n, workspace);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, workspace, nb, out);
    return cdetr_launch_status("cdetr_sumsq");
}

extern "C" int cdetr_adamw_step2(float* p, const float* g, float* m, float* v, const float* lr, float lr0, float lr1, int64_t lr_split,
                                 int64_t n, const float* sumsq, float* state, float max_norm, float beta1, float beta2, float eps,
                                 float weight_decay, float grad_div, void* stream) {
    CDETR_CHECK_ARG(p && g && m && v && sumsq && state && n >= 0, "cdetr_adamw_step2: null pointer");
    CDETR_CHECK_ARG(lr || (lr_split >= 0 && (lr_split & 3) == 0), "cdetr_adamw_step2: without a per-element table lr_split must be a multiple of 4");
    CDETR_CHECK_ARG(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                      reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(lr)) & 15) == 0, "cdetr_adamw_step2: arenas must be 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, st, p, g, m, v, lr, (long)n, sumsq, state, max_norm, beta1, beta2,
                       eps, weight_decay, grad_div, lr0, lr1, (long)lr_split);
    hipLaunchKernelGGL(adamw_finish_kernel, dim3(1), dim3(1), 0, st, sumsq, state, grad_div);
    return cdetr_launch_status("cdetr_adamw_step2");
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
                       beta1, beta2, eps, weight_decay, grad_div, 0.f, 0.f, 0L);
    hipLaunchKernelGGL(adamw_finish_kernel, dim3(1), dim3(1), 0, st, sumsq, state, grad_div);
    return cdetr_launch_status("cdetr_adamw_step");
}

extern "C" int cdetr_relu_mask2(const float* y, const float* dy, float* dz, void* dz16, int64_t n, float scale, void* stream) {
    CDETR_CHECK_ARG(y && dy && dz && n >= 0, "cdetr_relu_mask2: null pointer");
    CDETR_CHECK_ARG(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dz)) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(dz16) & 7) == 0, "cdetr_relu_mask2: buffers must be 16-byte (twin: 8-byte) aligned");
    if (n == 0) return CDETR_OK;
    hipLaunchKernelGGL(relu_mask_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y, dy, dz,
                       reinterpret_cast<__bf16*>(dz16), (long)n, scale);
    return cdetr_launch_status("cdetr_relu_mask2");
}

extern "C" int cdetr_relu_mask(const float* y, const float* dy, float* dz, int64_t n, float scale, void* stream) {
    CDETR_CHECK_ARG(y && dy && dz && n >= 0, "cdetr_relu_mask: bad args");
    CDETR_CHECK_ARG(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dz)) & 15) == 0,
                    "cdetr_relu_mask: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(relu_mask_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y, dy, dz,
                       (__bf16*)nullptr, (long)n, scale);
    return cdetr_launch_status("cdetr_relu_mask");
}

// =====================================================================================================================
// Normalisation / broadcast kernels of the transformer layers (HBM-bound, one wave per row of C <= 1024 channels)
// =====================================================================================================================
namespace {

constexpr int LN_MAXV = 4;   // float4 per lane: C <= 64 * 4 * 4 = 1024
constexpr int LN_RB = 4;     // rows a wave of the backward kernel keeps in flight (C == 256)

// y = (x - mean) * rstd * gamma + beta ; saves mean / rstd per row (A2/models/transformer.py norm1 / norm2 / ffn.norm2)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows, int C, float eps,
                                                     const float* __restrict__ a1, const float* __restrict__ a2,
                                                     float* __restrict__ o1, float* __restrict__ o2) {
    const int lane = threadIdx.x & 63;
    const int nv = C >> 8;   // float4 per lane (C multiple of 256) -- checked on the host
    if (nv == 1) {
        // C == 256 (every LayerNorm of this model; round 6): the row, gamma, beta and the optional addends are ALL requested before the first
        // reduction -- one memory round trip per row.  The general form below fetches gamma / beta / addends after the two reductions: a second
        // (and third) trip in a kernel that is nothing but a chain of them (~5 us per launch, 30 launches per step).  Absent addends read the
        // row itself (a valid address, result unused), so there is no branch around a load.
        const float4 g4 = reinterpret_cast<const float4*>(gamma)[lane];
        const float4 b4 = reinterpret_cast<const float4*>(beta)[lane];
        for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
            const float4 v = reinterpret_cast<const float4*>(x + (long)row * C)[lane];
            const float4 p1 = reinterpret_cast<const float4*>((a1 ? a1 : x) + (long)row * C)[lane];
            const float4 p2 = reinterpret_cast<const float4*>((a2 ? a2 : x) + (long)row * C)[lane];
            const float mu = wave_sum((v.x + v.y) + (v.z + v.w)) / C;
            const float a = v.x - mu, b = v.y - mu, c = v.z - mu, dd = v.w - mu;
            const float rs = rsqrtf(wave_sum((a * a + b * b) + (c * c + dd * dd)) / C + eps);
            if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
            float4 o;
            o.x = (v.x - mu) * rs * g4.x + b4.x; o.y = (v.y - mu) * rs * g4.y + b4.y;
            o.z = (v.z - mu) * rs * g4.z + b4.z; o.w = (v.w - mu) * rs * g4.w + b4.w;
            reinterpret_cast<float4*>(y + (long)row * C)[lane] = o;
            if (a1) reinterpret_cast<float4*>(o1 + (long)row * C)[lane] = make_float4(o.x + p1.x, o.y + p1.y, o.z + p1.z, o.w + p1.w);
            if (a2) reinterpret_cast<float4*>(o2 + (long)row * C)[lane] = make_float4(o.x + p2.x, o.y + p2.y, o.z + p2.z, o.w + p2.w);
        }
        return;
    }
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const float4* xr = reinterpret_cast<const float4*>(x + (long)row * C);
        float4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i)
            if (i < nv) { v[i] = xr[lane + 64 * i]; s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
        const float mu = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i)
            if (i < nv) {
                const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, dd = v[i].w - mu;
                q += (a * a + b * b) + (c * c + dd * dd);
            }
        const float rs = rsqrtf(wave_sum(q) / C + eps);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
        float4* yr = reinterpret_cast<float4*>(y + (long)row * C);
        // optional fused consumers: o1 = y + a1, o2 = y + a2 (the positional adds that follow a LayerNorm in the decoder); their operands
        // are requested before the normalised row is formed
        float4 p1[LN_MAXV], p2[LN_MAXV];
        if (a1) {
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i)
                if (i < nv) p1[i] = reinterpret_cast<const float4*>(a1 + (long)row * C)[lane + 64 * i];
        }
        if (a2) {
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i)
                if (i < nv) p2[i] = reinterpret_cast<const float4*>(a2 + (long)row * C)[lane + 64 * i];
        }
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i)
            if (i < nv) {
                const float4 g4 = reinterpret_cast<const float4*>(gamma)[lane + 64 * i];
                const float4 b4 = reinterpret_cast<const float4*>(beta)[lane + 64 * i];
                float4 o;
                o.x = (v[i].x - mu) * rs * g4.x + b4.x; o.y = (v[i].y - mu) * rs * g4.y + b4.y;
                o.z = (v[i].z - mu) * rs * g4.z + b4.z; o.w = (v[i].w - mu) * rs * g4.w + b4.w;
                yr[lane + 64 * i] = o;
                if (a1) reinterpret_cast<float4*>(o1 + (long)row * C)[lane + 64 * i] = make_float4(o.x + p1[i].x, o.y + p1[i].y, o.z + p1[i].z, o.w + p1[i].w);
                if (a2) reinterpret_cast<float4*>(o2 + (long)row * C)[lane + 64 * i] = make_float4(o.x + p2[i].x, o.y + p2[i].y, o.z + p2[i].z, o.w + p2[i].w);
            }
    }
}

// dx = rstd * (dy*g - mean_c(dy*g) - xhat * mean_c(dy*g*xhat)) (+ add) ; dgamma += sum_rows dy*xhat ; dbeta += sum_rows dy
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ add,
                                                     float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     int rows, int C, const float* __restrict__ g1 = nullptr,
                                                     const float* __restrict__ g2 = nullptr, float* __restrict__ acc1 = nullptr,
                                                     float* __restrict__ acc2 = nullptr, const float* __restrict__ Br = nullptr,
                                                     const float* __restrict__ Bc = nullptr, float sr = 0.f, float sc = 0.f, int bH = 1, int bW = 1,
                                                     __bf16* __restrict__ dx16 = nullptr) {
    // dx16 (C == 256 only): a bf16 twin of dx from the same pass -- the A operand of the plain-bf16 data-gradient GEMM that follows (cdetr_gemm_desc.A16)
    // Br / Bc (with g1; rows = N * bH * bW): two broadcast addends, dy_eff += sr * Br[n, x] + sc * Bc[n, y] (row = (n, y, x)) -- the encoder's
    // cdetr_bcast_add2_sum pass folded in the same way
    // g1 / g2 (C == 256 only): further addends of the incoming gradient, dy_eff = dy + g1 + g2, with acc1 += g1, acc2 += g2 in place -- the
    // cdetr_grad_merge pass that used to precede this kernel in the decoder's backward (its output had no other reader)
    __shared__ float red[2][4][1024];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nv = C >> 8;
    float4 ag[LN_MAXV], ab[LN_MAXV], g4[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[i] = ag[i];
        if (i < nv) g4[i] = reinterpret_cast<const float4*>(gamma)[lane + 64 * i];
    }
    // C == 256 (every LayerNorm of this model): a wave takes its rows FOUR at a time -- the 12 loads of a batch (x, dy, residual gradient) are
    // all in flight before the first reduction, so a wave pays one memory round trip per four rows instead of one per row (the kernel is a
    // chain of round trips: 157 workgroups x 8 rows per wave at the encoder shape).  Rows past the end are clamped and masked.
    if (nv == 1) {
        const float4 gg = g4[0];
        const int stride = gridDim.x * 4;
        for (int row0 = blockIdx.x * 4 + wid; row0 < rows; row0 += stride * LN_RB) {
            float4 xv[LN_RB], dv[LN_RB], av[LN_RB];
            float mu[LN_RB], rs[LN_RB];
#pragma unroll
            for (int b = 0; b < LN_RB; ++b) {
                const long r = min(row0 + b * stride, rows - 1);
                xv[b] = reinterpret_cast<const float4*>(x + r * C)[lane];
                dv[b] = reinterpret_cast<const float4*>(dy + r * C)[lane];
                av[b] = add ? reinterpret_cast<const float4*>(add + r * C)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
                mu[b] = mean[r];
                rs[b] = rstd[r];
            }
            if (g1) {            // merged addends (wave-uniform): their loads join the batch; the accumulators are updated for real rows only
                float4 u1[LN_RB], u2[LN_RB], c1[LN_RB], c2[LN_RB];
#pragma unroll
                for (int b = 0; b < LN_RB; ++b) {
                    const long r = min(row0 + b * stride, rows - 1);
                    u1[b] = reinterpret_cast<const float4*>(g1 + r * C)[lane];
                    u2[b] = g2 ? reinterpret_cast<const float4*>(g2 + r * C)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
                    c1[b] = acc1 ? reinterpret_cast<const float4*>(acc1 + r * C)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
                    c2[b] = acc2 ? reinterpret_cast<const float4*>(acc2 + r * C)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int b = 0; b < LN_RB; ++b) {
                    const int row = row0 + b * stride;
                    if (row >= rows) break;
                    dv[b].x += u1[b].x + u2[b].x; dv[b].y += u1[b].y + u2[b].y; dv[b].z += u1[b].z + u2[b].z; dv[b].w += u1[b].w + u2[b].w;
                    if (Br) {
                        const int xw = row % bW, t2 = row / bW, yh = t2 % bH, nn = t2 / bH;
                        const float4 br = reinterpret_cast<const float4*>(Br + ((long)nn * bW + xw) * C)[lane];
                        const float4 bc = reinterpret_cast<const float4*>(Bc + ((long)nn * bH + yh) * C)[lane];
                        dv[b].x += sr * br.x + sc * bc.x; dv[b].y += sr * br.y + sc * bc.y; dv[b].z += sr * br.z + sc * bc.z; dv[b].w += sr * br.w + sc * bc.w;
                    }
                    if (acc1) reinterpret_cast<float4*>(acc1 + (long)row * C)[lane] = make_float4(c1[b].x + u1[b].x, c1[b].y + u1[b].y, c1[b].z + u1[b].z, c1[b].w + u1[b].w);
                    if (acc2) reinterpret_cast<float4*>(acc2 + (long)row * C)[lane] = make_float4(c2[b].x + u2[b].x, c2[b].y + u2[b].y, c2[b].z + u2[b].z, c2[b].w + u2[b].w);
                }
            }
#pragma unroll
            for (int b = 0; b < LN_RB; ++b) {
                const int row = row0 + b * stride;
                if (row >= rows) break;               // wave-uniform
                const float4 xh = make_float4((xv[b].x - mu[b]) * rs[b], (xv[b].y - mu[b]) * rs[b], (xv[b].z - mu[b]) * rs[b], (xv[b].w - mu[b]) * rs[b]);
                const float4 dg = make_float4(dv[b].x * gg.x, dv[b].y * gg.y, dv[b].z * gg.z, dv[b].w * gg.w);
                const float s1 = (dg.x + dg.y) + (dg.z + dg.w);
                const float s2 = (dg.x * xh.x + dg.y * xh.y) + (dg.z * xh.z + dg.w * xh.w);
                ag[0].x += dv[b].x * xh.x; ag[0].y += dv[b].y * xh.y; ag[0].z += dv[b].z * xh.z; ag[0].w += dv[b].w * xh.w;
                ab[0].x += dv[b].x; ab[0].y += dv[b].y; ab[0].z += dv[b].z; ab[0].w += dv[b].w;
                const float m1 = wave_sum(s1) / C, m2 = wave_sum(s2) / C;
                float4 o;
                o.x = rs[b] * (dg.x - m1 - xh.x * m2) + av[b].x; o.y = rs[b] * (dg.y - m1 - xh.y * m2) + av[b].y;
                o.z = rs[b] * (dg.z - m1 - xh.z * m2) + av[b].z; o.w = rs[b] * (dg.w - m1 - xh.w * m2) + av[b].w;
                reinterpret_cast<float4*>(dx + (long)row * C)[lane] = o;
                if (dx16) {
                    const f32x2 v0 = {o.x, o.y}, v1 = {o.z, o.w};
                    uint2 h;
                    h.x = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, bf16x2));
                    h.y = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, bf16x2));
                    reinterpret_cast<uint2*>(dx16 + (long)row * C)[lane] = h;
                }
            }
        }
    } else
    for (int row = blockIdx.x * 4 + wid; row < rows; row += gridDim.x * 4) {
        const float4* xr = reinterpret_cast<const float4*>(x + (long)row * C);
        const float4* dr = reinterpret_cast<const float4*>(dy + (long)row * C);
        const float mu = mean[row], rs = rstd[row];
        float4 xh[LN_MAXV], dg[LN_MAXV], a4[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
        if (add) {      // the residual-branch gradient travels with the row's first round trip, not after the reductions
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i)
                if (i < nv) a4[i] = reinterpret_cast<const float4*>(add + (long)row * C)[lane + 64 * i];
        } else {
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i) a4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i)
            if (i < nv) {
                const float4 xv = xr[lane + 64 * i], dv = dr[lane + 64 * i];
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                dg[i] = make_float4(dv.x * g4[i].x, dv.y * g4[i].y, dv.z * g4[i].z, dv.w * g4[i].w);
                s1 += (dg[i].x + dg[i].y) + (dg[i].z + dg[i].w);
                s2 += (dg[i].x * xh[i].x + dg[i].y * xh[i].y) + (dg[i].z * xh[i].z + dg[i].w * xh[i].w);
                ag[i].x += dv.x * xh[i].x; ag[i].y += dv.y * xh[i].y; ag[i].z += dv.z * xh[i].z; ag[i].w += dv.w * xh[i].w;
                ab[i].x += dv.x; ab[i].y += dv.y; ab[i].z += dv.z; ab[i].w += dv.w;
            }
        const float m1 = wave_sum(s1) / C, m2 = wave_sum(s2) / C;
        float4* ox = reinterpret_cast<float4*>(dx + (long)row * C);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i)
            if (i < nv) {
                float4 o;
                o.x = rs * (dg[i].x - m1 - xh[i].x * m2); o.y = rs * (dg[i].y - m1 - xh[i].y * m2);
                o.z = rs * (dg[i].z - m1 - xh[i].z * m2); o.w = rs * (dg[i].w - m1 - xh[i].w * m2);
                o.x += a4[i].x; o.y += a4[i].y; o.z += a4[i].z; o.w += a4[i].w;
                ox[lane + 64 * i] = o;
            }
    }
    // reduce the 4 waves' partial dgamma / dbeta through LDS, one atomic per channel per workgroup
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
        if (i < nv) {
            const int c = (lane + 64 * i) * 4;
            red[0][wid][c] = ag[i].x; red[0][wid][c + 1] = ag[i].y; red[0][wid][c + 2] = ag[i].z; red[0][wid][c + 3] = ag[i].w;
            red[1][wid][c] = ab[i].x; red[1][wid][c + 1] = ab[i].y; red[1][wid][c + 2] = ab[i].z; red[1][wid][c + 3] = ab[i].w;
        }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(dgamma + c, (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]));
        atomicAdd(dbeta + c, (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]));
    }
}

// Encoder prologue: Qr[n,y,x,:] = X + Prow[n,x,:], Qc[n,y,x,:] = X + Pcol[n,y,:]  (A2/models/transformer.py:248-255)
__device__ __forceinline__ void posadd2_body(const float* __restrict__ X, const float* __restrict__ Prow, const float* __restrict__ Pcol,
                                             float* __restrict__ Qr, float* __restrict__ Qc, int N, int H, int W, int C4, int blk, int nblk) {
    const long total = (long)N * H * W * C4;
    for (long idx = (long)blk * 256 + threadIdx.x; idx < total; idx += (long)nblk * 256) {
        const int c = (int)(idx % C4);
        long t = idx / C4;
        const int xw = (int)(t % W); t /= W;
        const int yh = (int)(t % H);
        const int n = (int)(t / H);
        const float4 v = reinterpret_cast<const float4*>(X)[idx];
        const float4 pr = reinterpret_cast<const float4*>(Prow)[((long)n * W + xw) * C4 + c];
        const float4 pc = reinterpret_cast<const float4*>(Pcol)[((long)n * H + yh) * C4 + c];
        reinterpret_cast<float4*>(Qr)[idx] = make_float4(v.x + pr.x, v.y + pr.y, v.z + pr.z, v.w + pr.w);
        reinterpret_cast<float4*>(Qc)[idx] = make_float4(v.x + pc.x, v.y + pc.y, v.z + pc.z, v.w + pc.w);
    }
}
__global__ __launch_bounds__(256) void posadd2_kernel(const float* __restrict__ X, const float* __restrict__ Prow,
                                                      const float* __restrict__ Pcol, float* __restrict__ Qr, float* __restrict__ Qc,
                                                      int N, int H, int W, int C4) {
    posadd2_body(X, Prow, Pcol, Qr, Qc, N, H, W, C4, blockIdx.x, gridDim.x);
}

// Decoder glue (A2/models/transformer.py:366-403): O1 = T + A, O2 = T + B (B / O2 optional) in one pass.
__global__ __launch_bounds__(256) void add2_kernel(const float4* __restrict__ T, const float4* __restrict__ A, const float4* __restrict__ B,
                                                   float4* __restrict__ O1, float4* __restrict__ O2, const long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 t = T[i], a = A[i];
        O1[i] = make_float4(t.x + a.x, t.y + a.y, t.z + a.z, t.w + a.w);
        if (B) {
            const float4 b = B[i];
            O2[i] = make_float4(t.x + b.x, t.y + b.y, t.z + b.z, t.w + b.w);
        }
    }
}
// Backward of the same sites: out = base + g1 (+ g2); acc1 += g1; acc2 += g2  (acc* = gradient accumulators of the
// query-position terms shared by all decoder layers; any of g2 / acc1 / acc2 may be NULL).
__global__ __launch_bounds__(256) void grad_merge_kernel(const float4* __restrict__ base, const float4* __restrict__ g1,
                                                         const float4* __restrict__ g2, float4* __restrict__ acc1,
                                                         float4* __restrict__ acc2, float4* __restrict__ out, const long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 o = base[i];
        const float4 a = g1[i];
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        if (acc1) { float4 c = acc1[i]; c.x += a.x; c.y += a.y; c.z += a.z; c.w += a.w; acc1[i] = c; }
        if (g2) {
            const float4 b = g2[i];
            o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
            if (acc2) { float4 c = acc2[i]; c.x += b.x; c.y += b.y; c.z += b.z; c.w += b.w; acc2[i] = c; }
        }
        out[i] = o;
    }
}

// Sine positional embedding (A2/models/transformer.py:474-494): out[r, i] = i even ? sin(x) : cos(x),
// x = (pos[r] * 2 pi) / T^(2 floor(i/2) / nfeat); one launch instead of ~12 tensor kernels.  `pos` is read with stride
// `pstride` (so pos[..., 0] / pos[..., 1] of a [.., 2] tensor need no copy) and `out` rows with stride `ldo` (the 2-D
// embedding is two calls into the two halves of one buffer).  Backward: dpos[r] (+)= sum_i dout[r, i] * d out / d pos.
__global__ __launch_bounds__(256) void sine_embed_kernel(const float* __restrict__ pos, int pstride, float* __restrict__ out, long ldo,
                                                         int rows, int nfeat, float temperature) {
    const long total = (long)rows * nfeat;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int r = (int)(idx / nfeat), i = (int)(idx - (long)r * nfeat);
        const float dim_t = powf(temperature, (float)(2 * (i >> 1)) / (float)nfeat);
        const float x = (pos[(long)r * pstride] * 6.283185307179586f) / dim_t;
        out[(long)r * ldo + i] = (i & 1) ? cosf(x) : sinf(x);
    }
}
__global__ __launch_bounds__(256) void sine_embed_bwd_kernel(const float* __restrict__ pos, int pstride, const float* __restrict__ dout,
                                                             long ldo, float* __restrict__ dpos, int dstride, int rows, int nfeat,
                                                             float temperature, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);               // one wave per row
    if (r >= rows) return;
    const float p2 = pos[(long)r * pstride] * 6.283185307179586f;
    float acc = 0.f;
    for (int i = lane; i < nfeat; i += 64) {
        const float dim_t = powf(temperature, (float)(2 * (i >> 1)) / (float)nfeat);
        const float x = p2 / dim_t;
        const float d = (i & 1) ? -sinf(x) : cosf(x);
        acc = fmaf(dout[(long)r * ldo + i] * d, 6.283185307179586f / dim_t, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        float* dst = dpos + (long)r * dstride;
        *dst = accumulate ? *dst + acc : acc;
    }
}

// Reductions of an NHWC map over one spatial axis (+ optional small addend):
//   blocks [0, N*W):       Or[n,x,:] = scale_r * sum_y X[n,y,x,:] (+ Ar[n,x,:])
//   blocks [N*W, N*W+N*H): Oc[n,y,:] = scale_c * sum_x X[n,y,x,:] (+ Ac[n,y,:])
// forward: X = src, scale = 1/H, 1/W, addend = positional embeddings (k_row / k_col inputs, mean-before-project);
// backward: X = a logit-gradient map, scale = 1 (sum over the broadcast axis).  Xc (second map) may differ from Xr.
__device__ __forceinline__ void hw_reduce_body(const float* __restrict__ Xr, const float* __restrict__ Xc,
                                               const float* __restrict__ Ar, const float* __restrict__ Ac,
                                               float* __restrict__ Or, float* __restrict__ Oc, int N, int H, int W, int C,
                                               float scale_r, float scale_c, const int b) {
    const int c = threadIdx.x;          // C <= 256 handled per pass
    if (C == 256) {
        // thread = (channel quad, 1 of 4 interleaved slices of the reduced axis): 16-byte loads, 4 independent accumulation chains per
        // workgroup column and an unrolled loop keep ~8 loads in flight per thread (the scalar loop below waits for each in turn)
        __shared__ float4 part[4][64];
        const int c4 = c & 63, sl = c >> 6;
        const bool row = b < N * W;
        const int r = row ? b : b - N * W;
        const int cnt = row ? H : W;
        const float* base;
        long step;
        if (row) { const int n = b / W, xw = b - n * W; base = Xr + (((long)n * H) * W + xw) * C + c4 * 4; step = (long)W * C; }
        else { base = Xc + (long)r * W * C + c4 * 4; step = C; }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int k = sl; k < cnt; k += 4) {
            const float4 t = *reinterpret_cast<const float4*>(base + (long)k * step);
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        part[sl][c4] = acc;
        __syncthreads();
        if (sl == 0) {
            const float sc = row ? scale_r : scale_c;
            float4 v = part[0][c4];
            const float4 p1 = part[1][c4], p2 = part[2][c4], p3 = part[3][c4];
            v.x = ((v.x + p1.x) + (p2.x + p3.x)) * sc; v.y = ((v.y + p1.y) + (p2.y + p3.y)) * sc;
            v.z = ((v.z + p1.z) + (p2.z + p3.z)) * sc; v.w = ((v.w + p1.w) + (p2.w + p3.w)) * sc;
            const float* add = row ? Ar : Ac;
            if (add) { const float4 a = *reinterpret_cast<const float4*>(add + (long)r * C + c4 * 4); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
            *reinterpret_cast<float4*>((row ? Or : Oc) + (long)r * C + c4 * 4) = v;
        }
        return;
    }
    if (b < N * W) {
        const int n = b / W, xw = b % W;
        for (int cc = c; cc < C; cc += 256) {
            float s = 0.f;
            for (int y = 0; y < H; ++y) s += Xr[(((long)n * H + y) * W + xw) * C + cc];
            s *= scale_r;
            if (Ar) s += Ar[(long)b * C + cc];
            Or[(long)b * C + cc] = s;
        }
    } else {
        const int r = b - N * W;            // = n*H + y
        const float* base = Xc + (long)r * W * C;
        for (int cc = c; cc < C; cc += 256) {
            float s = 0.f;
            for (int xw = 0; xw < W; ++xw) s += base[(long)xw * C + cc];
            s *= scale_c;
            if (Ac) s += Ac[(long)r * C + cc];
            Oc[(long)r * C + cc] = s;
        }
    }
}
__global__ __launch_bounds__(256) void hw_reduce_kernel(const float* __restrict__ Xr, const float* __restrict__ Xc,
                                                        const float* __restrict__ Ar, const float* __restrict__ Ac,
                                                        float* __restrict__ Or, float* __restrict__ Oc, int N, int H, int W, int C,
                                                        float scale_r, float scale_c) {
    hw_reduce_body(Xr, Xc, Ar, Ac, Or, Oc, N, H, W, C, scale_r, scale_c, blockIdx.x);
}
// The encoder layer's prologue as ONE launch (two independent passes over the same source, A2/models/transformer.py:246-252 + the
// mean-before-project keys): workgroups [0, N (W + H)) form the key means, the rest the two positional adds.
__global__ __launch_bounds__(256) void posadd2_hw_reduce_kernel(const float* __restrict__ X, const float* __restrict__ Prow,
                                                                const float* __restrict__ Pcol, float* __restrict__ Qr,
                                                                float* __restrict__ Qc, float* __restrict__ Kr, float* __restrict__ Kc,
                                                                int N, int H, int W, int C, float scale_r, float scale_c) {
    const int nred = N * (W + H);
    if ((int)blockIdx.x < nred) hw_reduce_body(X, X, Prow, Pcol, Kr, Kc, N, H, W, C, scale_r, scale_c, blockIdx.x);
    else posadd2_body(X, Prow, Pcol, Qr, Qc, N, H, W, C / 4, blockIdx.x - nred, gridDim.x - nred);
}

// out[n,y,x,:] = T[n,y,x,:] + sr * Br[n,x,:] + sc * Bc[n,y,:]   (backward of the two key means: broadcast back)
// (T2 / T3: optional further addends of T's shape -- sibling data gradients that ran as one grouped launch instead of a residual chain)
__global__ __launch_bounds__(256) void bcast_add2_kernel(const float* __restrict__ T, const float* __restrict__ Br,
                                                         const float* __restrict__ Bc, float* __restrict__ out, int N, int H,
                                                         int W, int C4, float sr, float sc, const float* __restrict__ T2 = nullptr,
                                                         const float* __restrict__ T3 = nullptr) {
    const long total = (long)N * H * W * C4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C4);
        long t = idx / C4;
        const int xw = (int)(t % W); t /= W;
        const int yh = (int)(t % H);
        const int n = (int)(t / H);
        float4 v = reinterpret_cast<const float4*>(T)[idx];
        const float4 br = reinterpret_cast<const float4*>(Br)[((long)n * W + xw) * C4 + c];
        const float4 bc = reinterpret_cast<const float4*>(Bc)[((long)n * H + yh) * C4 + c];
        if (T2) { const float4 u = reinterpret_cast<const float4*>(T2)[idx]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        if (T3) { const float4 u = reinterpret_cast<const float4*>(T3)[idx]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        reinterpret_cast<float4*>(out)[idx] = make_float4(v.x + sr * br.x + sc * bc.x, v.y + sr * br.y + sc * bc.y,
                                                          v.z + sr * br.z + sc * bc.z, v.w + sr * br.w + sc * bc.w);
    }
}

}  // namespace

extern "C" int cdetr_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                   int32_t rows, int32_t C, float eps, void* stream) {
    CDETR_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows >= 0, "cdetr_layernorm_fwd: null pointer");
    CDETR_CHECK_ARG(C > 0 && (C & 255) == 0 && C <= 1024, "cdetr_layernorm_fwd: C must be a multiple of 256, <= 1024 (got %d)", C);
    if (rows == 0) return CDETR_OK;
    int blocks = (rows + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, gamma, beta, y, mean,
                       rstd, rows, C, eps, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    return cdetr_launch_status("cdetr_layernorm_fwd");
}

extern "C" int cdetr_layernorm_fwd_add(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                       const float* a1, const float* a2, float* o1, float* o2, int32_t rows, int32_t C, float eps,
                                       void* stream) {
    CDETR_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows >= 0, "cdetr_layernorm_fwd_add: null pointer");
    CDETR_CHECK_ARG(a1 && o1 && (!a2 == !o2), "cdetr_layernorm_fwd_add: a1 / o1 are required, a2 / o2 come as a pair");
    CDETR_CHECK_ARG(C > 0 && (C & 255) == 0 && C <= 1024, "cdetr_layernorm_fwd_add: C must be a multiple of 256, <= 1024 (got %d)", C);
    if (rows == 0) return CDETR_OK;
    int blocks = (rows + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, gamma, beta, y, mean,
                       rstd, rows, C, eps, a1, a2, o1, o2);
    return cdetr_launch_status("cdetr_layernorm_fwd_add");
}

extern "C" int cdetr_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                   int32_t B, int32_t P, int32_t C, int32_t G, float eps, void* stream) {
    CDETR_CHECK_ARG(x && gamma && beta && y && mean && rstd && B > 0 && P > 0 && C > 0 && G > 0, "cdetr_groupnorm_fwd: bad args");
    CDETR_CHECK_ARG(C % G == 0 && ((C / G) & 3) == 0 && C / G <= 64, "cdetr_groupnorm_fwd: channels per group must be a multiple of 4, <= 64 (got %d)", C / G);
    hipLaunchKernelGGL(gn_fwd_kernel, dim3(B * G), dim3(GN_NT), 0, reinterpret_cast<hipStream_t>(stream), x, gamma, beta, y, mean, rstd, P, C, G, eps);
    return cdetr_launch_status("cdetr_groupnorm_fwd");
}

extern "C" int cdetr_groupnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx,
                                   float* dgamma, float* dbeta, int32_t B, int32_t P, int32_t C, int32_t G, void* stream) {
    CDETR_CHECK_ARG(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && B > 0 && P > 0 && C > 0 && G > 0, "cdetr_groupnorm_bwd: bad args");
    CDETR_CHECK_ARG(C % G == 0 && ((C / G) & 3) == 0 && C / G <= 64, "cdetr_groupnorm_bwd: channels per group must be a multiple of 4, <= 64 (got %d)", C / G);
    hipLaunchKernelGGL(gn_bwd_kernel, dim3(B * G), dim3(GN_NT_BWD), 0, reinterpret_cast<hipStream_t>(stream), dy, x, mean, rstd, gamma, dx, dgamma, dbeta, P, C, G);
    return cdetr_launch_status("cdetr_groupnorm_bwd");
}

// The split forms (pixels of an image spread over workgroups, statistics through `ws`): C == 256, G a divisor of 64 with C / G a multiple of 4 and
// a power of two of channel quads per group; any other shape -- or a workspace that is too small -- runs the one-workgroup-per-(image, group) kernels.
static bool gn_split_ok(int B, int P, int C, int G, const void* ws, long ws_bytes, int per) {
    const int cg = C / G, cgq = cg >> 2;
    const int nchunk = (P + GN_CH - 1) / GN_CH;
    return C == 256 && G >= 32 && G <= 64 && (256 % G) == 0 && (cg & 3) == 0 && (cgq & (cgq - 1)) == 0 && P >= 4 * GN_CH && ws != nullptr &&
           ws_bytes >= (long)B * nchunk * G * per * 4 && (reinterpret_cast<uintptr_t>(ws) & 3) == 0;
}

extern "C" int cdetr_groupnorm_fwd_ws(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                      int32_t B, int32_t P, int32_t C, int32_t G, float eps, void* ws, int64_t ws_bytes, void* stream) {
    if (!(x && B > 0 && P > 0 && C > 0 && G > 0 && C % G == 0) || !gn_split_ok(B, P, C, G, ws, ws_bytes, 3))
        return cdetr_groupnorm_fwd(x, gamma, beta, y, mean, rstd, B, P, C, G, eps, stream);
    CDETR_CHECK_ARG(gamma && beta && y && mean && rstd, "cdetr_groupnorm_fwd_ws: null pointer");
    const int nchunk = (P + GN_CH - 1) / GN_CH;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gn_split_stats_kernel, dim3(nchunk, B), dim3(256), 0, st, x, reinterpret_cast<float*>(ws), P, G, nchunk);
    hipLaunchKernelGGL(gn_split_apply_kernel, dim3(nchunk, B), dim3(256), 0, st, x, gamma, beta, reinterpret_cast<const float*>(ws), y, mean, rstd, P, G, nchunk, eps);
    return cdetr_launch_status("cdetr_groupnorm_fwd_ws");
}

extern "C" int cdetr_groupnorm_bwd_ws(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx,
                                      float* dgamma, float* dbeta, int32_t B, int32_t P, int32_t C, int32_t G, void* ws, int64_t ws_bytes, void* stream) {
    if (!(x && B > 0 && P > 0 && C > 0 && G > 0 && C % G == 0) || !gn_split_ok(B, P, C, G, ws, ws_bytes, 2))
        return cdetr_groupnorm_bwd(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, B, P, C, G, stream);
    CDETR_CHECK_ARG(dy && mean && rstd && gamma && dx && dgamma && dbeta, "cdetr_groupnorm_bwd_ws: null pointer");
    const int nchunk = (P + GN_CH - 1) / GN_CH;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gn_split_bwd_stats_kernel, dim3(nchunk, B), dim3(256), 0, st, dy, x, mean, rstd, gamma, reinterpret_cast<float*>(ws), dgamma, dbeta, P, G, nchunk);
    hipLaunchKernelGGL(gn_split_bwd_apply_kernel, dim3(nchunk, B), dim3(256), 0, st, dy, x, mean, rstd, gamma, reinterpret_cast<const float*>(ws), dx, P, G, nchunk);
    return cdetr_launch_status("cdetr_groupnorm_bwd_ws");
}

extern "C" int cdetr_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                   const float* add, float* dx, float* dgamma, float* dbeta, int32_t rows, int32_t C, void* dx16, void* stream) {
    CDETR_CHECK_ARG(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && rows >= 0, "cdetr_layernorm_bwd: null pointer");
    CDETR_CHECK_ARG(!dx16 || C == 256, "cdetr_layernorm_bwd: the bf16 twin of dx exists for C = 256");
    CDETR_CHECK_ARG(C > 0 && (C & 255) == 0 && C <= 1024, "cdetr_layernorm_bwd: C must be a multiple of 256, <= 1024 (got %d)", C);
    if (rows == 0) return CDETR_OK;
    // >= 4 rows per wave amortise the per-workgroup dgamma / dbeta atomics on long inputs; short ones (decoder: 600 rows) are latency
    // bound, one row per wave there
    static const int rpb = getenv("CDETR_LN_BWD_ROWS") ? atoi(getenv("CDETR_LN_BWD_ROWS")) : 32;      // rows per workgroup on long inputs (A/B)
    int blocks = rows <= 2048 ? (rows + 3) / 4 : (rows + rpb - 1) / rpb;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, x, mean, rstd, gamma,
                       add, dx, dgamma, dbeta, rows, C, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 1, 1, reinterpret_cast<__bf16*>(dx16));
    return cdetr_launch_status("cdetr_layernorm_bwd");
}

extern "C" int cdetr_layernorm_bwd_merge(const float* dy, const float* g1, const float* g2, float* acc1, float* acc2, const float* Br,
                                         const float* Bc, float sr, float sc, int32_t H, int32_t W, const float* x,
                                         const float* mean, const float* rstd, const float* gamma, const float* add, float* dx, float* dgamma,
                                         float* dbeta, int32_t rows, int32_t C, void* dx16, void* stream) {
    CDETR_CHECK_ARG(dy && g1 && x && mean && rstd && gamma && dx && dgamma && dbeta && rows >= 0 && (g2 || !acc2), "cdetr_layernorm_bwd_merge: bad args");
    CDETR_CHECK_ARG(!Br == !Bc && (!Br || (H > 0 && W > 0 && rows % (H * W) == 0)), "cdetr_layernorm_bwd_merge: Br / Bc come as a pair, rows = N * H * W");
    CDETR_CHECK_ARG(C == 256, "cdetr_layernorm_bwd_merge: C must be 256 (got %d)", C);
    if (rows == 0) return CDETR_OK;
    static const int rpb = getenv("CDETR_LN_BWD_ROWS") ? atoi(getenv("CDETR_LN_BWD_ROWS")) : 32;
    int blocks = rows <= 2048 ? (rows + 3) / 4 : (rows + rpb - 1) / rpb;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, x, mean, rstd, gamma,
                       add, dx, dgamma, dbeta, rows, C, g1, g2, acc1, acc2, Br, Bc, sr, sc, Br ? H : 1, Br ? W : 1, reinterpret_cast<__bf16*>(dx16));
    return cdetr_launch_status("cdetr_layernorm_bwd_merge");
}

extern "C" int cdetr_posadd2(const float* X, const float* Prow, const float* Pcol, float* Qr, float* Qc, int32_t N, int32_t H,
                             int32_t W, int32_t C, void* stream) {
    CDETR_CHECK_ARG(X && Prow && Pcol && Qr && Qc && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "cdetr_posadd2: bad args");
    const long n4 = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(posadd2_kernel, dim3(grid_for(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), X, Prow, Pcol, Qr,
                       Qc, N, H, W, C / 4);
    return cdetr_launch_status("cdetr_posadd2");
}

extern "C" int cdetr_hw_reduce(const float* Xr, const float* Xc, const float* Ar, const float* Ac, float* Or, float* Oc, int32_t N,
                               int32_t H, int32_t W, int32_t C, float scale_r, float scale_c, void* stream) {
    CDETR_CHECK_ARG(Xr && Xc && Or && Oc && N > 0 && H > 0 && W > 0 && C > 0, "cdetr_hw_reduce: bad args");
    hipLaunchKernelGGL(hw_reduce_kernel, dim3(N * (W + H)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), Xr, Xc, Ar, Ac, Or,
                       Oc, N, H, W, C, scale_r, scale_c);
    return cdetr_launch_status("cdetr_hw_reduce");
}

extern "C" int cdetr_posadd2_hw_reduce(const float* X, const float* Prow, const float* Pcol, float* Qr, float* Qc, float* Kr, float* Kc,
                                       int32_t N, int32_t H, int32_t W, int32_t C, float scale_r, float scale_c, void* stream) {
    CDETR_CHECK_ARG(X && Prow && Pcol && Qr && Qc && Kr && Kc && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "cdetr_posadd2_hw_reduce: bad args");
    const long n4 = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(posadd2_hw_reduce_kernel, dim3(N * (W + H) + grid_for(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), X, Prow,
                       Pcol, Qr, Qc, Kr, Kc, N, H, W, C, scale_r, scale_c);
    return cdetr_launch_status("cdetr_posadd2_hw_reduce");
}

extern "C" int cdetr_bcast_add2(const float* T, const float* Br, const float* Bc, float* out, int32_t N, int32_t H, int32_t W,
                                int32_t C, float sr, float sc, void* stream) {
    CDETR_CHECK_ARG(T && Br && Bc && out && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "cdetr_bcast_add2: bad args");
    const long n4 = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(bcast_add2_kernel, dim3(grid_for(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), T, Br, Bc, out, N,
                       H, W, C / 4, sr, sc);
    return cdetr_launch_status("cdetr_bcast_add2");
}

extern "C" int cdetr_bcast_add2_sum(const float* T, const float* T2, const float* T3, const float* Br, const float* Bc, float* out, int32_t N,
                                    int32_t H, int32_t W, int32_t C, float sr, float sc, void* stream) {
    CDETR_CHECK_ARG(T && Br && Bc && out && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && (T2 || !T3), "cdetr_bcast_add2_sum: bad args");
    const long n4 = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(bcast_add2_kernel, dim3(grid_for(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), T, Br, Bc, out, N,
                       H, W, C / 4, sr, sc, T2, T3);
    return cdetr_launch_status("cdetr_bcast_add2_sum");
}

extern "C" int cdetr_weight_mirror(const cdetr_mirror_item* items_dev, int32_t n_items, int32_t total_tiles, void* stream) {
    CDETR_CHECK_ARG(items_dev && n_items > 0 && total_tiles > 0, "cdetr_weight_mirror: bad args");
    hipLaunchKernelGGL(weight_mirror_kernel, dim3(total_tiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), items_dev, n_items);
    return cdetr_launch_status("cdetr_weight_mirror");
}

extern "C" int cdetr_weight_images(const cdetr_mirror_item* items_dev, int32_t n_items, int32_t total_blocks, void* stream) {
    CDETR_CHECK_ARG(items_dev && n_items > 0 && total_blocks > 0, "cdetr_weight_images: bad args");
    hipLaunchKernelGGL(weight_image_kernel, dim3(total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), items_dev, n_items);
    return cdetr_launch_status("cdetr_weight_images");
}

extern "C" int cdetr_add2(const float* T, const float* A, const float* B, float* O1, float* O2, int64_t n, void* stream) {
    CDETR_CHECK_ARG(T && A && O1 && n > 0 && (n & 3) == 0 && (!B == !O2), "cdetr_add2: bad args");
    hipLaunchKernelGGL(add2_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(T), reinterpret_cast<const float4*>(A), reinterpret_cast<const float4*>(B),
                       reinterpret_cast<float4*>(O1), reinterpret_cast<float4*>(O2), (long)(n >> 2));
    return cdetr_launch_status("cdetr_add2");
}

extern "C" int cdetr_grad_merge(const float* base, const float* g1, const float* g2, float* acc1, float* acc2, float* out, int64_t n,
                                void* stream) {
    CDETR_CHECK_ARG(base && g1 && out && n > 0 && (n & 3) == 0 && (g2 || !acc2), "cdetr_grad_merge: bad args");
    hipLaunchKernelGGL(grad_merge_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(base), reinterpret_cast<const float4*>(g1), reinterpret_cast<const float4*>(g2),
                       reinterpret_cast<float4*>(acc1), reinterpret_cast<float4*>(acc2), reinterpret_cast<float4*>(out), (long)(n >> 2));
    return cdetr_launch_status("cdetr_grad_merge");
}

extern "C" int cdetr_sine_embed(const float* pos, int32_t pstride, float* out, int64_t ldo, int32_t rows, int32_t nfeat, float temperature,
                                void* stream) {
    CDETR_CHECK_ARG(pos && out && rows > 0 && nfeat > 0 && pstride > 0 && ldo >= nfeat, "cdetr_sine_embed: bad args");
    long blocks = ((long)rows * nfeat + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sine_embed_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pos, pstride, out,
                       (long)ldo, rows, nfeat, temperature);
    return cdetr_launch_status("cdetr_sine_embed");
}

extern "C" int cdetr_sine_embed_bwd(const float* pos, int32_t pstride, const float* dout, int64_t ldo, float* dpos, int32_t dstride,
                                    int32_t rows, int32_t nfeat, float temperature, int32_t accumulate, void* stream) {
    CDETR_CHECK_ARG(pos && dout && dpos && rows > 0 && nfeat > 0 && pstride > 0 && dstride > 0 && ldo >= nfeat, "cdetr_sine_embed_bwd: bad args");
    hipLaunchKernelGGL(sine_embed_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pos, pstride,
                       dout, (long)ldo, dpos, dstride, rows, nfeat, temperature, accumulate);
    return cdetr_launch_status("cdetr_sine_embed_bwd");
}
