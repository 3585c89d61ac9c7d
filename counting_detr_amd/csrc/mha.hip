// mha.hip -- fused multi-head self-attention core for the decoder queries (A2/models/transformer.py:337,369-370:
// nn.MultiheadAttention(256, 8) on B x Q = 2 x 300..900 queries, head dim 32, no masks).
//
// The problem is tiny (8 * Q^2 logits per image) and launch/latency bound: the reference path (and round-1's first
// version) spends 5-6 launches forward and ~12 backward on it (scale, bmm, softmax, bmm, permute copies, ...).  Here:
//   mha_fwd_kernel   : o = softmax(scale * q k^T) v, flash-style (online softmax over 64-key LDS tiles), saves the row
//                      log-sum-exp; lane (i, g) owns query i (16 per wave) and every 4th key -> no cross-lane work per key.
//   mha_bwd_q_kernel : dq (lanes own queries) and D_i = <dO_i, O_i>;
//   mha_bwd_kv_kernel: dk, dv (lanes own keys, loop over query tiles) -- no atomics, P recomputed from the saved LSE.
// fp32 VALU throughout (the matrix cores would not be fed by 32-wide heads at this size).
// Layouts: qk [N][L][2E] (q | k), v [N][L][E], o / dO [N][L][E], lse / D [N][nh][L]; E = nh * 32.
#include "../../include/cdetr_hip.h"
#include "common.h"

namespace {

constexpr int D = 32;
constexpr int KT = 64;   // keys (or queries) per LDS tile

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ float dot32(const float (&a)[D], const float* __restrict__ b) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
        const float4 t = *reinterpret_cast<const float4*>(b + c4 * 4);
        s0 = fmaf(a[c4 * 4 + 0], t.x, s0);
        s1 = fmaf(a[c4 * 4 + 1], t.y, s1);
        s2 = fmaf(a[c4 * 4 + 2], t.z, s2);
        s3 = fmaf(a[c4 * 4 + 3], t.w, s3);
    }
    return (s0 + s1) + (s2 + s3);
}

__device__ __forceinline__ void axpy32(float (&acc)[D], float p, const float* __restrict__ b) {
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
        const float4 t = *reinterpret_cast<const float4*>(b + c4 * 4);
        acc[c4 * 4 + 0] = fmaf(p, t.x, acc[c4 * 4 + 0]);
        acc[c4 * 4 + 1] = fmaf(p, t.y, acc[c4 * 4 + 1]);
        acc[c4 * 4 + 2] = fmaf(p, t.z, acc[c4 * 4 + 2]);
        acc[c4 * 4 + 3] = fmaf(p, t.w, acc[c4 * 4 + 3]);
    }
}

// rows [r0, r0+KT) of a [L][ld] matrix (32 columns starting at col0) -> LDS tile [KT][32]; zero beyond L
__device__ __forceinline__ void load_tile(float* __restrict__ dst, const float* __restrict__ src, long ld, int col0, int r0, int L) {
    for (int idx = threadIdx.x; idx < KT * 8; idx += blockDim.x) {
        const int r = idx >> 3, c4 = idx & 7;
        const float4 t = (r0 + r < L) ? ld4(src + (long)(r0 + r) * ld + col0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(dst + r * D + c4 * 4) = t;
    }
}

__global__ __launch_bounds__(256) void mha_fwd_kernel(const float* __restrict__ qk, const float* __restrict__ v,
                                                      float* __restrict__ o, float* __restrict__ lse, int N, int L, int nh,
                                                      float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[KT * D];
    __shared__ __attribute__((aligned(16))) float Vs[KT * D];
    const int E = nh * D;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i32 = lane & 15, g = lane >> 4;      // 16 rows per wave, 4-way split of the reduction axis
    const int n = blockIdx.y / nh, head = blockIdx.y % nh;
    const int q = blockIdx.x * 64 + wid * 16 + i32;
    const bool qv = q < L;
    const float* qkn = qk + (long)n * L * 2 * E;
    const float* vn = v + (long)n * L * E;
    float qr[D];
    {
        const float* qp = qkn + (long)(qv ? q : 0) * 2 * E + head * D;
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            const float4 t = ld4(qp + c4 * 4);
            qr[c4 * 4 + 0] = t.x * scale; qr[c4 * 4 + 1] = t.y * scale; qr[c4 * 4 + 2] = t.z * scale; qr[c4 * 4 + 3] = t.w * scale;
        }
    }
    float m = -INFINITY, l = 0.f, acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.f;
    for (int k0 = 0; k0 < L; k0 += KT) {
        __syncthreads();
        load_tile(Ks, qkn, 2 * E, E + head * D, k0, L);
        load_tile(Vs, vn, E, head * D, k0, L);
        __syncthreads();
        const int nk = min(KT, L - k0);
        float s[KT / 4];
        float tm = -INFINITY;
#pragma unroll
        for (int j = 0; j < KT / 4; ++j) {
            const int kk = 4 * j + g;
            s[j] = (kk < nk) ? dot32(qr, Ks + kk * D) : -INFINITY;
            tm = fmaxf(tm, s[j]);
        }
        const float mn = fmaxf(m, tm);
        if (mn > -INFINITY) {
            const float corr = expf(m - mn);       // m = -inf -> 0
            l *= corr;
#pragma unroll
            for (int c = 0; c < D; ++c) acc[c] *= corr;
#pragma unroll
            for (int j = 0; j < KT / 4; ++j) {
                const int kk = 4 * j + g;
                if (kk < nk) {
                    const float p = expf(s[j] - mn);
                    l += p;
                    axpy32(acc, p, Vs + kk * D);
                }
            }
            m = mn;
        }
    }
    // combine the four key-subset partials of each query
    float mn = fmaxf(m, __shfl_xor(m, 16, 64));
    mn = fmaxf(mn, __shfl_xor(mn, 32, 64));
    const float cs = (m > -INFINITY) ? expf(m - mn) : 0.f;
    l *= cs;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        float a = acc[c] * cs;
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        acc[c] = a * inv;
    }
    if (qv) {
        float* op = o + ((long)n * L + q) * E + head * D + g * 8;      // each lane group writes 8 channels
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4)
            *reinterpret_cast<float4*>(op + c4 * 4) = make_float4(acc[g * 8 + c4 * 4 + 0], acc[g * 8 + c4 * 4 + 1],
                                                                  acc[g * 8 + c4 * 4 + 2], acc[g * 8 + c4 * 4 + 3]);
        if (g == 0) lse[((long)n * nh + head) * L + q] = mn + logf(l);
    }
}

// dq[i] = scale * sum_j p_ij (dO_i.v_j - D_i) k_j,   D_i = dO_i . O_i   (lanes own queries)
__global__ __launch_bounds__(256) void mha_bwd_q_kernel(const float* __restrict__ qk, const float* __restrict__ v,
                                                        const float* __restrict__ o, const float* __restrict__ dO,
                                                        const float* __restrict__ lse, float* __restrict__ dqk,
                                                        float* __restrict__ Dbuf, int N, int L, int nh, float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[KT * D];
    __shared__ __attribute__((aligned(16))) float Vs[KT * D];
    const int E = nh * D;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i32 = lane & 15, g = lane >> 4;      // 16 rows per wave, 4-way split of the reduction axis
    const int n = blockIdx.y / nh, head = blockIdx.y % nh;
    const int q = blockIdx.x * 64 + wid * 16 + i32;
    const bool qv = q < L;
    const float* qkn = qk + (long)n * L * 2 * E;
    const float* vn = v + (long)n * L * E;
    float qr[D], dor[D], dq[D];
    float Di = 0.f;
    {
        const long row = (long)n * L + (qv ? q : 0);
        const float* qp = qkn + (long)(qv ? q : 0) * 2 * E + head * D;
        const float* dp = dO + row * E + head * D;
        const float* op = o + row * E + head * D;
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            const float4 t = ld4(qp + c4 * 4), u = ld4(dp + c4 * 4), w = ld4(op + c4 * 4);
            qr[c4 * 4 + 0] = t.x * scale; qr[c4 * 4 + 1] = t.y * scale; qr[c4 * 4 + 2] = t.z * scale; qr[c4 * 4 + 3] = t.w * scale;
            dor[c4 * 4 + 0] = u.x; dor[c4 * 4 + 1] = u.y; dor[c4 * 4 + 2] = u.z; dor[c4 * 4 + 3] = u.w;
            Di += (u.x * w.x + u.y * w.y) + (u.z * w.z + u.w * w.w);
        }
#pragma unroll
        for (int c = 0; c < D; ++c) dq[c] = 0.f;
    }
    const float li = qv ? lse[((long)n * nh + head) * L + q] : 0.f;
    for (int k0 = 0; k0 < L; k0 += KT) {
        __syncthreads();
        load_tile(Ks, qkn, 2 * E, E + head * D, k0, L);
        load_tile(Vs, vn, E, head * D, k0, L);
        __syncthreads();
        const int nk = min(KT, L - k0);
        for (int kk = g; kk < nk; kk += 4) {
            const float p = expf(dot32(qr, Ks + kk * D) - li);
            const float ds = p * (dot32(dor, Vs + kk * D) - Di);
            axpy32(dq, ds, Ks + kk * D);
        }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) {
        float a = dq[c];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        dq[c] = a * scale;
    }
    if (qv) {
        float* out = dqk + ((long)n * L + q) * 2 * E + head * D;
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4)
            *reinterpret_cast<float4*>(out + g * 8 + c4 * 4) = make_float4(dq[g * 8 + c4 * 4 + 0], dq[g * 8 + c4 * 4 + 1],
                                                                           dq[g * 8 + c4 * 4 + 2], dq[g * 8 + c4 * 4 + 3]);
        if (g == 0) Dbuf[((long)n * nh + head) * L + q] = Di;
    }
}

// dk[j] = scale * sum_i p_ij (dO_i.v_j - D_i) q_i,   dv[j] = sum_i p_ij dO_i   (lanes own keys, query tiles through LDS)
__global__ __launch_bounds__(256) void mha_bwd_kv_kernel(const float* __restrict__ qk, const float* __restrict__ v,
                                                         const float* __restrict__ dO, const float* __restrict__ lse,
                                                         const float* __restrict__ Dbuf, float* __restrict__ dqk,
                                                         float* __restrict__ dv, int N, int L, int nh, float scale) {
    __shared__ __attribute__((aligned(16))) float Qs[KT * D];
    __shared__ __attribute__((aligned(16))) float Os[KT * D];
    __shared__ float Ls[KT], Dsh[KT];
    const int E = nh * D;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i32 = lane & 15, g = lane >> 4;      // 16 rows per wave, 4-way split of the reduction axis
    const int n = blockIdx.y / nh, head = blockIdx.y % nh;
    const int j = blockIdx.x * 64 + wid * 16 + i32;
    const bool jv = j < L;
    const float* qkn = qk + (long)n * L * 2 * E;
    const float* don = dO + (long)n * L * E;
    float kr[D], vr[D], dk[D], dvv[D];
    {
        const float* kp = qkn + (long)(jv ? j : 0) * 2 * E + E + head * D;
        const float* vp = v + ((long)n * L + (jv ? j : 0)) * E + head * D;
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            const float4 t = ld4(kp + c4 * 4), u = ld4(vp + c4 * 4);
            kr[c4 * 4 + 0] = t.x * scale; kr[c4 * 4 + 1] = t.y * scale; kr[c4 * 4 + 2] = t.z * scale; kr[c4 * 4 + 3] = t.w * scale;
            vr[c4 * 4 + 0] = u.x; vr[c4 * 4 + 1] = u.y; vr[c4 * 4 + 2] = u.z; vr[c4 * 4 + 3] = u.w;
        }
#pragma unroll
        for (int c = 0; c < D; ++c) { dk[c] = 0.f; dvv[c] = 0.f; }
    }
    for (int q0 = 0; q0 < L; q0 += KT) {
        __syncthreads();
        load_tile(Qs, qkn, 2 * E, head * D, q0, L);
        load_tile(Os, don, E, head * D, q0, L);
        for (int r = threadIdx.x; r < KT; r += blockDim.x) {
            const bool ok = q0 + r < L;
            Ls[r] = ok ? lse[((long)n * nh + head) * L + q0 + r] : INFINITY;     // p = exp(s - inf) = 0 beyond L
            Dsh[r] = ok ? Dbuf[((long)n * nh + head) * L + q0 + r] : 0.f;
        }
        __syncthreads();
        const int nq = min(KT, L - q0);
        for (int r = g; r < nq; r += 4) {
            const float p = expf(dot32(kr, Qs + r * D) - Ls[r]);
            axpy32(dvv, p, Os + r * D);
            const float ds = p * (dot32(vr, Os + r * D) - Dsh[r]);
            axpy32(dk, ds, Qs + r * D);
        }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) {
        float a = dk[c], b = dvv[c];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        dk[c] = a * scale;
        dvv[c] = b;
    }
    if (jv) {
        float* ok = dqk + ((long)n * L + j) * 2 * E + E + head * D + g * 8;
        float* ov = dv + ((long)n * L + j) * E + head * D + g * 8;
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
            *reinterpret_cast<float4*>(ok + c4 * 4) = make_float4(dk[g * 8 + c4 * 4 + 0], dk[g * 8 + c4 * 4 + 1],
                                                                  dk[g * 8 + c4 * 4 + 2], dk[g * 8 + c4 * 4 + 3]);
            *reinterpret_cast<float4*>(ov + c4 * 4) = make_float4(dvv[g * 8 + c4 * 4 + 0], dvv[g * 8 + c4 * 4 + 1],
                                                                  dvv[g * 8 + c4 * 4 + 2], dvv[g * 8 + c4 * 4 + 3]);
        }
    }
}

}  // namespace

extern "C" int cdetr_mha_fwd(const float* qk, const float* v, float* o, float* lse, int32_t N, int32_t L, int32_t nh, float scale,
                             void* stream) {
    CDETR_CHECK_ARG(qk && v && o && lse && N > 0 && L > 0 && nh > 0, "cdetr_mha_fwd: bad args");
    dim3 grid((L + 63) / 64, N * nh);
    hipLaunchKernelGGL(mha_fwd_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), qk, v, o, lse, N, L, nh, scale);
    return cdetr_launch_status("cdetr_mha_fwd");
}

extern "C" int cdetr_mha_bwd(const float* qk, const float* v, const float* o, const float* d_o, const float* lse, float* d_qk,
                             float* d_v, float* work, int32_t N, int32_t L, int32_t nh, float scale, void* stream) {
    CDETR_CHECK_ARG(qk && v && o && d_o && lse && d_qk && d_v && work && N > 0 && L > 0 && nh > 0, "cdetr_mha_bwd: bad args");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((L + 63) / 64, N * nh);
    hipLaunchKernelGGL(mha_bwd_q_kernel, grid, dim3(256), 0, st, qk, v, o, d_o, lse, d_qk, work, N, L, nh, scale);
    hipLaunchKernelGGL(mha_bwd_kv_kernel, grid, dim3(256), 0, st, qk, v, d_o, lse, work, d_qk, d_v, N, L, nh, scale);
    return cdetr_launch_status("cdetr_mha_bwd");
}
