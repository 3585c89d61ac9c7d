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
#include <stdlib.h>
#include <type_traits>

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
    mn = xhalf_max(mn);
    const float cs = (m > -INFINITY) ? expf(m - mn) : 0.f;
    l *= cs;
    l += __shfl_xor(l, 16, 64);
    l = xhalf_sum(l);
    const float inv = 1.f / l;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        float a = acc[c] * cs;
        a += __shfl_xor(a, 16, 64);
        a = xhalf_sum(a);
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
        a = xhalf_sum(a);
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
        a = xhalf_sum(a);
        b += __shfl_xor(b, 16, 64);
        b = xhalf_sum(b);
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


// ------------------------------------------------------------------------------------------------ matrix-core variants
// split-bf16 (bf16x3) flash attention for the same problem.  Workgroup = 2 waves; a wave OWNS 32 rows (queries for the
// forward / dq kernels, keys for the dk-dv kernel) as the LANES of a transposed accumulator and walks over 32-row tiles of
// the other side:
//   X^T[other, own]  = sum_c Other[other, c] Own[own, c]      A = tile rows (k = c, natural layout), B = own rows (hoisted, split once)
//   Y^T[c, own]     += sum_other W[other, c] Z[other, own]    A = W^T tile (k = other, transposed at staging), B = Z straight
//                                                             from the registers of the first product (slot j of step s = reg 8s + j)
// so softmax statistics, the log-sum-exp and D are per-LANE scalars and no probability ever touches LDS or HBM.  Tiles are
// split into bf16 hi / lo once per workgroup while they are staged ([row][hi 32 | lo 32 | pad 8]); the transposed tiles
// store the reduction index in the order the accumulator registers enumerate it (perm_pos).  Three tiles are in flight in a
// statically indexed register ring (unconditional loads, rows clamped to L-1 and masked arithmetically).
namespace flash {
constexpr int NWF = 2, NTF = 64 * NWF, TS = 72, TILE = 32 * TS;

// first position of the 4-row group starting at row k0 (multiple of 4) inside a transposed tile
__device__ __forceinline__ int perm_pos(int k0) { return (k0 & 16) | ((k0 & 4) << 1) | ((k0 & 8) >> 1); }
// tile-row index that accumulator register r of lane group g enumerates
__device__ __forceinline__ int reg_row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

struct NatRegs { float4 v[2]; };
struct TrRegs { float t[16]; };      // t[4 kk + c]: plain scalars (a float4 array read component-wise ends up in scratch)
// natural tile: rows r0.. of a [L][ld] matrix, 32 columns from col0; thread -> (row = idx >> 3, c4 = idx & 7), idx = tid + 128 s
__device__ __forceinline__ void nat_fetch(NatRegs& r, const float* __restrict__ src, long ld, int col0, int r0, int L, int tid) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int idx = tid + NTF * s;
        r.v[s] = ld4(src + (long)min(r0 + (idx >> 3), L - 1) * ld + col0 + (idx & 7) * 4);
    }
}
__device__ __forceinline__ void nat_stash(const NatRegs& r, __bf16* tile, int tid) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int idx = tid + NTF * s;
        stash_split4(tile + (idx >> 3) * TS + (idx & 7) * 4, 32, r.v[s].x, r.v[s].y, r.v[s].z, r.v[s].w);
    }
}
// transposed tile (64 threads `t`): block (row quad kq = t & 7, column quad cq = t >> 3) = 4 rows x 4 columns
__device__ __forceinline__ void tr_fetch(TrRegs& r, const float* __restrict__ src, long ld, int col0, int r0, int L, int t) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const float4 v = ld4(src + (long)min(r0 + 4 * (t & 7) + kk, L - 1) * ld + col0 + (t >> 3) * 4);
        r.t[4 * kk] = v.x; r.t[4 * kk + 1] = v.y; r.t[4 * kk + 2] = v.z; r.t[4 * kk + 3] = v.w;
    }
}
__device__ __forceinline__ void tr_stash(const TrRegs& r, __bf16* tile, int t) {
    __bf16* dst = tile + (4 * (t >> 3)) * TS + perm_pos(4 * (t & 7));
#pragma unroll
    for (int c = 0; c < 4; ++c) stash_split4(dst + c * TS, 32, r.t[c], r.t[4 + c], r.t[8 + c], r.t[12 + c]);
}
// acc += A(tile rows, this lane's row i32) * B (hi/lo fragments of the two k-steps); TERMS = 1: hi * hi only (the lo halves are not read)
template <int TERMS = 3>
__device__ __forceinline__ f32x16 mma_tile(f32x16 acc, const __bf16* tile, int i32, int g, const bf16x8 (&bh)[2], const bf16x8 (&bl)[2]) {
    const __bf16* a = tile + i32 * TS + 8 * g;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(a + 16 * s);
        if constexpr (TERMS == 1) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[s], acc, 0, 0, 0);
        } else {
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(a + 32 + 16 * s);
            acc = mfma_bf16x3(ah, al, bh[s], bl[s], acc);
        }
    }
    return acc;
}
// hoisted B operand from this lane's own row: columns 16s + 8g .. +7, scaled
__device__ __forceinline__ void own_frag(const float* __restrict__ row, int g, float scale, bf16x8 (&bh)[2], bf16x8 (&bl)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float4 t0 = ld4(row + 16 * s + 8 * g), t1 = ld4(row + 16 * s + 8 * g + 4);
        const float x[8] = {t0.x * scale, t0.y * scale, t0.z * scale, t0.w * scale, t1.x * scale, t1.y * scale, t1.z * scale, t1.w * scale};
        split_bf16x8(x, bh[s], bl[s]);
    }
}
__device__ __forceinline__ void reg_frag(const f32x16& z, bf16x8 (&bh)[2], bf16x8 (&bl)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float x[8] = {z[8 * s], z[8 * s + 1], z[8 * s + 2], z[8 * s + 3], z[8 * s + 4], z[8 * s + 5], z[8 * s + 6], z[8 * s + 7]};
        split_bf16x8(x, bh[s], bl[s]);
    }
}
// transposed accumulator -> 32 contiguous floats of this lane's row (registers r = 4j..4j+3 are columns 8j + 4g ..)
__device__ __forceinline__ void store_own(float* __restrict__ row, int g, const f32x16& a, float mul) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        *reinterpret_cast<float4*>(row + 8 * j + 4 * g) = make_float4(a[4 * j] * mul, a[4 * j + 1] * mul, a[4 * j + 2] * mul, a[4 * j + 3] * mul);
}
__device__ __forceinline__ f32x16 zero16() { return f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }

__global__ __launch_bounds__(NTF) void fwd_kernel(const float* __restrict__ qk, const float* __restrict__ v, float* __restrict__ o,
                                                  float* __restrict__ lse, int N, int L, int nh, float scale) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * 2 * TILE];          // [buf][K | V^T]
    const int E = nh * D;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i32 = lane & 31, g = lane >> 5;
    const int n = blockIdx.y / nh, head = blockIdx.y % nh;
    const int q = blockIdx.x * 32 * NWF + wid * 32 + i32;
    const float* qkn = qk + (long)n * L * 2 * E;
    const float* vn = v + (long)n * L * E;
    bf16x8 qh[2], ql[2];
    own_frag(qkn + (long)min(q, L - 1) * 2 * E + head * D, g, scale, qh, ql);
    float m = -INFINITY, l = 0.f;
    f32x16 OT = zero16();
    constexpr int PD = 3;                                                      // tiles in flight (register ring, statically indexed)
    struct Ring { NatRegs k; TrRegs v; } r0, r1, r2;          // separate variables, not an array: keeps them in registers
    const int ntile = (L + 31) / 32;
    auto fetch = [&](NatRegs& a, TrRegs& b, int t) __attribute__((always_inline)) {
        const int r0 = min(t, ntile - 1) * 32;                                 // surplus prefetches re-read the last tile
        nat_fetch(a, qkn, 2 * E, E + head * D, r0, L, tid);
        tr_fetch(b, vn, E, head * D, r0, L, lane);          // every wave loads (only wave 0 stages): keeps the ring in registers
    };
    auto stash = [&](const NatRegs& a, const TrRegs& b, int buf) __attribute__((always_inline)) {
        nat_stash(a, lds + buf * 2 * TILE, tid);
        if (wid == 0) tr_stash(b, lds + buf * 2 * TILE + TILE, lane);
    };
    fetch(r0.k, r0.v, 0); fetch(r1.k, r1.v, 1); fetch(r2.k, r2.v, 2);
    stash(r0.k, r0.v, 0);
    __syncthreads();
    auto step = [&](Ring& cur, const Ring& nxt, int t) __attribute__((always_inline)) {
        // steps past the last tile run on a re-read of it with every row masked: numerically a no-op, and the ring stays branch-free
        const int buf = t & 1;
        fetch(cur.k, cur.v, t + PD);                                           // `cur` (tile t) was staged one step ago
        if (t < ntile) {           // workgroup-uniform: the (up to PD - 1) surplus steps of the last ring turn only stage and synchronise
            f32x16 S = mma_tile(zero16(), lds + buf * 2 * TILE, i32, g, qh, ql);
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (t * 32 + reg_row(r, g) >= L) S[r] = -INFINITY;
                mx = fmaxf(mx, S[r]);
            }
            mx = xhalf_max(mx);
            const float mn = fmaxf(m, mx);
            const float corr = __expf(m - mn);
            float ls = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { S[r] = __expf(S[r] - mn); ls += S[r]; }
            ls = xhalf_sum(ls);
            l = l * corr + ls;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) OT[r] *= corr;
            bf16x8 ph[2], pl[2];
            reg_frag(S, ph, pl);
            OT = mma_tile(OT, lds + buf * 2 * TILE + TILE, i32, g, ph, pl);
        }
        stash(nxt.k, nxt.v, buf ^ 1);                                          // tile t + 1
        __syncthreads();
    };
    for (int t0 = 0; t0 < ntile; t0 += PD) {
        step(r0, r1, t0);
        step(r1, r2, t0 + 1);
        step(r2, r0, t0 + 2);
    }
    if (q < L) {
        store_own(o + ((long)n * L + q) * E + head * D, g, OT, 1.f / l);
        if (g == 0) lse[((long)n * nh + head) * L + q] = m + logf(l);
    }
}

// The same forward with the KEYS of a query tile cut in two: 4 waves per workgroup -- waves (0, 1) walk the first half of the key tiles, waves (2, 3)
// the second half, for the same 2 x 32 queries -- and merge their (running maximum, sum, O^T) through LDS at the end.  A wave's life is a chain of
// [barrier, two dependent MFMA groups, a softmax update] per key tile with nobody to overlap with (160 waves on 1024 SIMDs at L = 300): half the steps per
// wave.  Each half stages its own tiles (its two waves = the 128 staging threads of fwd_kernel), both halves share the barriers.
__global__ __launch_bounds__(2 * NTF) void fwd_ks_kernel(const float* __restrict__ qk, const float* __restrict__ v, float* __restrict__ o,
                                                         float* __restrict__ lse, int N, int L, int nh, float scale) {
    __shared__ __attribute__((aligned(16))) __bf16 lds_all[2 * 2 * 2 * TILE];      // [key half][buf][K | V^T]
    const int E = nh * D;
    const int tid = threadIdx.x & (NTF - 1), lane = tid & 63, wid = tid >> 6, i32 = lane & 31, g = lane >> 5;
    const int kh = threadIdx.x / NTF;                                          // key half of this wave pair
    __bf16* lds = lds_all + kh * (2 * 2 * TILE);
    const int n = blockIdx.y / nh, head = blockIdx.y % nh;
    const int q = blockIdx.x * 32 * NWF + wid * 32 + i32;
    const float* qkn = qk + (long)n * L * 2 * E;
    const float* vn = v + (long)n * L * E;
    bf16x8 qh[2], ql[2];
    own_frag(qkn + (long)min(q, L - 1) * 2 * E + head * D, g, scale, qh, ql);
    float m = -INFINITY, l = 0.f;
    f32x16 OT = zero16();
    constexpr int PD = 3;
    struct Ring { NatRegs k; TrRegs v; } r0, r1, r2;
    const int ntile = (L + 31) / 32, nth = (ntile + 1) / 2;                     // tiles in all / per half (the second half may hold fewer)
    const int tbase = kh * nth, nmine = max(0, min(nth, ntile - tbase));
    auto fetch = [&](NatRegs& a, TrRegs& b, int t) __attribute__((always_inline)) {
        const int r0_ = min(tbase + min(t, max(nmine - 1, 0)), ntile - 1) * 32;   // surplus prefetches re-read this half's last tile
        nat_fetch(a, qkn, 2 * E, E + head * D, r0_, L, tid);
        tr_fetch(b, vn, E, head * D, r0_, L, lane);
    };
    auto stash = [&](const NatRegs& a, const TrRegs& b, int buf) __attribute__((always_inline)) {
        nat_stash(a, lds + buf * 2 * TILE, tid);
        if (wid == 0) tr_stash(b, lds + buf * 2 * TILE + TILE, lane);
    };
    fetch(r0.k, r0.v, 0); fetch(r1.k, r1.v, 1); fetch(r2.k, r2.v, 2);
    stash(r0.k, r0.v, 0);
    __syncthreads();
    auto step = [&](Ring& cur, const Ring& nxt, int t) __attribute__((always_inline)) {
        const int buf = t & 1;
        fetch(cur.k, cur.v, t + PD);
        if (t < nmine) {           // wave-pair-uniform
            f32x16 S = mma_tile(zero16(), lds + buf * 2 * TILE, i32, g, qh, ql);
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if ((tbase + t) * 32 + reg_row(r, g) >= L) S[r] = -INFINITY;
                mx = fmaxf(mx, S[r]);
            }
            mx = xhalf_max(mx);
            const float mn = fmaxf(m, mx);
            const float corr = __expf(m - mn);
            float ls = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { S[r] = __expf(S[r] - mn); ls += S[r]; }
            ls = xhalf_sum(ls);
            l = l * corr + ls;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) OT[r] *= corr;
            bf16x8 ph[2], pl[2];
            reg_frag(S, ph, pl);
            OT = mma_tile(OT, lds + buf * 2 * TILE + TILE, i32, g, ph, pl);
        }
        stash(nxt.k, nxt.v, buf ^ 1);
        __syncthreads();
    };
    for (int t0 = 0; t0 < nth; t0 += PD) {         // (both halves run nth steps: the barriers are the workgroup's)
        step(r0, r1, t0);
        step(r1, r2, t0 + 1);
        step(r2, r0, t0 + 2);
    }
    // merge: the second half parks (m, l, O^T) per lane, the first half folds them in and stores
    float* mrg = reinterpret_cast<float*>(lds_all) + (wid * 64 + lane) * 18;     // (the tile buffers are free: every wave is behind the loop's last barrier)
    if (kh == 1) {
        mrg[0] = m; mrg[1] = l;
#pragma unroll
        for (int r = 0; r < 16; ++r) mrg[2 + r] = OT[r];
    }
    __syncthreads();
    if (kh == 0 && q < L) {
        const float m2 = mrg[0], l2 = mrg[1];
        const float mn = fmaxf(m, m2);
        const float c1 = (m == -INFINITY) ? 0.f : __expf(m - mn), c2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
        const float lt = l * c1 + l2 * c2;
        f32x16 OM;
#pragma unroll
        for (int r = 0; r < 16; ++r) OM[r] = OT[r] * c1 + mrg[2 + r] * c2;
        store_own(o + ((long)n * L + q) * E + head * D, g, OM, 1.f / lt);
        if (g == 0) lse[((long)n * nh + head) * L + q] = mn + logf(lt);
    }
}

// (The same split of the walked side was built for the backward -- the halves' accumulators simply add -- and measured: 21 -> 20 us.  The backward holds
// 320 registers per lane (one wave per SIMD): 320 workgroups of four waves are 1280 waves for 1024 slots, a second round eats the halved loop.  Not kept.)
// The backward is ONE launch: workgroups with blockIdx.z == 0 own queries (dq), those with blockIdx.z == 1 own keys (dk, dv).  The two
// halves are independent -- the key half forms D = rowsum(dO * O) of each query tile itself while it stages the tile (it used to read the
// query half's result, which serialised two 17 us launches of 80 workgroups each) -- so they share the chip instead of following each other.
// TB = bf16 MFMAs per product of the four GRADIENT contractions (dP = dO V^T, dQ, dK, dV): 3 = split-bf16 like the forward, 1 = plain bf16 (the
// backward's default arithmetic, ops.PRECISION_BWD 3).  The recomputed scores S = Q K^T keep all three terms in both: p = exp(S - lse) against the
// FORWARD's log-sum-exp turns an error of S into a relative error of every probability of the row.
template <int TB>
__device__ __forceinline__ void bwd_q_body(__bf16* lds, const float* __restrict__ qk, const float* __restrict__ v, const float* __restrict__ o,
                                           const float* __restrict__ dO, const float* __restrict__ lse, float* __restrict__ dqk,
                                           int N, int L, int nh, float scale) {
    // lds: [buf][K | V | K^T]
    const int E = nh * D;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i32 = lane & 31, g = lane >> 5;
    const int n = blockIdx.y / nh, head = blockIdx.y % nh;
    const int q = blockIdx.x * 32 * NWF + wid * 32 + i32;
    const int qc = min(q, L - 1);
    const float* qkn = qk + (long)n * L * 2 * E;
    const float* vn = v + (long)n * L * E;
    bf16x8 qh[2], ql[2], dh[2], dl[2];
    own_frag(qkn + (long)qc * 2 * E + head * D, g, scale, qh, ql);
    const float* dop = dO + ((long)n * L + qc) * E + head * D;
    own_frag(dop, g, 1.f, dh, dl);
    float Di = 0.f;
    {
        const float* op = o + ((long)n * L + qc) * E + head * D;
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            const float4 u = ld4(dop + c4 * 4), w = ld4(op + c4 * 4);
            Di += (u.x * w.x + u.y * w.y) + (u.z * w.z + u.w * w.w);
        }
    }
    const float li = lse[((long)n * nh + head) * L + qc];
    f32x16 dQT = zero16();
    constexpr int PD = 3;
    struct Ring { NatRegs k, v; TrRegs kt; } r0, r1, r2;
    const int ntile = (L + 31) / 32;
    auto fetch = [&](NatRegs& a, NatRegs& b, TrRegs& c, int t) __attribute__((always_inline)) {
        const int r0 = min(t, ntile - 1) * 32;
        nat_fetch(a, qkn, 2 * E, E + head * D, r0, L, tid);
        nat_fetch(b, vn, E, head * D, r0, L, tid);
        tr_fetch(c, qkn, 2 * E, E + head * D, r0, L, lane);
    };
    auto stash = [&](const NatRegs& a, const NatRegs& b, const TrRegs& c, int buf) __attribute__((always_inline)) {
        __bf16* nb = lds + buf * 3 * TILE;
        nat_stash(a, nb, tid);
        nat_stash(b, nb + TILE, tid);
        if (wid == 0) tr_stash(c, nb + 2 * TILE, lane);
    };
    fetch(r0.k, r0.v, r0.kt, 0); fetch(r1.k, r1.v, r1.kt, 1); fetch(r2.k, r2.v, r2.kt, 2);
    stash(r0.k, r0.v, r0.kt, 0);
    __syncthreads();
    auto step = [&](Ring& cur, const Ring& nxt, int t) __attribute__((always_inline)) {
        const int buf = t & 1;
        fetch(cur.k, cur.v, cur.kt, t + PD);
        const __bf16* base = lds + buf * 3 * TILE;
        if (t < ntile) {           // workgroup-uniform: surplus steps of the last ring turn only stage and synchronise
            f32x16 S = mma_tile(zero16(), base, i32, g, qh, ql);
            const f32x16 dP = mma_tile<TB>(zero16(), base + TILE, i32, g, dh, dl);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = (t * 32 + reg_row(r, g) < L) ? __expf(S[r] - li) : 0.f;
                S[r] = p * (dP[r] - Di);
            }
            bf16x8 sh[2], sl[2];
            reg_frag(S, sh, sl);
            dQT = mma_tile<TB>(dQT, base + 2 * TILE, i32, g, sh, sl);
        }
        stash(nxt.k, nxt.v, nxt.kt, buf ^ 1);
        __syncthreads();
    };
    for (int t0 = 0; t0 < ntile; t0 += PD) {
        step(r0, r1, t0);
        step(r1, r2, t0 + 1);
        step(r2, r0, t0 + 2);
    }
    if (q < L) {
        store_own(dqk + ((long)n * L + q) * 2 * E + head * D, g, dQT, scale);
    }
}

template <int TB>
__device__ __forceinline__ void bwd_kv_body(__bf16* lds, float (*stat)[2][32], const float* __restrict__ qk, const float* __restrict__ v,
                                            const float* __restrict__ o, const float* __restrict__ dO, const float* __restrict__ lse,
                                            float* __restrict__ dqk, float* __restrict__ dv, int N, int L, int nh, float scale) {
    // lds: [buf][Q | dO | Q^T | dO^T]; stat: [buf][lse | D][query of the tile]
    const int E = nh * D;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i32 = lane & 31, g = lane >> 5;
    const int n = blockIdx.y / nh, head = blockIdx.y % nh;
    const int j = blockIdx.x * 32 * NWF + wid * 32 + i32;
    const int jc = min(j, L - 1);
    const float* qkn = qk + (long)n * L * 2 * E;
    const float* don = dO + (long)n * L * E;
    const float* on = o + (long)n * L * E;
    const float* lsn = lse + ((long)n * nh + head) * L;
    bf16x8 kh[2], kl[2], vh[2], vl[2];
    own_frag(qkn + (long)jc * 2 * E + E + head * D, g, scale, kh, kl);
    own_frag(v + ((long)n * L + jc) * E + head * D, g, 1.f, vh, vl);
    f32x16 dKT = zero16(), dVT = zero16();
    constexpr int PD = 3;
    const float* trsrc = (wid == 0) ? qkn : don;                               // wave 0 stages Q^T, wave 1 dO^T
    const long trld = (wid == 0) ? 2 * E : E;
    struct Ring { NatRegs q, d, o; TrRegs t; float s; } r0, r1, r2;
    const int ntile = (L + 31) / 32;
    auto fetch = [&](Ring& r, int t) __attribute__((always_inline)) {
        const int r0 = min(t, ntile - 1) * 32;
        nat_fetch(r.q, qkn, 2 * E, head * D, r0, L, tid);
        nat_fetch(r.d, don, E, head * D, r0, L, tid);
        nat_fetch(r.o, on, E, head * D, r0, L, tid);                        // the forward's output rows of the tile: D = rowsum(dO * O)
        tr_fetch(r.t, trsrc, trld, head * D, r0, L, lane);
        const int qq = t * 32 + (tid & 31);                                // lse of the tile (lanes 0-31 of wave 0)
        const float x = lsn[min(qq, L - 1)];                               // (true, unclamped index: surplus tiles must be inert)
        r.s = (qq < L) ? x : INFINITY;                                     // p = exp(s - inf) = 0 beyond L
    };
    auto stash = [&](const Ring& r, int buf) __attribute__((always_inline)) {
        __bf16* nb = lds + buf * 4 * TILE;
        nat_stash(r.q, nb, tid);
        nat_stash(r.d, nb + TILE, tid);
        tr_stash(r.t, nb + (wid == 0 ? 2 : 3) * TILE, lane);
        if (tid < 32) stat[buf][0][tid] = r.s;
        // D of the tile's 32 queries: thread (row = idx >> 3, 4 columns) holds a quarter-row product of dO and O; the 8 threads of a row
        // are consecutive lanes (rows past L are clamped copies: their p is 0, any finite D will do)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float pd = (r.d.v[s2].x * r.o.v[s2].x + r.d.v[s2].y * r.o.v[s2].y) + (r.d.v[s2].z * r.o.v[s2].z + r.d.v[s2].w * r.o.v[s2].w);
            pd += __shfl_xor(pd, 1, 64);
            pd += __shfl_xor(pd, 2, 64);
            pd += __shfl_xor(pd, 4, 64);
            if ((tid & 7) == 0) stat[buf][1][(tid + NTF * s2) >> 3] = pd;
        }
    };
    fetch(r0, 0); fetch(r1, 1); fetch(r2, 2);
    stash(r0, 0);
    __syncthreads();
    auto step = [&](Ring& cur, const Ring& nxt, int t) __attribute__((always_inline)) {
        const int buf = t & 1;
        fetch(cur, t + PD);
        const __bf16* base = lds + buf * 4 * TILE;
        if (t < ntile) {           // workgroup-uniform: surplus steps of the last ring turn only stage and synchronise
        f32x16 S = mma_tile(zero16(), base, i32, g, kh, kl);                   // S^T[q, key]
        const f32x16 dP = mma_tile<TB>(zero16(), base + TILE, i32, g, vh, vl);     // dP^T[q, key]
        f32x16 P;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float4 ls4 = *reinterpret_cast<const float4*>(&stat[buf][0][8 * jj + 4 * g]);
            const float4 d4 = *reinterpret_cast<const float4*>(&stat[buf][1][8 * jj + 4 * g]);
            const float lsv[4] = {ls4.x, ls4.y, ls4.z, ls4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float p = __expf(S[4 * jj + b] - lsv[b]);
                P[4 * jj + b] = p;
                S[4 * jj + b] = p * (dP[4 * jj + b] - dvv[b]);
            }
        }
        bf16x8 ph[2], pl[2], sh[2], sl[2];
        reg_frag(P, ph, pl);
        dVT = mma_tile<TB>(dVT, base + 3 * TILE, i32, g, ph, pl);              // dV^T[c, key] += dO^T[c, q] P[q, key]
        reg_frag(S, sh, sl);
        dKT = mma_tile<TB>(dKT, base + 2 * TILE, i32, g, sh, sl);              // dK^T[c, key] += Q^T[c, q] dS[q, key]
        }
        stash(nxt, buf ^ 1);
        __syncthreads();
    };
    for (int t0 = 0; t0 < ntile; t0 += PD) {
        step(r0, r1, t0);
        step(r1, r2, t0 + 1);
        step(r2, r0, t0 + 2);
    }
    if (j < L) {
        store_own(dqk + ((long)n * L + j) * 2 * E + E + head * D, g, dKT, scale);
        store_own(dv + ((long)n * L + j) * E + head * D, g, dVT, 1.f);
    }
}

template <int TB>
__global__ __launch_bounds__(NTF) void bwd_kernel(const float* __restrict__ qk, const float* __restrict__ v, const float* __restrict__ o,
                                                  const float* __restrict__ dO, const float* __restrict__ lse, float* __restrict__ dqk,
                                                  float* __restrict__ dv, int N, int L, int nh, float scale) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * 4 * TILE];
    __shared__ __attribute__((aligned(16))) float stat[2][2][32];
    if (blockIdx.z == 0) bwd_q_body<TB>(lds, qk, v, o, dO, lse, dqk, N, L, nh, scale);
    else bwd_kv_body<TB>(lds, stat, qk, v, o, dO, lse, dqk, dv, N, L, nh, scale);
}
}  // namespace flash

}  // namespace

extern "C" int cdetr_mha_fwd(const float* qk, const float* v, float* o, float* lse, int32_t N, int32_t L, int32_t nh, float scale,
                             int32_t precision, void* stream) {
    CDETR_CHECK_ARG(qk && v && o && lse && N > 0 && L > 0 && nh > 0, "cdetr_mha_fwd: bad args");
    dim3 grid((L + 63) / 64, N * nh);
    static const int use_mfma = getenv("CDETR_MHA_MFMA") ? atoi(getenv("CDETR_MHA_MFMA")) : 1;
    static const int key_split = getenv("CDETR_MHA_KEY_SPLIT") ? atoi(getenv("CDETR_MHA_KEY_SPLIT")) : 1;      // A/B: 0 = two waves walk all key tiles
    if (use_mfma && precision == 1 && key_split && L >= 128)
        hipLaunchKernelGGL(flash::fwd_ks_kernel, grid, dim3(2 * flash::NTF), 0, reinterpret_cast<hipStream_t>(stream), qk, v, o, lse, N, L, nh, scale);
    else if (use_mfma && precision == 1) hipLaunchKernelGGL(flash::fwd_kernel, grid, dim3(flash::NTF), 0, reinterpret_cast<hipStream_t>(stream), qk, v, o, lse, N, L, nh, scale);
    else hipLaunchKernelGGL(mha_fwd_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), qk, v, o, lse, N, L, nh, scale);
    return cdetr_launch_status("cdetr_mha_fwd");
}

extern "C" int cdetr_mha_bwd(const float* qk, const float* v, const float* o, const float* d_o, const float* lse, float* d_qk,
                             float* d_v, float* work, int32_t N, int32_t L, int32_t nh, float scale, int32_t precision, void* stream) {
    CDETR_CHECK_ARG(qk && v && o && d_o && lse && d_qk && d_v && work && N > 0 && L > 0 && nh > 0, "cdetr_mha_bwd: bad args");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((L + 63) / 64, N * nh);
    static const int use_mfma = getenv("CDETR_MHA_MFMA") ? atoi(getenv("CDETR_MHA_MFMA")) : 1;
    if (use_mfma && precision == 3) {      // split-bf16 scores, plain-bf16 gradient contractions (flash::bwd_q_body)
        hipLaunchKernelGGL(flash::bwd_kernel<1>, dim3(grid.x, grid.y, 2), dim3(flash::NTF), 0, st, qk, v, o, d_o, lse, d_qk, d_v, N, L, nh, scale);
        return cdetr_launch_status("cdetr_mha_bwd");
    }
    if (use_mfma && precision == 1) {
        hipLaunchKernelGGL(flash::bwd_kernel<3>, dim3(grid.x, grid.y, 2), dim3(flash::NTF), 0, st, qk, v, o, d_o, lse, d_qk, d_v, N, L, nh, scale);
        return cdetr_launch_status("cdetr_mha_bwd");
    }
    hipLaunchKernelGGL(mha_bwd_q_kernel, grid, dim3(256), 0, st, qk, v, o, d_o, lse, d_qk, work, N, L, nh, scale);
    hipLaunchKernelGGL(mha_bwd_kv_kernel, grid, dim3(256), 0, st, qk, v, d_o, lse, work, d_qk, d_v, N, L, nh, scale);
    return cdetr_launch_status("cdetr_mha_bwd");
}
