// igemm.hip -- implicit-GEMM convolution / linear kernels on the fp32 matrix cores of gfx950.
//
// One kernel family serves every dense contraction of the Counting-DETR step:
//   cdetr_gemm  : C = epi( sum_tap A[row(m,tap)] . W_tap )  -- conv forward (NHWC gather, no im2col buffer),
//                 conv data-gradient (transposed gather), linear forward / data-gradient, small batched GEMMs;
//   cdetr_wgrad : dW += dY^T . X[row(p,tap)]                  -- weight gradients, split-K with fp32 atomics.
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate) -- 64 FLOP/clk/SIMD, 157 TFLOP/s peak.
// Tiling: 256 threads = 4 waves (2x2), wave tile (BM/2)x(BN/2) of 32x32 fragments, BK = 16, double-buffered LDS with
// register prefetch (one barrier per k-tile).  LDS rows are padded to 20 floats: the 16-lane groups of ds_read_b128
// then hit 16 distinct 16-byte slots (conflict-free, see MI355X_MICROARCH.md LDS table).
// Within a 16-wide k-tile the reduction order is permuted (lane group g supplies k = 8h + 4g + s at step s) so that one
// ds_read_b128 feeds four MFMAs; A and B use the same permutation, so the sum is unchanged up to fp32 reassociation.
#include "../../include/cdetr_hip.h"
#include "common.h"
#include "rows.h"
#include <stdlib.h>
#include <type_traits>
#include <vector>
#include <algorithm>

namespace {

constexpr int BK = 16;
constexpr int LDS_K = BK + 4;  // padded k-contiguous row

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));      // a native vector (stays in registers through a select, unlike the uint4 struct)
__device__ __forceinline__ u32x4 ld16(const __bf16* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ------------------------------------------------------------------------------------------------ forward / dgrad
template <int BM, int BN, int BL>
__global__ __launch_bounds__(256) void igemm_kernel(const cdetr_gemm_desc d, const int tilesM, const int vecA,
                                                    const int vecB) {
    constexpr int FM = BM / 64, FN = BN / 64;
    constexpr int A_SLOTS = BM / 64, B_SLOTS = BN / 64;
    constexpr int LDS_N = BN + 4;
    constexpr int B_TILE = (BL == 0) ? BN * LDS_K : BK * LDS_N;
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDS_K + 2 * B_TILE];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDS_K;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int i32 = lane & 31, g = lane >> 5;
    const int tm = blockIdx.x % tilesM, tn = blockIdx.x / tilesM;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z;
    const float* __restrict__ A = d.A + batch_off(z, d.batch_inner, d.sA, d.sA2);
    const float* __restrict__ B = d.B + batch_off(z, d.batch_inner, d.sB, d.sB2);
    float* __restrict__ C = d.C + batch_off(z, d.batch_inner, d.sC, d.sC2);
    __bf16* __restrict__ C16 = d.C16 ? reinterpret_cast<__bf16*>(d.C16) + batch_off(z, d.batch_inner, d.sC, d.sC2) : nullptr;   // bf16 twin of C
    __bf16* __restrict__ C16lo = (d.C16 && d.C16lo) ? reinterpret_cast<__bf16*>(d.C16lo) + batch_off(z, d.batch_inner, d.sC, d.sC2) : nullptr;   // its lo plane
    const int K = d.K, taps = d.taps, Ktot = d.K * d.taps;
    const int nkt = (Ktot + BK - 1) / BK;

    // ---- per-thread loader state
    const int kq = tid & 3;
    RowCoord arow[A_SLOTS];
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) arow[i] = decode_row(d.g, m0 + (tid >> 2) + 64 * i, d.M);
    float bscale0[B_SLOTS];
    if (BL == 0) {
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
            const int n = n0 + (tid >> 2) + 64 * i;
            bscale0[i] = (d.w_scale && n < d.N) ? d.w_scale[n] : 1.f;
        }
    }

    float4 ra[A_SLOTS], rb[B_SLOTS];

    auto fetch = [&](int kt) {
        // A: rows gathered, k contiguous
        const int kk = kt * BK + kq * 4;
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            float4 v = zero4();
            if (vecA) {
                if (kk < Ktot) {
                    const int tap = (taps == 1) ? 0 : kk / K;
                    const int c = kk - tap * K;
                    const long row = gather_row(d.g, arow[i], tap);
                    if (row >= 0) v = ld4(A + row * d.lda + c);
                }
            } else if (arow[i].valid) {  // scalar path: dense rows only (checked on the host)
                const float* p = A + (long)arow[i].m * d.lda;
                if (kk + 0 < Ktot) v.x = p[kk + 0];
                if (kk + 1 < Ktot) v.y = p[kk + 1];
                if (kk + 2 < Ktot) v.z = p[kk + 2];
                if (kk + 3 < Ktot) v.w = p[kk + 3];
            }
            ra[i] = v;
        }
        if (BL == 0) {
#pragma unroll
            for (int i = 0; i < B_SLOTS; ++i) {
                const int n = n0 + (tid >> 2) + 64 * i;
                float4 v = zero4();
                if (n < d.N) {
                    const float* p = B + (long)n * d.ldb;
                    if (vecB) {
                        if (kk < Ktot) v = ld4(p + kk);
                    } else {
                        if (kk + 0 < Ktot) v.x = p[kk + 0];
                        if (kk + 1 < Ktot) v.y = p[kk + 1];
                        if (kk + 2 < Ktot) v.z = p[kk + 2];
                        if (kk + 3 < Ktot) v.w = p[kk + 3];
                    }
                    const float s = bscale0[i];
                    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
                }
                rb[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_SLOTS; ++i) {
                const int idx = tid + 256 * i;
                const int kr = idx / (BN / 4);
                const int nq = idx - kr * (BN / 4);
                const int kb = kt * BK + kr;
                const int n = n0 + nq * 4;
                float4 v = zero4();
                if (kb < Ktot && n < d.N) {
                    const int tap = (taps == 1) ? 0 : kb / K;
                    const int k = kb - tap * K;
                    const float* p = B + ((long)k * taps + tap) * d.ldb + n;
                    if (vecB) {
                        v = ld4(p);
                    } else {
                        v.x = p[0];
                        if (n + 1 < d.N) v.y = p[1];
                        if (n + 2 < d.N) v.z = p[2];
                        if (n + 3 < d.N) v.w = p[3];
                    }
                    if (d.w_scale) {
                        const float s = d.w_scale[k];
                        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
                    }
                }
                rb[i] = v;
            }
        }
    };

    auto stash = [&](int buf) {
        float* as = As + buf * BM * LDS_K;
        float* bs = Bs + buf * B_TILE;
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i)
            *reinterpret_cast<float4*>(as + ((tid >> 2) + 64 * i) * LDS_K + kq * 4) = ra[i];
        if (BL == 0) {
#pragma unroll
            for (int i = 0; i < B_SLOTS; ++i)
                *reinterpret_cast<float4*>(bs + ((tid >> 2) + 64 * i) * LDS_K + kq * 4) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < B_SLOTS; ++i) {
                const int idx = tid + 256 * i;
                const int kr = idx / (BN / 4);
                const int nq = idx - kr * (BN / 4);
                *reinterpret_cast<float4*>(bs + kr * LDS_N + nq * 4) = rb[i];
            }
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    fetch(0);
    stash(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) fetch(kt + 1);
        const float* as = As + buf * BM * LDS_K + (wm * (BM / 2) + i32) * LDS_K + g * 4;
        const float* bs = (BL == 0) ? Bs + buf * B_TILE + (wn * (BN / 2) + i32) * LDS_K + g * 4
                                    : Bs + buf * B_TILE + (g * 4) * LDS_N + wn * (BN / 2) + i32;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float a4[FM][4], b4[FN][4];
#pragma unroll
            for (int a = 0; a < FM; ++a) {
                const float4 t = *reinterpret_cast<const float4*>(as + a * 32 * LDS_K + h * 8);
                a4[a][0] = t.x; a4[a][1] = t.y; a4[a][2] = t.z; a4[a][3] = t.w;
            }
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                if (BL == 0) {
                    const float4 t = *reinterpret_cast<const float4*>(bs + b * 32 * LDS_K + h * 8);
                    b4[b][0] = t.x; b4[b][1] = t.y; b4[b][2] = t.z; b4[b][3] = t.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) b4[b][s] = bs[(h * 8 + s) * LDS_N + b * 32];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int a = 0; a < FM; ++a)
#pragma unroll
                    for (int b = 0; b < FN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[a][s], b4[b][s], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < nkt) stash(buf ^ 1);
        __syncthreads();
    }

    mfma_drain(acc);
    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int a = 0; a < FM; ++a) {
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            const int n = n0 + wn * (BN / 2) + b * 32 + i32;
            const bool nvalid = n < d.N;
            const int nc = nvalid ? n : d.N - 1;
            const float bias = d.bias ? d.bias[nc] : 0.f;
            const int mbase = m0 + wm * (BM / 2) + a * 32 + 4 * g;
            float rv[16], gv[16];          // residual / gate operands as one batch of unconditional loads (see igemm_fast_body)
#pragma unroll
            for (int r = 0; r < 16; ++r) { rv[r] = 0.f; gv[r] = 1.f; }
            if (d.resid) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = d.resid[(long)min(mbase + (r & 3) + 8 * (r >> 2), d.M - 1) * d.ldr + nc];
            }
            if (d.gate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) gv[r] = d.gate[(long)min(mbase + (r & 3) + 8 * (r >> 2), d.M - 1) * d.ldg + nc];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                float v = (acc[a][b][r] + bias) * d.out_scale + rv[r];
                v = gv[r] > 0.f ? v : 0.f;
                if (d.relu) v = fmaxf(v, 0.f);
                if (nvalid && m < d.M) { C[(long)m * d.ldc + n] = v; if (C16) C16[(long)m * d.ldc + n] = (__bf16)v; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ wgrad
// dW[i][tap][c] += scale[i] * sum_p dY[p][i] * X[row(p,tap)][c].  Block = (i-tile, (tap, c-tile), k-slice).
template <int BI, int BJ>
__global__ __launch_bounds__(256) void wgrad_kernel(const cdetr_wgrad_desc d, const int tilesI, const int tilesJ,
                                                    const int kt_per_slice) {
    constexpr int FM = BI / 64, FN = BJ / 64;
    constexpr int A_SLOTS = BI / 64, B_SLOTS = BJ / 64;
    constexpr int LDI = BI + 4, LDJ = BJ + 4;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * LDI + 2 * BK * LDJ];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDI;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int i32 = lane & 31, g = lane >> 5;
    const int ti = blockIdx.x % tilesI;
    const int tj = blockIdx.x / tilesI;          // over taps * tilesJ
    const int tap = tj / tilesJ;
    const int c0 = (tj - tap * tilesJ) * BJ;
    const int i0 = ti * BI;
    const int z = blockIdx.z;
    const float* __restrict__ dY = d.dY + batch_off(z, d.batch_inner, d.sY, d.sY2);
    const float* __restrict__ X = d.X + batch_off(z, d.batch_inner, d.sX, d.sX2);
    float* __restrict__ dW = d.dW + batch_off(z, d.batch_inner, d.sW, d.sW2);
    const int nkt_all = (d.P + BK - 1) / BK;
    const int kt_begin = blockIdx.y * kt_per_slice;
    const int kt_end = min(nkt_all, kt_begin + kt_per_slice);
    if (kt_begin >= kt_end) return;

    float4 ra[A_SLOTS], rb[B_SLOTS];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) {
            const int idx = tid + 256 * s;
            const int kr = idx / (BI / 4);
            const int iq = idx - kr * (BI / 4);
            const int p = kt * BK + kr;
            const int i = i0 + iq * 4;
            float4 v = zero4();
            if (p < d.P && i < d.Nout) {
                const float* q = dY + (long)p * d.ldy + i;
                if (i + 3 < d.Nout && (d.ldy & 3) == 0) {
                    v = ld4(q);
                } else {
                    v.x = q[0];
                    if (i + 1 < d.Nout) v.y = q[1];
                    if (i + 2 < d.Nout) v.z = q[2];
                    if (i + 3 < d.Nout) v.w = q[3];
                }
            }
            ra[s] = v;
        }
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) {
            const int idx = tid + 256 * s;
            const int kr = idx / (BJ / 4);
            const int jq = idx - kr * (BJ / 4);
            const int p = kt * BK + kr;
            const int c = c0 + jq * 4;
            float4 v = zero4();
            if (p < d.P && c < d.Cin) {
                const RowCoord rc = decode_row(d.g, p, d.P);
                const long row = gather_row(d.g, rc, tap);
                if (row >= 0) {
                    const float* q = X + row * d.ldx + c;
                    if (c + 3 < d.Cin && (d.ldx & 3) == 0) {
                        v = ld4(q);
                    } else {
                        v.x = q[0];
                        if (c + 1 < d.Cin) v.y = q[1];
                        if (c + 2 < d.Cin) v.z = q[2];
                        if (c + 3 < d.Cin) v.w = q[3];
                    }
                }
            }
            rb[s] = v;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) {
            const int idx = tid + 256 * s;
            const int kr = idx / (BI / 4);
            const int iq = idx - kr * (BI / 4);
            *reinterpret_cast<float4*>(As + buf * BK * LDI + kr * LDI + iq * 4) = ra[s];
        }
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) {
            const int idx = tid + 256 * s;
            const int kr = idx / (BJ / 4);
            const int jq = idx - kr * (BJ / 4);
            *reinterpret_cast<float4*>(Bs + buf * BK * LDJ + kr * LDJ + jq * 4) = rb[s];
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    fetch(kt_begin);
    stash(0);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int buf = (kt - kt_begin) & 1;
        if (kt + 1 < kt_end) fetch(kt + 1);
        const float* as = As + buf * BK * LDI + (g * 4) * LDI + wm * (BI / 2) + i32;
        const float* bs = Bs + buf * BK * LDJ + (g * 4) * LDJ + wn * (BJ / 2) + i32;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float av[FM], bv[FN];
#pragma unroll
                for (int a = 0; a < FM; ++a) av[a] = as[(h * 8 + s) * LDI + a * 32];
#pragma unroll
                for (int b = 0; b < FN; ++b) bv[b] = bs[(h * 8 + s) * LDJ + b * 32];
#pragma unroll
                for (int a = 0; a < FM; ++a)
#pragma unroll
                    for (int b = 0; b < FN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
            }
        }
        if (kt + 1 < kt_end) stash(buf ^ 1);
        __syncthreads();
    }

    mfma_drain(acc);
    const bool single = (gridDim.y == 1);
#pragma unroll
    for (int a = 0; a < FM; ++a) {
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            const int c = c0 + wn * (BJ / 2) + b * 32 + i32;
            if (c >= d.Cin) continue;
            float ws[16];                  // per-row weight scales as one batch of unconditional loads (clamped row)
#pragma unroll
            for (int r = 0; r < 16; ++r) ws[r] = 1.f;
            if (d.w_scale) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ws[r] = d.w_scale[min(i0 + wm * (BI / 2) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, d.Nout - 1)];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wm * (BI / 2) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (i >= d.Nout) continue;
                float v = acc[a][b][r] * ws[r];
                float* dst = dW + (long)i * d.ldw + (long)tap * d.Cin + c;
                (void)single;
                atomicAdd(dst, v);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------ fast forward / dgrad
// Same math as igemm_kernel, for the common case K % 32 == 0 with 16-byte aligned operands (every heavy layer):
//   * a k-tile never straddles a filter tap: the tap is wave-uniform, its gather is resolved once per tap into
//     per-thread row pointers and the steady-state loop has no integer division at all;
//   * workgroup = WM x WN waves, each wave FM x FN fragments of 32x32 (tile BM = 32*FM*WM, BN = 32*FN*WN):
//     64x64 (2x2 waves), 32x64 (1x2 waves, 128 threads: finer granularity against tail quantisation on 256 CUs),
//     128x64 / 128x128 (2x2 waves, 2x1 / 2x2 fragments);
//   * TWO k-tiles of global loads in flight: tile t+2 is issued before tile t is computed and is written to LDS one
//     iteration later, so a load has two tile-times (>= 2048 matrix-pipe cycles) to land (ablation in tools/gemm_sweep.py:
//     with a single tile in flight ~27 % of the kernel was exposed load latency).
// split-plane LDS rows are groups of 32 k-values, [hi 32 | lo 32] each (= the byte layout of cdetr_gemm_desc.B_split): position
// of k inside a row, in bf16 units; the lo plane of the same k sits 32 further.
#define KPOS(k) ((((k) >> 5) << 6) + ((k) & 31))
// (SPLITK_COUNTERS: common.h -- the arrival counters at the head of cdetr_gemm_desc.splitk_ws, shared with igemm_dl.hip)
// TERMS (split-bf16 staging modes only): bf16 MFMAs per algorithmic product -- 3 = hi*hi + hi*lo + lo*hi ("bf16x3", ~5e-6 relative),
// 2 = hi*hi + lo*hi (the B operand rounded to bf16, A exact to 2^-17: "bf16x2"), 1 = hi*hi (both rounded: plain bf16).  Fewer terms
// skip the MFMAs, the LDS reads of the unused lo planes and (TERMS 1) the lo half of the staging split of A.
// PREC: 0 = fp32 MFMA (exact), 1 = split-bf16 x3 with the split per fragment (the n-contiguous operand form), 2 = split once at staging,
// 3 = as 2 with B arriving pre-split from HBM.  (Measured dead ends of round 1, no longer built: a 4-deep register ring -- 0-25 % slower
// at two images per GPU --, a staging-time transpose of the n-contiguous operand, 128-wide tiles on 4 waves, 32x64 / 128x64 tiles.)
// A16: the A operand is read from its bf16 TWIN (cdetr_gemm_desc.A16; plain-bf16 products only): one 16-byte load = 8 k-values per thread,
// stored into the hi plane as it is -- half the operand bytes and no conversion at staging (the data-gradient chain of the backbone).
// KG: wave groups per workgroup that share ONE output tile and split every staged k-tile between them (KG = 2: BKF = 64, group 0 multiplies
// k 0..31 of the tile, group 1 k 32..63); the groups' accumulators meet in LDS before the epilogue.  Twice the waves per output tile and
// half the barrier-separated k-steps: for the problems whose grid cannot fill the CUs (see gemm_ksplit) without any traffic through HBM.
template <int WM, int WN, int FM, int FN, int BL, int BKF, int PREC, int TERMS = 3, bool A16 = false, int KG = 1>
__device__ __forceinline__ void igemm_fast_body(const cdetr_gemm_desc& d, const int tilesM, const int bx, const int bz,
                                                const int ksplit = 1, const int kslice = 0) {
    constexpr int PD = 2;                    // k-tiles in flight in registers
    static_assert(KG == 1 || ((PREC == 2 || PREC == 3) && (BKF / 16) % KG == 0), "wave groups exist for the staging-split forms");
    static_assert(!A16 || (TERMS == 1 && BL == 0), "the bf16 twin of A feeds the plain-bf16 products");
    static_assert(TERMS == 3 || ((PREC == 2 || PREC == 3) && BL == 0), "reduced-term products exist for the staging-split k-contiguous forms");
    static_assert(BL == 0 || PREC <= 1, "the n-contiguous operand has no staging-split form");
    constexpr int NTQ = 64 * WM * WN;        // threads of one wave group
    constexpr int NT = NTQ * KG;
    constexpr int BM = 32 * FM * WM, BN = 32 * FN * WN;
    constexpr int LDK = BKF + 4;             // 36 or 68 floats: (LDK/4) odd -> conflict-free ds_read_b128
    constexpr int F4R = BKF / 4;             // float4 per k-row (8 or 16)
    constexpr int RPP = NT / F4R;            // rows covered by one pass of the workgroup's threads
    // B rows per staging pass: normally every thread takes part (RPP rows); when the thread count does not divide BN (the 12-wave
    // 96x128 tile) only the first 64 * F4R threads stage B, 64 rows per pass -- the others re-read rows they do not keep
    constexpr int RPPB = (BN % RPP == 0) ? RPP : 64;
    constexpr int CHA = A16 ? BKF / 8 : F4R;    // 16-byte chunks per A row
    constexpr int RPPA = NT / CHA;              // A rows covered by one pass of the workgroup's threads
    constexpr int A_SLOTS = (BM + RPPA - 1) / RPPA, B_SLOTS = (BL == 0) ? BN / RPPB : BKF * BN / (4 * NT);
    static_assert(BM % RPPA == 0 || RPPA > BM, "A tile / thread mapping");
    constexpr int LDN = BN + 4;
    constexpr int A_TILE = BM * LDK;
    // PREC >= 2: the tile is split into bf16 hi / lo ONCE while it is staged; an LDS row holds [hi k0..BKF-1 | lo k0..BKF-1 | pad]
    // = the same LDK*4 bytes as the fp32 row.
    constexpr bool SPL = (PREC == 2 || PREC == 3);     // operands live in LDS as bf16 hi / lo planes
    constexpr bool BRAW = (PREC == 3);                  // B arrives pre-split from HBM (cdetr_gemm_desc.B_split): staged by a plain copy
    constexpr int B_TILE = (BL == 0) ? BN * LDK : BKF * LDN;
    static_assert((A16 || BM % RPP == 0) && (BL != 0 || BN % RPPB == 0) && (BL == 0 || (BKF * BN) % (4 * NT) == 0), "tile / thread mapping");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * A_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = (tid >> 6) % (WM * WN), kg = (tid >> 6) / (WM * WN);
    const int tidq = tid % NTQ;
    const int wm = wid / WN, wn = wid % WN;
    const int i32 = lane & 31, g = lane >> 5;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (private 4 MiB L2 each).  XCD x owns a contiguous band of
    // row-tiles and sweeps all column-tiles of a row-tile before moving on, so the A row-panel of a tile is fetched
    // from HBM/MALL once per XCD and the (small) weight matrix stays L2-resident.  Placement only affects speed.
    const int tilesN = (d.N + BN - 1) / BN;
    const int band = (tilesM + 7) >> 3;
    const int xcd = bx & 7, jloc = bx >> 3;
    const int tm = xcd * band + jloc / tilesN, tn = jloc % tilesN;
    if (tm >= tilesM) return;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = bz;
    using AElem = typename std::conditional<A16, __bf16, float>::type;
    using AVec = typename std::conditional<A16, u32x4, float4>::type;
    const AElem* __restrict__ A = (A16 ? reinterpret_cast<const AElem*>(d.A16) : reinterpret_cast<const AElem*>(d.A)) + batch_off(z, d.batch_inner, d.sA, d.sA2);
    const float* __restrict__ B = (BRAW ? reinterpret_cast<const float*>(d.B_split) : d.B) + batch_off(z, d.batch_inner, d.sB, d.sB2);
    float* __restrict__ C = d.C + batch_off(z, d.batch_inner, d.sC, d.sC2);
    __bf16* __restrict__ C16 = d.C16 ? reinterpret_cast<__bf16*>(d.C16) + batch_off(z, d.batch_inner, d.sC, d.sC2) : nullptr;   // bf16 twin of C
    __bf16* __restrict__ C16lo = (d.C16 && d.C16lo) ? reinterpret_cast<__bf16*>(d.C16lo) + batch_off(z, d.batch_inner, d.sC, d.sC2) : nullptr;   // its lo plane
    const int K = d.K, taps = d.taps;
    // split reduction (ksplit > 1): slice kslice of this output tile owns k-tiles [kt0, kt0 + nkt) of the (K / BKF) * taps
    const int nkt_all = (K / BKF) * taps;
    const int kt0 = (int)((long)nkt_all * kslice / ksplit);
    const int nkt = (int)((long)nkt_all * (kslice + 1) / ksplit) - kt0;

    const int kq = tid % F4R, r8 = tid / F4R;
    const int kqa = tid % CHA, r8a = tid / CHA;          // the A operand's own mapping (== kq, r8 unless A16)
    const bool a_on = (RPPA <= BM) || r8a < BM;
    const int r8b = (RPPB == RPP) ? r8 : r8 % RPPB;
    const bool b_on = (RPPB == RPP) || r8 < RPPB;
    // Every global load of the pipeline is UNCONDITIONAL: a predicated load sits in its own basic block, the compiler then
    // loses track of the outstanding-load count and drains vmcnt(0) before every LDS stash, which serialises the two tiles
    // in flight (measured: ~1.2 us per k-tile).  Rows that do not exist (conv padding, m >= M, n >= N) are redirected to
    // row 0 / the last row; padding rows are zeroed when they are staged (amask), the others are never stored.
    RowCoord arow[A_SLOTS];
    const AElem* ap[A_SLOTS];
    unsigned amask = 0;            // bit i: slot i of the tap being fetched is a real row
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) arow[i] = decode_row(d.g, m0 + r8a + RPPA * i, d.M);
    auto set_tap = [&](int tap) {
        amask = 0;
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const long row = gather_row(d.g, arow[i], tap);
            ap[i] = A + (row >= 0 ? row : 0) * d.lda + kqa * (A16 ? 8 : 4);
            amask |= (row >= 0 ? 1u : 0u) << i;
        }
    };
    const float* bp[B_SLOTS];
    float bscale0[B_SLOTS];
    if (BL == 0) {
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
            const int n = min(n0 + r8b + RPPB * i, d.N - 1);
            bp[i] = B + (long)n * d.ldb + kq * 4;
            bscale0[i] = (d.w_scale && !BRAW) ? d.w_scale[n] : 1.f;
        }
    }
    // register sets for the two k-tiles in flight.  Fetches are RAW loads (nothing in fetch() consumes a loaded value, so
    // no s_waitcnt lands between issuing a tile and computing on the previous one); scaling / splitting happens in stash().
    AVec ra[PD][A_SLOTS];
    unsigned rm[PD];                           // amask of each register set
    float4 rb[PD][B_SLOTS];
    auto fetch = [&](AVec (&qa)[A_SLOTS], float4 (&qb)[B_SLOTS], unsigned& qm, int tap, int kc) __attribute__((always_inline)) {
        qm = amask;
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            if constexpr (A16) qa[i] = ld16(ap[i] + kc);
            else qa[i] = ld4(ap[i] + kc);
        }
        if constexpr (BL == 0) {
            const int koff = tap * K + kc;
#pragma unroll
            for (int i = 0; i < B_SLOTS; ++i) qb[i] = ld4(bp[i] + koff);
        } else {
#pragma unroll
            for (int i = 0; i < B_SLOTS; ++i) {
                const int idx = tid + NT * i;
                const int kr = idx / (BN / 4);
                const int nq = idx - kr * (BN / 4);
                const int n = min(n0 + nq * 4, d.N - 4);
                const int k = kc + kr;
                float4 v = ld4(B + ((long)k * taps + tap) * d.ldb + n);
                if (d.w_scale) {
                    const float s = d.w_scale[k];
                    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
                }
                qb[i] = v;
            }
        }
    };
    auto stash = [&](const AVec (&qa0)[A_SLOTS], const float4 (&qb)[B_SLOTS], unsigned qm, int buf) __attribute__((always_inline)) {
        float* as = As + buf * A_TILE;
        float* bs = Bs + buf * B_TILE;
        AVec qa[A_SLOTS];
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            if constexpr (A16) { const u32x4 z4 = {0u, 0u, 0u, 0u}; qa[i] = ((qm >> i) & 1u) ? qa0[i] : z4; }
            else qa[i] = ((qm >> i) & 1u) ? qa0[i] : zero4();
        }
        if constexpr (A16) {
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i)
                if (a_on) *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(as + (r8a + RPPA * i) * LDK) + KPOS(kqa * 8)) = qa[i];
        }
        if constexpr (SPL) {
            if constexpr (!A16) {
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i) {
                if constexpr (TERMS == 1) stash_hi4(reinterpret_cast<__bf16*>(as + (r8 + RPP * i) * LDK) + KPOS(kq * 4), qa[i].x, qa[i].y, qa[i].z, qa[i].w);
                else stash_split4(reinterpret_cast<__bf16*>(as + (r8 + RPP * i) * LDK) + KPOS(kq * 4), 32, qa[i].x, qa[i].y, qa[i].z, qa[i].w);
            }
            }
            if constexpr (BRAW) {
#pragma unroll
                for (int i = 0; i < B_SLOTS; ++i)
                    if (b_on) *reinterpret_cast<float4*>(bs + (r8b + RPPB * i) * LDK + kq * 4) = qb[i];    // pre-split operand (or ablation): plain copy
            } else {
#pragma unroll
                for (int i = 0; i < B_SLOTS; ++i) {
                    const float s = bscale0[i];
                    if (b_on) stash_split4(reinterpret_cast<__bf16*>(bs + (r8b + RPPB * i) * LDK) + KPOS(kq * 4), 32, qb[i].x * s, qb[i].y * s, qb[i].z * s, qb[i].w * s);
                }
            }
        } else if constexpr (!A16) {
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) *reinterpret_cast<float4*>(as + (r8 + RPP * i) * LDK + kq * 4) = qa[i];
        if constexpr (BL == 0) {
#pragma unroll
            for (int i = 0; i < B_SLOTS; ++i) {
                const float s = bscale0[i];
                if (b_on) *reinterpret_cast<float4*>(bs + (r8b + RPPB * i) * LDK + kq * 4) = make_float4(qb[i].x * s, qb[i].y * s, qb[i].z * s, qb[i].w * s);
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_SLOTS; ++i) {
                const int idx = tid + NT * i;
                const int kr = idx / (BN / 4);
                const int nq = idx - kr * (BN / 4);
                *reinterpret_cast<float4*>(bs + kr * LDN + nq * 4) = qb[i];
            }
        }
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float* as = As + buf * A_TILE + (wm * (32 * FM) + i32) * LDK + g * 4;
        const float* bs = (BL == 0) ? Bs + buf * B_TILE + (wn * (32 * FN) + i32) * LDK + g * 4
                                    : Bs + buf * B_TILE + (g * 4) * LDN + wn * (32 * FN) + i32;
        if (PREC == 0) {
#pragma unroll
            for (int h = 0; h < BKF / 8; ++h) {
                float a4[FM][4], b4[FN][4];
#pragma unroll
                for (int a = 0; a < FM; ++a) {
                    const float4 t = *reinterpret_cast<const float4*>(as + a * 32 * LDK + h * 8);
                    a4[a][0] = t.x; a4[a][1] = t.y; a4[a][2] = t.z; a4[a][3] = t.w;
                }
#pragma unroll
                for (int b = 0; b < FN; ++b) {
                    if (BL == 0) {
                        const float4 t = *reinterpret_cast<const float4*>(bs + b * 32 * LDK + h * 8);
                        b4[b][0] = t.x; b4[b][1] = t.y; b4[b][2] = t.z; b4[b][3] = t.w;
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s) b4[b][s] = bs[(h * 8 + s) * LDN + b * 32];
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int a = 0; a < FM; ++a)
#pragma unroll
                        for (int b = 0; b < FN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[a][s], b4[b][s], acc[a][b], 0, 0, 0);
            }
        } else {
            if constexpr (PREC == 1) {
                // split-bf16: 16 k per step; lane group g supplies k = 16hp + {4g..4g+3, 8+4g..8+4g+3} for A and B alike
    #pragma unroll
                for (int hp = 0; hp < BKF / 16; ++hp) {
                    bf16x8 ah[FM], al[FM], bh[FN], bl[FN];
    #pragma unroll
                    for (int a = 0; a < FM; ++a) {
                        const float4 t0 = *reinterpret_cast<const float4*>(as + a * 32 * LDK + hp * 16);
                        const float4 t1 = *reinterpret_cast<const float4*>(as + a * 32 * LDK + hp * 16 + 8);
                        const float x[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                        split_bf16x8(x, ah[a], al[a]);
                    }
    #pragma unroll
                    for (int b = 0; b < FN; ++b) {
                        float x[8];
                        if (BL == 0) {
                            const float4 t0 = *reinterpret_cast<const float4*>(bs + b * 32 * LDK + hp * 16);
                            const float4 t1 = *reinterpret_cast<const float4*>(bs + b * 32 * LDK + hp * 16 + 8);
                            x[0] = t0.x; x[1] = t0.y; x[2] = t0.z; x[3] = t0.w; x[4] = t1.x; x[5] = t1.y; x[6] = t1.z; x[7] = t1.w;
                        } else {
    #pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                x[s] = bs[(hp * 16 + s) * LDN + b * 32];
                                x[4 + s] = bs[(hp * 16 + 8 + s) * LDN + b * 32];
                            }
                        }
                        split_bf16x8(x, bh[b], bl[b]);
                    }
    #pragma unroll
                    for (int a = 0; a < FM; ++a)
    #pragma unroll
                        for (int b = 0; b < FN; ++b) acc[a][b] = mfma_bf16x3(ah[a], al[a], bh[b], bl[b], acc[a][b]);
                }
            } else {
                // split-bf16: 16 k per step; lane group g supplies k = 16hp + 8g .. 8g+7 for A and B alike: one ds_read_b128
                // of the hi plane and one of the lo plane per fragment, no VALU work between LDS and the matrix pipe
                const __bf16* a16 = reinterpret_cast<const __bf16*>(As + buf * A_TILE + (wm * (32 * FM) + i32) * LDK) + g * 8;
                const __bf16* b16 = reinterpret_cast<const __bf16*>(Bs + buf * B_TILE + (wn * (32 * FN) + i32) * LDK) + g * 8;
    #pragma unroll
                for (int hq = 0; hq < BKF / 16 / KG; ++hq) {
                    const int hp = kg * (BKF / 16 / KG) + hq;        // this wave group's share of the staged k-tile
                    bf16x8 ah[FM], al[FM], bh[FN], bl[FN];
    #pragma unroll
                    for (int a = 0; a < FM; ++a) {
                        ah[a] = *reinterpret_cast<const bf16x8*>(a16 + a * 32 * 2 * LDK + KPOS(hp * 16));
                        if constexpr (TERMS >= 2) al[a] = *reinterpret_cast<const bf16x8*>(a16 + a * 32 * 2 * LDK + KPOS(hp * 16) + 32);
                    }
    #pragma unroll
                    for (int b = 0; b < FN; ++b) {
                        bh[b] = *reinterpret_cast<const bf16x8*>(b16 + b * 32 * 2 * LDK + KPOS(hp * 16));
                        if constexpr (TERMS == 3) bl[b] = *reinterpret_cast<const bf16x8*>(b16 + b * 32 * 2 * LDK + KPOS(hp * 16) + 32);
                    }
    #pragma unroll
                    for (int a = 0; a < FM; ++a)
    #pragma unroll
                        for (int b = 0; b < FN; ++b) {
                            if constexpr (TERMS == 3) acc[a][b] = mfma_bf16x3(ah[a], al[a], bh[b], bl[b], acc[a][b]);
                            else {
                                if constexpr (TERMS == 2) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                            }
                        }
                }
            }
        }
    };

    // ---- software pipeline: LDS[t & 1] holds tile t, register set (t + j) % PD holds tile t + j (j = 1 .. PD-1), tile t + PD is
    // being issued into the set tile t was staged from.  PD = 4 keeps three tiles in flight behind the one being computed: a
    // GEMM of this model rarely has more than one or two workgroups per CU, so bytes in flight per CU -- not the matrix pipe --
    // set the pace of the k-loop (Little: ~2 us of L2/HBM latency x the bandwidth a CU needs).
    // The loop body is branch-free around the loads (every step fetches; past the end the last tile is fetched again and
    // never used): with conditional fetches the loaded registers become loop PHIs, the allocator copies them right after
    // the load is issued and the copy drags an s_waitcnt vmcnt(~0) in front of the MFMA block -- one tile in flight at best.
    int f_tap = kt0 / (K / BKF), f_kc = (kt0 % (K / BKF)) * BKF, f_idx = 0;       // coordinates of the most recently issued tile
    auto advance = [&]() {
        if (f_idx + 1 < nkt) {
            ++f_idx;
            f_kc += BKF;
            if (f_kc == K) { f_kc = 0; ++f_tap; set_tap(f_tap); }
        }
    };
    auto step = [&](auto sc) __attribute__((always_inline)) {      // tile t (t % PD == S): issue t + PD, compute t, stage t + 1
        constexpr int S = decltype(sc)::value, NX = (S + 1) % PD;
        advance();
        fetch(ra[S], rb[S], rm[S], f_tap, f_kc);
        compute(S & 1);
        stash(ra[NX], rb[NX], rm[NX], NX & 1);
        __syncthreads();
    };
    auto drain_step = [&](auto sc) __attribute__((always_inline)) { // tail: stage tile t (already in registers) and compute it
        constexpr int S = decltype(sc)::value;
        stash(ra[S], rb[S], rm[S], S & 1);
        __syncthreads();
        compute(S & 1);
    };
    set_tap(f_tap);
    fetch(ra[0], rb[0], rm[0], f_tap, f_kc);
#pragma unroll
    for (int j = 1; j < PD; ++j) {
        advance();
        fetch(ra[j], rb[j], rm[j], f_tap, f_kc);
    }
    stash(ra[0], rb[0], rm[0], 0);
    __syncthreads();
    int kt = 0;
    for (; kt + PD < nkt; kt += PD) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
    }
    // residual / gate operands of this wave's outputs: ONE batch of unconditional loads per fragment (clamped row / column, masked
    // at the store), issued before the last k-tiles are computed so the round trip overlaps them.  (A load inside the per-element
    // `if (m < M)` of the store loop sits in its own basic block and the compiler then waits for each of the 16 round trips in turn:
    // measured with cold operands, tools/cold_gemm.py, that doubled every GEMM with a residual -- 20000x512x128: 30 -> 64 us.)
    float rv[FM][FN][16], gv[FM][FN][16];
    auto load_rg = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < FM; ++a) {
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            const int nc = min(n0 + wn * (32 * FN) + b * 32 + i32, d.N - 1);
            const int mbase = m0 + wm * (32 * FM) + a * 32 + 4 * g;
#pragma unroll
            for (int r = 0; r < 16; ++r) { rv[a][b][r] = 0.f; gv[a][b][r] = 1.f; }
            if (d.resid) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[a][b][r] = d.resid[(long)min(mbase + (r & 3) + 8 * (r >> 2), d.M - 1) * d.ldr + nc];
            }
            if (d.gate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) gv[a][b][r] = d.gate[(long)min(mbase + (r & 3) + 8 * (r >> 2), d.M - 1) * d.ldg + nc];
            }
        }
    }
    };
    if (ksplit == 1 && kg == 0) load_rg();   // a split tile fetches them once it knows it is the one that finishes the tile
    const int rem = nkt - kt;                                      // 1 .. PD tiles left: tile kt is staged, the others sit in registers
    compute(0);
    if (rem > 1) drain_step(std::integral_constant<int, 1>{});
    mfma_drain(acc);

    if constexpr (KG > 1) {
        // the wave groups' partial tiles meet in LDS (the tile buffers are free once every wave is past its last read); group 0 carries on
        __syncthreads();
        float* red = smem;
        if (kg > 0) {
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
                for (int b = 0; b < FN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(((kg - 1) * FM * FN + a * FN + b) * 16 + r) * NTQ + tidq] = acc[a][b][r];
        }
        __syncthreads();
        if (kg > 0) return;                  // (finished waves leave the workgroup's barrier count)
#pragma unroll
        for (int q = 0; q < KG - 1; ++q)
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
                for (int b = 0; b < FN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] += red[((q * FM * FN + a * FN + b) * 16 + r) * NTQ + tidq];
    }

    if (ksplit > 1) {
        // ---- split reduction: every slice parks its partial tile in the scratch (fragment order: element e of thread t at [e][t], fully
        // coalesced), then counts itself in; the slice that arrives last adds the partials IN SLICE ORDER (its own from registers) -- the
        // same sum whichever slice that is -- and carries on into the fused epilogue.  No slice ever waits for another.
        // Coherence across the 8 XCDs (one L2 each) WITHOUT fences: a device-scope fence writes back / invalidates the whole L2
        // (measured: +80 us per launch).  The partials travel as relaxed device-scope atomic stores / loads (sc1: written through to,
        // and read from, the level all XCDs share), s_waitcnt vmcnt(0) orders them before the arrival count, which is a device-scope RMW.
        constexpr int FR = FM * FN * 16;
        int* cnt = reinterpret_cast<int*>(d.splitk_ws);
        float* wsp = reinterpret_cast<float*>(cnt + SPLITK_COUNTERS);
        const int tile = tm * tilesN + tn;
        float* mine = wsp + ((long)tile * ksplit + kslice) * FR * NTQ + tidq;
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int b = 0; b < FN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __hip_atomic_store(mine + ((a * FN + b) * 16 + r) * NTQ, acc[a][b][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);       // the stores above have been acknowledged
        __syncthreads();                     // ... by every wave; and every wave is past its last LDS read: the tile buffers are free
        int* flag = reinterpret_cast<int*>(smem);
        if (tidq == 0) {
            const int old = __hip_atomic_fetch_add(cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (old == ksplit - 1) ? 1 : 0;
            if (last) __hip_atomic_store(cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // leave the counters zero for the next launch
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0) return;
        load_rg();
        float sum[FR];
#pragma unroll
        for (int e = 0; e < FR; ++e) sum[e] = 0.f;
        for (int q = 0; q < ksplit; ++q) {
            if (q == kslice) {
#pragma unroll
                for (int a = 0; a < FM; ++a)
#pragma unroll
                    for (int b = 0; b < FN; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sum[(a * FN + b) * 16 + r] += acc[a][b][r];
            } else {
                const float* other = wsp + ((long)tile * ksplit + q) * FR * NTQ + tidq;
                float t[FR];
#pragma unroll
                for (int e = 0; e < FR; ++e) t[e] = __hip_atomic_load(other + e * NTQ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int e = 0; e < FR; ++e) sum[e] += t[e];
            }
        }
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int b = 0; b < FN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = sum[(a * FN + b) * 16 + r];
    }

#pragma unroll
    for (int a = 0; a < FM; ++a) {
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            const int n = n0 + wn * (32 * FN) + b * 32 + i32;
            const bool nvalid = n < d.N;
            const float bias = d.bias ? d.bias[nvalid ? n : d.N - 1] : 0.f;
            const int mbase = m0 + wm * (32 * FM) + a * 32 + 4 * g;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                float v = (acc[a][b][r] + bias) * d.out_scale + rv[a][b][r];
                v = gv[a][b][r] > 0.f ? v : 0.f;
                if (d.relu) v = fmaxf(v, 0.f);
                if (nvalid && m < d.M) {
                    C[(long)m * d.ldc + n] = v;
                    if (C16) {
                        const __bf16 h = (__bf16)v;
                        C16[(long)m * d.ldc + n] = h;
                        if (C16lo) C16lo[(long)m * d.ldc + n] = (__bf16)(v - (float)h);
                    }
                }
            }
        }
    }
}

template <int WM, int WN, int FM, int FN, int BL, int BKF, int PREC, int TERMS = 3, bool A16 = false, int KG = 1>
__global__ __launch_bounds__(64 * WM * WN * KG) void igemm_fast_kernel(const cdetr_gemm_desc d, const int tilesM) {
    igemm_fast_body<WM, WN, FM, FN, BL, BKF, PREC, TERMS, A16, KG>(d, tilesM, blockIdx.x, blockIdx.z, gridDim.y, blockIdx.y);
}

// Grouped launch of up to GG_MAX independent GEMMs of one kernel class (same idea as WgradGroupArgs below): the problems'
// workgroups are concatenated along grid.x, a workgroup finds its problem with a short scalar scan of the kernel-argument table.
constexpr int GG_MAX = 12;
struct GemmGroupItem { cdetr_gemm_desc d; int tilesM, tilesN, vecA, vecB, nx, pad_; };
struct GemmGroupArgs { int n; int blk0[GG_MAX + 1]; GemmGroupItem it[GG_MAX]; };
static_assert(sizeof(GemmGroupArgs) <= 4000, "grouped launch arguments must fit the kernel-argument segment");

template <int WM, int WN, int FM, int FN, int BL, int BKF, int PREC, int TERMS = 3>
__global__ __launch_bounds__(64 * WM * WN) void igemm_fast_group_kernel(const GemmGroupArgs g) {
    int p = 0;
    while (p + 1 < g.n && (int)blockIdx.x >= g.blk0[p + 1]) ++p;
    const GemmGroupItem& it = g.it[p];
    const int lb = blockIdx.x - g.blk0[p];             // nx is a multiple of 8: the XCD banding of the body is preserved
    igemm_fast_body<WM, WN, FM, FN, BL, BKF, PREC, TERMS>(it.d, it.tilesM, lb % it.nx, lb / it.nx);
}

// ------------------------------------------------------------------------------------------------ fast wgrad
// dW[i][tap][c] += scale[i] * sum_p dY[p][i] * X[row(p,tap)][c]  (+ optional dbias[i] += sum_p dY[p][i]).
// BK = 32 pixels per tile; every thread owns ONE pixel row of the tile (8 threads x 16 B per 128-byte run), so the
// pixel -> (n, y, x) coordinates are advanced incrementally (no division) and the tap gather costs a few integer ops.
template <int BI, int BJ, int PREC>
__global__ __launch_bounds__(256) void wgrad_fast_kernel(const cdetr_wgrad_desc d, const int tilesI, const int tilesJ,
                                                         const int kt_per_slice, float* __restrict__ dbias) {
    constexpr int BKF = 32;
    constexpr int FM = BI / 64, FN = BJ / 64;
    constexpr int A_SLOTS = BI / 32, B_SLOTS = BJ / 32;
    constexpr int LDI = BI + 4, LDJ = BJ + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [2][32][LDI]
    float* Bs = smem + 2 * BKF * LDI;      // [2][32][LDJ]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int i32 = lane & 31, g = lane >> 5;
    const int ti = blockIdx.x % tilesI;
    const int tj = blockIdx.x / tilesI;
    const int tap = tj / tilesJ;
    const int c0 = (tj - tap * tilesJ) * BJ;
    const int i0 = ti * BI;
    const int z = blockIdx.z;
    const float* __restrict__ dY = d.dY + batch_off(z, d.batch_inner, d.sY, d.sY2);
    const float* __restrict__ X = d.X + batch_off(z, d.batch_inner, d.sX, d.sX2);
    float* __restrict__ dW = d.dW + batch_off(z, d.batch_inner, d.sW, d.sW2);
    const int nkt_all = (d.P + BKF - 1) / BKF;
    const int kt_begin = blockIdx.y * kt_per_slice;
    const int kt_end = min(nkt_all, kt_begin + kt_per_slice);
    if (kt_begin >= kt_end) return;

    const int kr = tid >> 3, cq = (tid & 7) * 4;
    const bool dense = d.g.mode == CDETR_ROWS_DENSE;
    const int ky = dense ? 0 : tap / d.g.kw, kx = dense ? 0 : tap - (tap / d.g.kw) * d.g.kw;
    // this thread's pixel for the first tile
    int p = kt_begin * BKF + kr;
    int pn = 0, py = 0, px = 0;
    if (!dense) {
        const int hw = d.g.Hc * d.g.Wc;
        pn = p / hw;
        const int rem = p - pn * hw;
        py = rem / d.g.Wc;
        px = rem - py * d.g.Wc;
    }
    const bool do_bias = (dbias != nullptr) && tj == 0;
    float4 bsum[A_SLOTS];
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) bsum[s] = zero4();

    // Two register sets = two pixel tiles in flight.  Loads are unconditional (see igemm_fast_kernel): a pixel beyond P or a
    // padding tap reads a clamped address and is zeroed when it is staged (flag bits), column tails are clamped and never stored.
    float4 ra[2][A_SLOTS], rb[2][B_SLOTS];
    unsigned rf[2] = {0, 0};                 // bit 0: dY row is real, bit 1: X row is real
    int acol[A_SLOTS], bcol[B_SLOTS];
    const int nk = kt_end - kt_begin;
    int ft = 0;                              // tiles fetched so far
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) acol[s] = min(i0 + cq + 32 * s, d.Nout - 4);
#pragma unroll
    for (int s = 0; s < B_SLOTS; ++s) bcol[s] = min(c0 + cq + 32 * s, d.Cin - 4);
    auto fetch = [&](float4 (&qa)[A_SLOTS], float4 (&qb)[B_SLOTS], unsigned& qf) __attribute__((always_inline)) {
        // loads the tile whose pixel row for this thread is (p; pn, py, px), then advances by 32 pixels
        const bool pv = p < d.P;
        const float* yp = dY + (long)(pv ? p : d.P - 1) * d.ldy;
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) qa[s] = ld4(yp + acol[s]);
        long row = -1;
        if (pv) {
            if (dense) row = p;
            else {
                const int iy = py * d.g.stride - d.g.pad + ky * d.g.dil;
                const int ix = px * d.g.stride - d.g.pad + kx * d.g.dil;
                if (iy >= 0 && iy < d.g.Ha && ix >= 0 && ix < d.g.Wa) row = ((long)pn * d.g.Ha + iy) * d.g.Wa + ix;
            }
        }
        qf = ((pv && ft < nk) ? 1u : 0u) | (row >= 0 ? 2u : 0u);     // surplus tiles past the slice end stage zeros
        ++ft;
        const float* xp = X + (row >= 0 ? row : 0) * d.ldx;
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) qb[s] = ld4(xp + bcol[s]);
        p += BKF;
        if (!dense) {
            px += BKF;
            while (px >= d.g.Wc) { px -= d.g.Wc; ++py; }
            while (py >= d.g.Hc) { py -= d.g.Hc; ++pn; }
        }
    };
    auto stash = [&](const float4 (&qa)[A_SLOTS], const float4 (&qb)[B_SLOTS], unsigned qf, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) {
            const float4 v = (qf & 1u) ? qa[s] : zero4();
            *reinterpret_cast<float4*>(As + buf * BKF * LDI + kr * LDI + cq + 32 * s) = v;
            if (do_bias) { bsum[s].x += v.x; bsum[s].y += v.y; bsum[s].z += v.z; bsum[s].w += v.w; }
        }
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s)
            *reinterpret_cast<float4*>(Bs + buf * BKF * LDJ + kr * LDJ + cq + 32 * s) = (qf & 2u) ? qb[s] : zero4();
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float* as = As + buf * BKF * LDI + (g * 4) * LDI + wm * (BI / 2) + i32;
        const float* bs = Bs + buf * BKF * LDJ + (g * 4) * LDJ + wn * (BJ / 2) + i32;
        if (PREC == 0) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float av[FM], bv[FN];
#pragma unroll
                    for (int a = 0; a < FM; ++a) av[a] = as[(h * 8 + s) * LDI + a * 32];
#pragma unroll
                    for (int b = 0; b < FN; ++b) bv[b] = bs[(h * 8 + s) * LDJ + b * 32];
#pragma unroll
                    for (int a = 0; a < FM; ++a)
#pragma unroll
                        for (int b = 0; b < FN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
                bf16x8 ah[FM], al[FM], bh[FN], bl[FN];
#pragma unroll
                for (int a = 0; a < FM; ++a) {
                    float x[8];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        x[s] = as[(hp * 16 + s) * LDI + a * 32];
                        x[4 + s] = as[(hp * 16 + 8 + s) * LDI + a * 32];
                    }
                    split_bf16x8(x, ah[a], al[a]);
                }
#pragma unroll
                for (int b = 0; b < FN; ++b) {
                    float x[8];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        x[s] = bs[(hp * 16 + s) * LDJ + b * 32];
                        x[4 + s] = bs[(hp * 16 + 8 + s) * LDJ + b * 32];
                    }
                    split_bf16x8(x, bh[b], bl[b]);
                }
#pragma unroll
                for (int a = 0; a < FM; ++a)
#pragma unroll
                    for (int b = 0; b < FN; ++b) acc[a][b] = mfma_bf16x3(ah[a], al[a], bh[b], bl[b], acc[a][b]);
            }
        }
    };

    // branch-free two-deep pipeline: tiles past the slice end have p >= their bound only at the global end (zeroed);
    // inside the loop every step fetches, the surplus fetches of the last steps are never staged
    fetch(ra[0], rb[0], rf[0]);
    fetch(ra[1], rb[1], rf[1]);
    stash(ra[0], rb[0], rf[0], 0);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        fetch(ra[0], rb[0], rf[0]);            // tile kt+2
        compute(0);
        stash(ra[1], rb[1], rf[1], 1);
        __syncthreads();
        fetch(ra[1], rb[1], rf[1]);            // tile kt+3
        compute(1);
        stash(ra[0], rb[0], rf[0], 0);
        __syncthreads();
    }
    if (kt < nk) compute(0);

    mfma_drain(acc);
    const bool single = (gridDim.y == 1);
#pragma unroll
    for (int a = 0; a < FM; ++a) {
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            const int c = c0 + wn * (BJ / 2) + b * 32 + i32;
            if (c >= d.Cin) continue;
            float ws[16];                  // per-row weight scales as one batch of unconditional loads (clamped row)
#pragma unroll
            for (int r = 0; r < 16; ++r) ws[r] = 1.f;
            if (d.w_scale) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ws[r] = d.w_scale[min(i0 + wm * (BI / 2) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, d.Nout - 1)];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wm * (BI / 2) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (i >= d.Nout) continue;
                float v = acc[a][b][r] * ws[r];
                float* dst = dW + (long)i * d.ldw + (long)tap * d.Cin + c;
                if (single) *dst += v;          // this launch owns the element: plain read-modify-write
                else atomicAdd(dst, v);
            }
        }
    }
    if (do_bias) {   // column sums of dY: reduce the 32 pixel rows of the block through LDS, one atomic per column
        __syncthreads();
        float* red = smem;   // [32][BI]
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) *reinterpret_cast<float4*>(red + kr * BI + cq + 32 * s) = bsum[s];
        __syncthreads();
        for (int i = tid; i < BI; i += 256) {
            float t = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) t += red[r * BI + i];
            if (i0 + i < d.Nout) atomicAdd(dbias + i0 + i, t);
        }
    }
}



// ------------------------------------------------------------------------------------------------ wgrad, transpose-read form
// split-bf16 only.  The reduction runs over pixels while both operands are channel-contiguous, so an MFMA fragment (8
// consecutive k = pixels per lane) is a TRANSPOSED view of the natural tile.  gfx950's ds_read_b64_tr_b16 does that transpose
// in the LDS read path: within a 16-lane group, lane 4a+b receives element b of the 8-byte chunks addressed by lanes a, 4+a,
// 8+a, 12+a (measured: tools/probe/tr_probe.hip).  So
//   * the stash keeps the coalesced shape of wgrad_fast_kernel (thread = one pixel row x 4 channels, 128-byte runs) and only
//     splits: packed hi / lo conversion + one ds_write_b64 per plane into [32-channel block][pixel][32] bf16 images (64-byte rows:
//     a half-wave's transpose read covers 4 rows x 64 B = 256 contiguous bytes, one pass over the 64 banks);
//   * a fragment is 4 transpose reads (hi, lo x two 4-pixel groups) and no VALU: ~8 VALU per MFMA instead of ~30.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 lds_tr8(const __bf16* p0, const __bf16* p1) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
    const s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, c);
}

// TERMS: bf16 MFMAs per product, as in igemm_fast_body (3 = bf16x3; 2 = X rounded to bf16, dY split; 1 = plain bf16).
template <int BI, int BJ, int TERMS = 3>
__device__ __forceinline__ void wgrad_tr_body(const cdetr_wgrad_desc& d, const int tilesI, const int tilesJ, const int kt_per_slice,
                                              float* __restrict__ dbias, const int bx, const int by, const int bz, const bool single) {
    constexpr int BKF = 32;
    constexpr int FM = BI / 64, FN = BJ / 64;
    constexpr int A_SLOTS = BI / 32, B_SLOTS = BJ / 32;
    constexpr int BLK = 32 * 32;                                  // bf16 per [32 pixels][32 channels] block
    constexpr int PLANE_A = A_SLOTS * BLK, PLANE_B = B_SLOTS * BLK;
    constexpr int BUF = 2 * (PLANE_A + PLANE_B);                  // per buffer: dY hi | dY lo | X hi | X lo
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* Zs = reinterpret_cast<__bf16*>(smem);                 // [2][BUF]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int i32 = lane & 31, g = lane >> 5;
    const int ti = bx % tilesI;
    const int tj = bx / tilesI;
    const int tap = tj / tilesJ;
    const int c0 = (tj - tap * tilesJ) * BJ;
    const int i0 = ti * BI;
    const int z = bz;
    const float* __restrict__ dY = d.dY + batch_off(z, d.batch_inner, d.sY, d.sY2);
    const float* __restrict__ X = d.X + batch_off(z, d.batch_inner, d.sX, d.sX2);
    float* __restrict__ dW = d.dW + batch_off(z, d.batch_inner, d.sW, d.sW2);
    const int nkt_all = (d.P + BKF - 1) / BKF;
    const int kt_begin = by * kt_per_slice;
    const int kt_end = min(nkt_all, kt_begin + kt_per_slice);
    if (kt_begin >= kt_end) return;

    const int kr = tid >> 3, cq = (tid & 7) * 4;
    const bool dense = d.g.mode == CDETR_ROWS_DENSE;
    const int ky = dense ? 0 : tap / d.g.kw, kx = dense ? 0 : tap - (tap / d.g.kw) * d.g.kw;
    int p = kt_begin * BKF + kr;
    int pn = 0, py = 0, px = 0;
    if (!dense) {
        const int hw = d.g.Hc * d.g.Wc;
        pn = p / hw;
        const int rem = p - pn * hw;
        py = rem / d.g.Wc;
        px = rem - py * d.g.Wc;
    }
    const bool do_bias = (dbias != nullptr) && tj == 0;
    float4 bsum[A_SLOTS];
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) bsum[s] = zero4();

    // two register sets = two pixel tiles in flight; unconditional clamped loads + validity bits (see wgrad_fast_kernel)
    float4 ra[2][A_SLOTS], rb[2][B_SLOTS];
    unsigned rf[2] = {0, 0};
    int acol[A_SLOTS], bcol[B_SLOTS];
    const int nk = kt_end - kt_begin;
    int ft = 0;
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) acol[s] = min(i0 + cq + 32 * s, d.Nout - 4);
#pragma unroll
    for (int s = 0; s < B_SLOTS; ++s) bcol[s] = min(c0 + cq + 32 * s, d.Cin - 4);
    auto fetch = [&](float4 (&qa)[A_SLOTS], float4 (&qb)[B_SLOTS], unsigned& qf) __attribute__((always_inline)) {
        const bool pv = p < d.P;
        const float* yp = dY + (long)(pv ? p : d.P - 1) * d.ldy;
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) qa[s] = ld4(yp + acol[s]);
        long row = -1;
        if (pv) {
            if (dense) row = p;
            else {
                const int iy = py * d.g.stride - d.g.pad + ky * d.g.dil;
                const int ix = px * d.g.stride - d.g.pad + kx * d.g.dil;
                if (iy >= 0 && iy < d.g.Ha && ix >= 0 && ix < d.g.Wa) row = ((long)pn * d.g.Ha + iy) * d.g.Wa + ix;
            }
        }
        qf = ((pv && ft < nk) ? 1u : 0u) | (row >= 0 ? 2u : 0u);
        ++ft;
        const float* xp = X + (row >= 0 ? row : 0) * d.ldx;
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) qb[s] = ld4(xp + bcol[s]);
        p += BKF;
        if (!dense) {
            px += BKF;
            while (px >= d.g.Wc) { px -= d.g.Wc; ++py; }
            while (py >= d.g.Hc) { py -= d.g.Hc; ++pn; }
        }
    };
    __bf16* const wbase = Zs + kr * 32 + cq;
    auto stash = [&](const float4 (&qa)[A_SLOTS], const float4 (&qb)[B_SLOTS], unsigned qf, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) {
            const float4 v = (qf & 1u) ? qa[s] : zero4();
            if constexpr (TERMS == 1) stash_hi4(wbase + buf * BUF + s * BLK, v.x, v.y, v.z, v.w);
            else stash_split4(wbase + buf * BUF + s * BLK, PLANE_A, v.x, v.y, v.z, v.w);
            if (do_bias) { bsum[s].x += v.x; bsum[s].y += v.y; bsum[s].z += v.z; bsum[s].w += v.w; }
        }
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) {
            const float4 v = (qf & 2u) ? qb[s] : zero4();
            if constexpr (TERMS < 3) stash_hi4(wbase + buf * BUF + 2 * PLANE_A + s * BLK, v.x, v.y, v.z, v.w);
            else stash_split4(wbase + buf * BUF + 2 * PLANE_A + s * BLK, PLANE_B, v.x, v.y, v.z, v.w);
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transpose-read source address of this lane: 16-lane group q -> channel half (q & 1), pixel octet (q >> 1) = the MFMA k-group;
    // inside the group lane 4j + a addresses pixel j, channels 4a .. 4a+3
    const int l16 = lane & 15;
    const int roff = ((g * 8 + (l16 >> 2)) * 32) + ((lane >> 4) & 1) * 16 + (l16 & 3) * 4;
    const __bf16* const abase = Zs + (wm * FM) * BLK + roff;
    const __bf16* const bbase = Zs + 2 * PLANE_A + (wn * FN) * BLK + roff;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const __bf16* as = abase + buf * BUF;
        const __bf16* bs = bbase + buf * BUF;
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            bf16x8 ah[FM], al[FM], bh[FN], bl[FN];
#pragma unroll
            for (int a = 0; a < FM; ++a) {
                ah[a] = lds_tr8(as + a * BLK + hp * 512, as + a * BLK + hp * 512 + 128);
                if constexpr (TERMS >= 2) al[a] = lds_tr8(as + PLANE_A + a * BLK + hp * 512, as + PLANE_A + a * BLK + hp * 512 + 128);
            }
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                bh[b] = lds_tr8(bs + b * BLK + hp * 512, bs + b * BLK + hp * 512 + 128);
                if constexpr (TERMS == 3) bl[b] = lds_tr8(bs + PLANE_B + b * BLK + hp * 512, bs + PLANE_B + b * BLK + hp * 512 + 128);
            }
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
                for (int b = 0; b < FN; ++b) {
                    if constexpr (TERMS == 3) acc[a][b] = mfma_bf16x3(ah[a], al[a], bh[b], bl[b], acc[a][b]);
                    else {
                        if constexpr (TERMS == 2) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                    }
                }
        }
    };

    fetch(ra[0], rb[0], rf[0]);
    fetch(ra[1], rb[1], rf[1]);
    stash(ra[0], rb[0], rf[0], 0);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        fetch(ra[0], rb[0], rf[0]);            // tile kt+2
        compute(0);
        stash(ra[1], rb[1], rf[1], 1);
        __syncthreads();
        fetch(ra[1], rb[1], rf[1]);            // tile kt+3
        compute(1);
        stash(ra[0], rb[0], rf[0], 0);
        __syncthreads();
    }
    if (kt < nk) compute(0);

    mfma_drain(acc);
#pragma unroll
    for (int a = 0; a < FM; ++a) {
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            const int c = c0 + wn * (BJ / 2) + b * 32 + i32;
            if (c >= d.Cin) continue;
            float ws[16];                  // per-row weight scales as one batch of unconditional loads (clamped row)
#pragma unroll
            for (int r = 0; r < 16; ++r) ws[r] = 1.f;
            if (d.w_scale) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ws[r] = d.w_scale[min(i0 + wm * (BI / 2) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, d.Nout - 1)];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wm * (BI / 2) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (i >= d.Nout) continue;
                float v = acc[a][b][r] * ws[r];
                float* dst = dW + (long)i * d.ldw + (long)tap * d.Cin + c;
                if (single) *dst += v;
                else atomicAdd(dst, v);
            }
        }
    }
    if (do_bias) {   // column sums of dY: reduce the 32 pixel rows of the block through LDS, one atomic per column
        __syncthreads();
        float* red = smem;   // [32][BI]
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) *reinterpret_cast<float4*>(red + kr * BI + cq + 32 * s) = bsum[s];
        __syncthreads();
        for (int i = tid; i < BI; i += 256) {
            float t = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) t += red[r * BI + i];
            if (i0 + i < d.Nout) atomicAdd(dbias + i0 + i, t);
        }
    }
}

// The same kernel fed from bf16 TWINS of its operands (cdetr_wgrad_desc.dY16 / X16: the producers' epilogues write a bf16 copy of every
// activation / activation gradient a plain-bf16 weight gradient consumes).  Half the L2 -> register bytes per k-tile (one 16-byte load =
// 8 channels per thread and operand instead of two 4-channel loads), no conversion at staging (the 16 bytes go straight into the
// [32-channel block][pixel][32] plane with one ds_write_b128), hi planes only.  Plain bf16 products (TERMS 1) by construction.


// KP: 32-pixel blocks per staged k-tile (round 5).  A workgroup's slice is only 10-50 blocks long and a block is two MFMAs per fragment pair: with
// KP = 1 the loop is a barrier + a load round trip every 2 (64x64) to 8 (128x128) MFMAs per wave.  KP = 2 / 4 stages 64 / 128 pixels per
// barrier (KP x the loads in flight per thread, KP x the MFMAs between two barriers); the host keeps counting slices in 32-pixel blocks.
template <int BI, int BJ, int KP = 1>
__device__ __forceinline__ void wgrad_tr16_body(const cdetr_wgrad_desc& d, const int tilesI, const int tilesJ, const int kt_per_slice,
                                                const int bx, const int by, const int bz, const bool single) {
    constexpr int BKF = 32 * KP;
    constexpr int FM = BI / 64, FN = BJ / 64;
    constexpr int A_BLK = BI / 32, B_BLK = BJ / 32;               // 32-channel blocks per operand tile
    constexpr int A_SLOTS = BI / 64, B_SLOTS = BJ / 64;            // 256 threads = 32 pixels x 8 chunks of 8 channels (128 contiguous bytes per pixel row) per pass
    constexpr int BLK = BKF * 32 + 32;                            // bf16 per [BKF pixels][32 channels] block + 64 B: the two blocks a pixel row's
                                                                  // 8 lanes write to land on different banks
    constexpr int PLANE_A = A_BLK * BLK, PLANE_B = B_BLK * BLK;
    constexpr int BUF = PLANE_A + PLANE_B;                        // per buffer: dY | X (hi planes only)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* Zs = reinterpret_cast<__bf16*>(smem);                 // [2][BUF]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int i32 = lane & 31, g = lane >> 5;
    const int ti = bx % tilesI;
    const int tj = bx / tilesI;
    const int tap = tj / tilesJ;
    const int c0 = (tj - tap * tilesJ) * BJ;
    const int i0 = ti * BI;
    const int z = bz;
    const __bf16* __restrict__ dY = reinterpret_cast<const __bf16*>(d.dY16) + batch_off(z, d.batch_inner, d.sY, d.sY2);
    const __bf16* __restrict__ X = reinterpret_cast<const __bf16*>(d.X16) + batch_off(z, d.batch_inner, d.sX, d.sX2);
    float* __restrict__ dW = d.dW + batch_off(z, d.batch_inner, d.sW, d.sW2);
    const int nkt_all = (d.P + 31) / 32;
    const int kt_begin = by * kt_per_slice;
    const int kt_end = min(nkt_all, kt_begin + kt_per_slice);
    if (kt_begin >= kt_end) return;
    const int p_end = min(d.P, kt_end * 32);                      // this slice: pixels [32 kt_begin, p_end)

    const int kr = tid >> 3, c8 = (tid & 7) * 8;
    const bool dense = d.g.mode == CDETR_ROWS_DENSE;
    const int ky = dense ? 0 : tap / d.g.kw, kx = dense ? 0 : tap - (tap / d.g.kw) * d.g.kw;
    int p = kt_begin * 32 + kr;
    int pn = 0, py = 0, px = 0;
    if (!dense) {
        const int hw = d.g.Hc * d.g.Wc;
        pn = p / hw;
        const int rem = p - pn * hw;
        py = rem / d.g.Wc;
        px = rem - py * d.g.Wc;
    }
    // two register sets = two pixel tiles in flight; unconditional clamped loads + validity bits
    u32x4 ra[2][KP][A_SLOTS], rb[2][KP][B_SLOTS];
    unsigned rf[2] = {0, 0};
    int acol[A_SLOTS], bcol[B_SLOTS];
    const int nk = (kt_end - kt_begin + KP - 1) / KP;
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) acol[s] = min(i0 + 64 * s + c8, d.Nout - 8);
#pragma unroll
    for (int s = 0; s < B_SLOTS; ++s) bcol[s] = min(c0 + 64 * s + c8, d.Cin - 8);
    auto fetch = [&](u32x4 (&qa)[KP][A_SLOTS], u32x4 (&qb)[KP][B_SLOTS], unsigned& qf) __attribute__((always_inline)) {
        qf = 0;
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            const bool pv = p < p_end;
            const __bf16* yp = dY + (long)(pv ? p : d.P - 1) * d.ldy;
#pragma unroll
            for (int s = 0; s < A_SLOTS; ++s) qa[kp][s] = ld16(yp + acol[s]);
            long row = -1;
            if (pv) {
                if (dense) row = p;
                else {
                    const int iy = py * d.g.stride - d.g.pad + ky * d.g.dil;
                    const int ix = px * d.g.stride - d.g.pad + kx * d.g.dil;
                    if (iy >= 0 && iy < d.g.Ha && ix >= 0 && ix < d.g.Wa) row = ((long)pn * d.g.Ha + iy) * d.g.Wa + ix;
                }
            }
            qf |= ((pv ? 1u : 0u) | (row >= 0 ? 2u : 0u)) << (2 * kp);
            const __bf16* xp = X + (row >= 0 ? row : 0) * d.ldx;
#pragma unroll
            for (int s = 0; s < B_SLOTS; ++s) qb[kp][s] = ld16(xp + bcol[s]);
            p += 32;
            if (!dense) {
                px += 32;
                while (px >= d.g.Wc) { px -= d.g.Wc; ++py; }
                while (py >= d.g.Hc) { py -= d.g.Hc; ++pn; }
            }
        }
    };
    __bf16* const wbase = Zs + (c8 >> 5) * BLK + kr * 32 + (c8 & 31);
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    auto stash = [&](const u32x4 (&qa)[KP][A_SLOTS], const u32x4 (&qb)[KP][B_SLOTS], unsigned qf, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
#pragma unroll
            for (int s = 0; s < A_SLOTS; ++s)
                *reinterpret_cast<u32x4*>(wbase + buf * BUF + 2 * s * BLK + kp * 1024) = ((qf >> (2 * kp)) & 1u) ? qa[kp][s] : z4;
#pragma unroll
            for (int s = 0; s < B_SLOTS; ++s)
                *reinterpret_cast<u32x4*>(wbase + buf * BUF + PLANE_A + 2 * s * BLK + kp * 1024) = ((qf >> (2 * kp)) & 2u) ? qb[kp][s] : z4;
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int l16 = lane & 15;
    const int roff = ((g * 8 + (l16 >> 2)) * 32) + ((lane >> 4) & 1) * 16 + (l16 & 3) * 4;
    const __bf16* const abase = Zs + (wm * FM) * BLK + roff;
    const __bf16* const bbase = Zs + PLANE_A + (wn * FN) * BLK + roff;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const __bf16* as = abase + buf * BUF;
        const __bf16* bs = bbase + buf * BUF;
#pragma unroll
        for (int hp = 0; hp < 2 * KP; ++hp) {
            bf16x8 ah[FM], bh[FN];
#pragma unroll
            for (int a = 0; a < FM; ++a) ah[a] = lds_tr8(as + a * BLK + hp * 512, as + a * BLK + hp * 512 + 128);
#pragma unroll
            for (int b = 0; b < FN; ++b) bh[b] = lds_tr8(bs + b * BLK + hp * 512, bs + b * BLK + hp * 512 + 128);
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
                for (int b = 0; b < FN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
        }
    };

    fetch(ra[0], rb[0], rf[0]);
    fetch(ra[1], rb[1], rf[1]);
    stash(ra[0], rb[0], rf[0], 0);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        fetch(ra[0], rb[0], rf[0]);            // tile kt+2
        compute(0);
        stash(ra[1], rb[1], rf[1], 1);
        __syncthreads();
        fetch(ra[1], rb[1], rf[1]);            // tile kt+3
        compute(1);
        stash(ra[0], rb[0], rf[0], 0);
        __syncthreads();
    }
    if (kt < nk) compute(0);

    mfma_drain(acc);
#pragma unroll
    for (int a = 0; a < FM; ++a) {
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            const int c = c0 + wn * (BJ / 2) + b * 32 + i32;
            if (c >= d.Cin) continue;
            float ws[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) ws[r] = 1.f;
            if (d.w_scale) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ws[r] = d.w_scale[min(i0 + wm * (BI / 2) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, d.Nout - 1)];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wm * (BI / 2) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (i >= d.Nout) continue;
                float v = acc[a][b][r] * ws[r];
                float* dst = dW + (long)i * d.ldw + (long)tap * d.Cin + c;
                if (single) *dst += v;
                else atomicAdd(dst, v);
            }
        }
    }
}

// XCD-aware slice placement of the twin-fed weight gradients (1-D grid of nx * ny workgroups, ny % 8 == 0): workgroup b runs on XCD b % 8
// (private 4 MiB L2 each); XCD x is given the pixel slices x, x + 8, ... and runs every output tile of a slice back to back.  All tiles of a
// slice read the SAME pixel rows of dY16 / X16, so those rows cross the fabric once per slice instead of once per XCD that happens to hold
// one of the slice's tiles.  The weight gradients run at ~4 TB/s of counter traffic on 2.2x their algorithmic bytes (profiles/r3_traffic.json):
// this is the kernel family that IS fabric-bound.  Placement only affects speed.
__device__ __forceinline__ void wgrad_xcd_slice(int lid, int nx, int& bx, int& by) {
    const int xcd = lid & 7, j = lid >> 3;
    bx = j % nx;
    by = (j / nx) * 8 + xcd;
}

template <int BI, int BJ, int KP = 1>
__global__ __launch_bounds__(256) void wgrad_tr16_kernel(const cdetr_wgrad_desc d, const int tilesI, const int tilesJ, const int kt_per_slice,
                                                         const int nx_xcd) {
    if (nx_xcd > 0) {      // 1-D grid, XCD-aware slices (never a single slice: ny is a multiple of 8; empty trailing slices exit at once)
        int bx, by;
        wgrad_xcd_slice(blockIdx.x, nx_xcd, bx, by);
        wgrad_tr16_body<BI, BJ, KP>(d, tilesI, tilesJ, kt_per_slice, bx, by, 0, false);
        return;
    }
    wgrad_tr16_body<BI, BJ, KP>(d, tilesI, tilesJ, kt_per_slice, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y == 1);
}

template <int BI, int BJ, int TERMS = 3>
__global__ __launch_bounds__(256) void wgrad_tr_kernel(const cdetr_wgrad_desc d, const int tilesI, const int tilesJ,
                                                       const int kt_per_slice, float* __restrict__ dbias) {
    wgrad_tr_body<BI, BJ, TERMS>(d, tilesI, tilesJ, kt_per_slice, dbias, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y == 1);
}

// Grouped launch: up to WG_MAX independent weight-gradient problems (arbitrary pointers / shapes, same kernel variant) in ONE
// launch.  The problems' workgroups are concatenated along grid.x; a workgroup finds its problem with a short scalar scan of
// the (kernel-argument resident) table.  Problems of one group may accumulate into the same dW, so every partial sum is added
// atomically (`single` = false).
constexpr int WG_MAX = 16;
struct WgradGroupItem { cdetr_wgrad_desc d; int tilesI, tilesJ, per, nx, ny; int pad_; };
struct WgradGroupArgs { int n; int blk0[WG_MAX + 1]; WgradGroupItem it[WG_MAX]; };
static_assert(sizeof(WgradGroupArgs) <= 4000, "grouped launch arguments must fit the kernel-argument segment");

template <int BI, int BJ, int TERMS = 3>
__global__ __launch_bounds__(256) void wgrad_tr_group_kernel(const WgradGroupArgs g) {
    int p = 0;
    while (p + 1 < g.n && (int)blockIdx.x >= g.blk0[p + 1]) ++p;
    const WgradGroupItem& it = g.it[p];
    const int lb = blockIdx.x - g.blk0[p];
    const int per_z = it.nx * it.ny;
    const int z = lb / per_z, l = lb - z * per_z;
    if (z >= it.d.batch) return;                           // padding blocks (every item's count is rounded up to a multiple of 8)
    wgrad_tr_body<BI, BJ, TERMS>(it.d, it.tilesI, it.tilesJ, it.per, it.d.dbias, l % it.nx, l / it.nx, z, false);
}

template <int BI, int BJ, int KP = 1>
__global__ __launch_bounds__(256) void wgrad_tr16_group_kernel(const WgradGroupArgs g) {
    int p = 0;
    while (p + 1 < g.n && (int)blockIdx.x >= g.blk0[p + 1]) ++p;
    const WgradGroupItem& it = g.it[p];
    const int lb = blockIdx.x - g.blk0[p];
    const int per_z = it.nx * it.ny;
    const int z = lb / per_z, l = lb - z * per_z;
    if (z >= it.d.batch) return;
    int bx, by;
    if (it.pad_) wgrad_xcd_slice(l, it.nx, bx, by);    // (host: ny % 8 == 0, batch 1; blk0 is a multiple of 8: l % 8 == blockIdx.x % 8 == the XCD)
    else { bx = l % it.nx; by = l / it.nx; }
    wgrad_tr16_body<BI, BJ, KP>(it.d, it.tilesI, it.tilesJ, it.per, bx, by, z, false);
}

// ------------------------------------------------------------------------------------------------ direct small GEMMs
// Latency-optimised path for the ~450 small contractions per step (decoder M = B*Q = 600 rows, positional MLPs,
// per-head dq/dk of RCDA): one wave = one 16x16 output tile on v_mfma_f32_16x16x4_f32, operands loaded straight from
// global/L2 in the MFMA register layout -- no LDS, no barrier, every wave independent, so a 600x256x256 GEMM runs as
// 608 concurrent waves of 64 MFMAs instead of 40 workgroups stepping through 8 barrier-separated k-tiles.
template <int BL, int UB>   // UB = 16-wide k-chunks (per wave) whose loads are all in flight before the first MFMA
__device__ __forceinline__ void igemm_direct_body(const cdetr_gemm_desc& d, const int tilesM, const int tilesN, const int vecA,
                                                  const int vecB, const int bx, const int bz) {
    // one WORKGROUP = one 16x16 output tile; its 4 waves split K four ways (each wave: one short memory round trip),
    // partial accumulators are summed through LDS and wave 0 runs the fused epilogue.
    __shared__ float red[4][256];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = bx;
    const int tm = tile % tilesM, tn = tile / tilesM;
    const int i = lane & 15, g4 = lane >> 4;
    const int m = tm * 16 + i, n = tn * 16 + i;
    const int z = bz;
    const float* __restrict__ A = d.A + batch_off(z, d.batch_inner, d.sA, d.sA2);
    const float* __restrict__ B = d.B + batch_off(z, d.batch_inner, d.sB, d.sB2);
    float* __restrict__ C = d.C + batch_off(z, d.batch_inner, d.sC, d.sC2);
    __bf16* __restrict__ C16 = d.C16 ? reinterpret_cast<__bf16*>(d.C16) + batch_off(z, d.batch_inner, d.sC, d.sC2) : nullptr;   // bf16 twin of C
    __bf16* __restrict__ C16lo = (d.C16 && d.C16lo) ? reinterpret_cast<__bf16*>(d.C16lo) + batch_off(z, d.batch_inner, d.sC, d.sC2) : nullptr;   // its lo plane
    const int K = d.K;
    const int kchunks = (K + 15) >> 4;
    const int cpw = (kchunks + 3) >> 2;                 // chunks per wave
    const int kbeg = wid * cpw * 16, kend = min(K, kbeg + cpw * 16);
    const bool mv = m < d.M, nv = n < d.N;
    const float* arow = A + (long)(mv ? m : 0) * d.lda;
    const float* brow = B + (BL == 0 ? (long)(nv ? n : 0) * d.ldb : (long)(nv ? n : 0));
    const float s0 = (BL == 0 && d.w_scale && nv) ? d.w_scale[n] : 1.f;
    const float am = mv ? 1.f : 0.f;
    const float bm = nv ? s0 : 0.f;
    const float* wsc = d.w_scale;
    // epilogue operands of the output elements wave 0 will write (row tm*16 + g4*4 + r, column tn*16 + i): fetched up front so
    // their round trip overlaps the operand loads instead of following the reduction
    float rv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {1.f, 1.f, 1.f, 1.f};
    float bias = 0.f;
    if (wid == 0) {
        const int nc = min(n, d.N - 1);
        if (d.bias) bias = d.bias[nc];
        if (d.resid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) rv[r] = d.resid[(long)min(tm * 16 + g4 * 4 + r, d.M - 1) * d.ldr + nc];
        }
        if (d.gate) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] = d.gate[(long)min(tm * 16 + g4 * 4 + r, d.M - 1) * d.ldg + nc];
        }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // The operand forms (16-byte loads or four scalars) are decided ONCE, outside the batch loop (round 6): with `if (vecA)` / `if (vecB)`
    // inside it every chunk's load sat in its own basic block followed by `s_waitcnt vmcnt(0)` -- 2 UB serialised round trips per batch in a
    // kernel whose whole life is ~7 us (found in the ISA; the comment below states what the loop was always meant to do).
    auto run = [&](auto VA_, auto VB_) __attribute__((always_inline)) {
    constexpr bool VA = decltype(VA_)::value, VB = decltype(VB_)::value;
    for (int k0 = kbeg; k0 < kend; k0 += 16 * UB) {
        float a[UB][4], b[UB][4];
        float bw[BL == 0 ? 1 : UB][4];                               // b_layout 1: the per-k weight scales
        const float* wsp = wsc ? wsc : arow;
        // phase 1: every load of the batch, raw (branch-free, clamped addresses); phase 2: masks / scales + the MFMA chain.  The scheduling
        // barrier between them keeps the compiler from sinking each load to its first use (it did: two loads in flight, `s_waitcnt vmcnt(0)`
        // before every second MFMA -- the register-pressure heuristic of a 256-thread kernel)
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int k = k0 + u * 16 + g4 * 4;
            if constexpr (VA) {
                const float4 t = ld4(arow + (k < kend ? k : 0));
                a[u][0] = t.x; a[u][1] = t.y; a[u][2] = t.z; a[u][3] = t.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) a[u][s] = arow[k + s < kend ? k + s : 0];
            }
            if (BL == 0) {
                if constexpr (VB) {
                    const float4 t = ld4(brow + (k < kend ? k : 0));
                    b[u][0] = t.x; b[u][1] = t.y; b[u][2] = t.z; b[u][3] = t.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) b[u][s] = brow[k + s < kend ? k + s : 0];
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int kk = k + s < kend ? k + s : 0;
                    b[u][s] = brow[(long)kk * d.ldb];
                    bw[u][s] = wsp[wsc ? kk : 0];                    // (no branch around the load: absent scales read one valid float and are ignored)
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int k = k0 + u * 16 + g4 * 4;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool oka = VA ? (k < kend) : (k + s < kend);
                const bool okb = (BL == 0 && VB) ? (k < kend) : (k + s < kend);
                a[u][s] *= oka ? am : 0.f;
                if (BL == 0) b[u][s] *= okb ? bm : 0.f;
                else b[u][s] = (okb && nv) ? (wsc ? b[u][s] * bw[BL == 0 ? 0 : u][s] : b[u][s]) : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s], b[u][s], acc, 0, 0, 0);
    }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (vecA) { if (vecB || BL != 0) run(T_{}, T_{}); else run(T_{}, F_{}); }
    else { if (vecB || BL != 0) run(F_{}, T_{}); else run(F_{}, F_{}); }
    mfma_drain(acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wid][r * 64 + lane] = acc[r];
    __syncthreads();
    if (wid != 0) return;
    // C/D layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + r
    const int no = tn * 16 + i;
    if (no >= d.N) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int mo = tm * 16 + g4 * 4 + r;
        if (mo >= d.M) continue;
        const float sum = (red[0][r * 64 + lane] + red[1][r * 64 + lane]) + (red[2][r * 64 + lane] + red[3][r * 64 + lane]);
        float v = (sum + bias) * d.out_scale + rv[r];
        v = gv[r] > 0.f ? v : 0.f;
        if (d.relu) v = fmaxf(v, 0.f);
        C[(long)mo * d.ldc + no] = v;
        if (C16) C16[(long)mo * d.ldc + no] = (__bf16)v;
    }
}

template <int BL, int UB>
__global__ __launch_bounds__(256) void igemm_direct_kernel(const cdetr_gemm_desc d, const int tilesM, const int tilesN,
                                                           const int vecA, const int vecB) {
    igemm_direct_body<BL, UB>(d, tilesM, tilesN, vecA, vecB, blockIdx.x, blockIdx.z);
}

template <int BL, int UB>
__global__ __launch_bounds__(256) void igemm_direct_group_kernel(const GemmGroupArgs g) {
    int p = 0;
    while (p + 1 < g.n && (int)blockIdx.x >= g.blk0[p + 1]) ++p;
    const GemmGroupItem& it = g.it[p];
    const int lb = blockIdx.x - g.blk0[p];
    igemm_direct_body<BL, UB>(it.d, it.tilesM, it.tilesN, it.vecA, it.vecB, lb % it.nx, lb / it.nx);
}

// Few-row problems (K <= 256: one round trip per wave) AND 64x128-tile problems of one gemm_queue in ONE launch: the two classes are independent
// by contract, and a few-row group next to a tile group used to cost its own 6-10 us link in the launch chain (the encoder's key projections
// beside its query / value projections, the decoder's per-layer query gradients beside the memory gradient).  `pad_` of an item = its class;
// a few-row item uses the first 4 waves of the 512-thread block.
template <int TERMS>
__global__ __launch_bounds__(512) void igemm_mixed_group_kernel(const GemmGroupArgs g) {
    int p = 0;
    while (p + 1 < g.n && (int)blockIdx.x >= g.blk0[p + 1]) ++p;
    const GemmGroupItem& it = g.it[p];
    const int lb = blockIdx.x - g.blk0[p];
    if (it.pad_ == 0) {
        if (threadIdx.x >= 256) return;
        igemm_direct_body<0, 4>(it.d, it.tilesM, it.tilesN, it.vecA, it.vecB, lb % it.nx, lb / it.nx);
    } else {
        igemm_fast_body<2, 4, 1, 1, 0, 32, 3, TERMS>(it.d, it.tilesM, lb % it.nx, lb / it.nx);
    }
}

// dW[i][c] += scale[i] * sum_p dY[p][i] X[p][c] (+ dbias[i] += sum_p dY[p][i]) for short reductions (P <= 1024):
// one wave = one 16x16 output tile x one slice of the pixel range (grid.y slices); results are added atomically.
__device__ __forceinline__ void wgrad_direct_body(const cdetr_wgrad_desc& d, const int tilesI, const int tilesJ, const int p_per_slice,
                                                  float* __restrict__ dbias, const int bx, const int by, const int bz, const bool single) {
    constexpr int UB = 8;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = bx * 4 + wid;
    if (tile >= tilesI * tilesJ) return;
    const int ti = tile % tilesI, tj = tile / tilesI;
    const int i = lane & 15, g4 = lane >> 4;
    const int z = bz;
    const float* __restrict__ dY = d.dY + batch_off(z, d.batch_inner, d.sY, d.sY2);
    const float* __restrict__ X = d.X + batch_off(z, d.batch_inner, d.sX, d.sX2);
    float* __restrict__ dW = d.dW + batch_off(z, d.batch_inner, d.sW, d.sW2);
    const int ci = ti * 16 + i, cc = tj * 16 + i;
    const bool iv = ci < d.Nout, cv = cc < d.Cin;
    const float* ya = dY + (iv ? ci : 0);
    const float* xb = X + (cv ? cc : 0);
    const float fa = iv ? 1.f : 0.f, fb = cv ? 1.f : 0.f;
    const int pbeg = by * p_per_slice, pend = min(d.P, pbeg + p_per_slice);
    float ws[4] = {1.f, 1.f, 1.f, 1.f};        // output-row weight scales, fetched with the first operand batch
    if (d.w_scale) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ws[r] = d.w_scale[min(ti * 16 + g4 * 4 + r, d.Nout - 1)];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    for (int p0 = pbeg; p0 < pend; p0 += 16 * UB) {
        float a[UB][4], b[UB][4];
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int p = p0 + u * 16 + g4 * 4 + s;
                const bool pv = p < pend;
                const long pc = pv ? p : 0;
                a[u][s] = ya[pc * d.ldy] * (pv ? fa : 0.f);
                b[u][s] = xb[pc * d.ldx] * (pv ? fb : 0.f);
            }
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s], b[u][s], acc, 0, 0, 0);
                bsum += a[u][s];
            }
    }
    mfma_drain(acc);
    const int co = tj * 16 + i;
    if (co < d.Cin) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int io = ti * 16 + g4 * 4 + r;
            if (io >= d.Nout) continue;
            const float v = acc[r] * ws[r];
            float* dst = dW + (long)io * d.ldw + co;
            if (single) *dst += v;               // one owner per element within a launch
            else atomicAdd(dst, v);
        }
    }
    if (dbias != nullptr && tj == 0) {
        bsum += __shfl_xor(bsum, 16, 64);
        bsum = xhalf_sum(bsum);
        if (g4 == 0 && iv) atomicAdd(dbias + ci, bsum);
    }
}

__global__ __launch_bounds__(256) void wgrad_direct_kernel(const cdetr_wgrad_desc d, const int tilesI, const int tilesJ,
                                                           const int p_per_slice, float* __restrict__ dbias) {
    wgrad_direct_body(d, tilesI, tilesJ, p_per_slice, dbias, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y == 1);
}

__global__ __launch_bounds__(256) void wgrad_direct_group_kernel(const WgradGroupArgs g) {
    int p = 0;
    while (p + 1 < g.n && (int)blockIdx.x >= g.blk0[p + 1]) ++p;
    const WgradGroupItem& it = g.it[p];
    const int lb = blockIdx.x - g.blk0[p];
    const int bx = lb % it.nx, r = lb / it.nx;
    wgrad_direct_body(it.d, it.tilesI, it.tilesJ, it.per, it.d.dbias, bx, r % it.ny, r / it.ny, false);
}

// ------------------------------------------------------------------------------------------------ small kernels
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long ldx, int M, int N,
                                                     float* __restrict__ out, int rows_per_block) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    // eight rows per batch of loads, eight partial sums: a slice is 8 memory round trips instead of rows_per_block of them (the loop with one
    // accumulator is not unrolled by the compiler -- 27 us for 5000 x 256 on the main chain of the step, tools/step_listing.py)
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int r = r0;
    for (; r + 8 <= r1; r += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = X[(long)(r + i) * ldx + n];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] += v[i];
    }
    for (; r < r1; ++r) s[0] += X[(long)r * ldx + n];
    atomicAdd(out + n, ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7])));
}

__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ X, float* __restrict__ Y, __bf16* __restrict__ Y16,
                                                      __bf16* __restrict__ Y16lo, int Nimg, int H, int W, int C, int Ho, int Wo) {
    const long total = (long)Nimg * Ho * Wo * (C / 4);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % (C / 4));
        long t = idx / (C / 4);
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                const float4 v = ld4(X + (((long)n * H + iy) * W + ix) * C + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        const long o = (((long)n * Ho + oy) * Wo + ox) * C + c4 * 4;
        *reinterpret_cast<float4*>(Y + o) = m;
        if (Y16) {                            // split-bf16 planes of the pooled map: the A16 / A16lo operand of layer1's first convolutions
            uint2 h, l;
            split_bf16_pk(m.x, m.y, h.x, l.x);
            split_bf16_pk(m.z, m.w, h.y, l.y);
            *reinterpret_cast<uint2*>(Y16 + o) = h;
            if (Y16lo) *reinterpret_cast<uint2*>(Y16lo + o) = l;
        }
    }
}

template <int BM, int BN>
int launch_gemm(const cdetr_gemm_desc& d, hipStream_t st, int vecA, int vecB) {
    const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
    dim3 grid(tilesM * tilesN, 1, d.batch), block(256);
    if (d.b_layout == 0)
        hipLaunchKernelGGL((igemm_kernel<BM, BN, 0>), grid, block, 0, st, d, tilesM, vecA, vecB);
    else
        hipLaunchKernelGGL((igemm_kernel<BM, BN, 1>), grid, block, 0, st, d, tilesM, vecA, vecB);
    return cdetr_launch_status("cdetr_gemm");
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename F>
int raise_lds(F func, int bytes, const char* what) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(func), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) {
            cdetr_set_error("%s: hipFuncSetAttribute(%d): %s", what, bytes, hipGetErrorString(e));
            return CDETR_ERR_LAUNCH;
        }
    }
    return CDETR_OK;
}

// bf16 twin of A usable: given, 8-element granularity of every row / batch stride (16-byte loads of 8 k-values)
inline bool gemm_has_a_twin(const cdetr_gemm_desc& d) {
    static const int on = getenv("CDETR_GEMM_A16") ? atoi(getenv("CDETR_GEMM_A16")) : 1;
    return on && d.A16 && (d.lda & 7) == 0 && (d.sA & 7) == 0 && (d.sA2 & 7) == 0 && (reinterpret_cast<uintptr_t>(d.A16) & 15) == 0;
}

// Slices per output tile (1 = no split).  The tile kernels hide their load latency with OTHER workgroups of the same CU; a problem of two
// images has too few tiles for that (5000 x 256 outputs on 64x128 tiles: 158 workgroups on 256 CUs -- tools/m_scaling.py: twice the rows
// cost 2-30 % more time), so its reduction is cut into slices until the grid reaches ~2 workgroups of 8 waves per CU.
inline int gemm_ksplit(const cdetr_gemm_desc& d, int waves, long tiles, int nkt_all, long tile_bytes, int min_kt = 8) {
    if (!d.splitk_ws || d.batch != 1 || tiles > SPLITK_COUNTERS) return 1;
    const char* fs = cdetr_tune_env("CDETR_GEMM_SPLITK");           // A/B knob: 0 = never, n = n slices wherever legal
    const int force = fs ? atoi(fs) : -1;
    if (force == 0) return 1;
    int S = force > 0 ? force : (int)((256L * 16 / waves) / tiles);
    S = std::min(S, 4);
    S = std::min(S, nkt_all / min_kt);                       // >= 8 k-tiles per slice: below that the exchange costs more than it hides (tools/splitk_sweep.py)
    const long avail = d.splitk_ws_bytes - (long)SPLITK_COUNTERS * 4;
    while (S > 1 && tiles * S * tile_bytes > avail) --S;
    return std::max(S, 1);
}

template <int WM, int WN, int FM, int FN, int BKF, int PREC, int KG = 1>
int launch_gemm_fast_pd(const cdetr_gemm_desc& d, hipStream_t st) {
    constexpr int BM = 32 * FM * WM, BN = 32 * FN * WN;
    const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
    const int S = gemm_ksplit(d, WM * WN * KG, (long)tilesM * tilesN, (d.K / BKF) * d.taps, (long)BM * BN * 4, KG > 1 ? 4 : 8);
    dim3 grid(8 * ((tilesM + 7) / 8) * tilesN, S, d.batch), block(64 * WM * WN * KG);      // 8 XCD bands (see the kernel); grid.y = reduction slices
    int rc;
    if constexpr (PREC >= 2) {           // k-contiguous operands, split at staging (3: B pre-split); reduced-term forms by d.precision
        const int bytes = (2 * BM * (BKF + 4) + 2 * BN * (BKF + 4)) * 4;
        if (d.precision == 2) {          // bf16x2: B rounded to bf16
            if ((rc = raise_lds(igemm_fast_kernel<WM, WN, FM, FN, 0, BKF, PREC, 2, false, KG>, bytes, "cdetr_gemm"))) return rc;
            hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, FM, FN, 0, BKF, PREC, 2, false, KG>), grid, block, bytes, st, d, tilesM);
        } else if (d.precision == 3 && gemm_has_a_twin(d)) {   // plain bf16, A read from its bf16 twin
            if ((rc = raise_lds(igemm_fast_kernel<WM, WN, FM, FN, 0, BKF, PREC, 1, true, KG>, bytes, "cdetr_gemm"))) return rc;
            hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, FM, FN, 0, BKF, PREC, 1, true, KG>), grid, block, bytes, st, d, tilesM);
        } else if (d.precision == 3) {   // plain bf16
            if ((rc = raise_lds(igemm_fast_kernel<WM, WN, FM, FN, 0, BKF, PREC, 1, false, KG>, bytes, "cdetr_gemm"))) return rc;
            hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, FM, FN, 0, BKF, PREC, 1, false, KG>), grid, block, bytes, st, d, tilesM);
        } else {
            if ((rc = raise_lds(igemm_fast_kernel<WM, WN, FM, FN, 0, BKF, PREC, 3, false, KG>, bytes, "cdetr_gemm"))) return rc;
            hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, FM, FN, 0, BKF, PREC, 3, false, KG>), grid, block, bytes, st, d, tilesM);
        }
    } else if constexpr (KG != 1) {
        cdetr_set_error("cdetr_gemm: wave groups exist for the staging-split forms only");
        return CDETR_ERR_UNSUPPORTED;
    } else if (d.b_layout == 0) {        // fp32 MFMA
        static_assert(PREC == 0 || PREC == 1, "");
        const int bytes = (2 * BM * (BKF + 4) + 2 * BN * (BKF + 4)) * 4;
        if ((rc = raise_lds(igemm_fast_kernel<WM, WN, FM, FN, 0, BKF, PREC>, bytes, "cdetr_gemm"))) return rc;
        hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, FM, FN, 0, BKF, PREC>), grid, block, bytes, st, d, tilesM);
    } else if constexpr ((BKF * BN) % (4 * 64 * WM * WN) != 0) {
        cdetr_set_error("cdetr_gemm: this tile variant has no n-contiguous (b_layout 1) form");
        return CDETR_ERR_UNSUPPORTED;
    } else {                             // n-contiguous operand: fp32 MFMA or the per-fragment split
        const int bytes = (2 * BM * (BKF + 4) + 2 * BKF * (BN + 4)) * 4;
        if ((rc = raise_lds(igemm_fast_kernel<WM, WN, FM, FN, 1, BKF, PREC>, bytes, "cdetr_gemm"))) return rc;
        hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, FM, FN, 1, BKF, PREC>), grid, block, bytes, st, d, tilesM);
    }
    return cdetr_launch_status("cdetr_gemm");
}

// 64x64 tile, 64-deep k-tiles shared by TWO wave groups (8 waves): precision >= 1, k-contiguous weight
int launch_gemm_fast_kg2(const cdetr_gemm_desc& d, hipStream_t st) {
    if (d.B_split && d.batch == 1) return launch_gemm_fast_pd<2, 2, 1, 1, 64, 3, 2>(d, st);
    return launch_gemm_fast_pd<2, 2, 1, 1, 64, 2, 2>(d, st);
}

template <int WM, int WN, int FM, int FN, int BKF>
int launch_gemm_fast(const cdetr_gemm_desc& d, hipStream_t st) {
    if (d.precision >= 1) {
        // bf16x3 (precision 1; 2 / 3 = the reduced-term forms, launch_gemm_fast_pd): k-contiguous operands (b_layout 0) are split once
        // while they are staged into LDS -- the weight operand arrives pre-split from the step's weight images when there is one
        // (B_split: a plain copy) --; the n-contiguous operand keeps the per-fragment split (its staging transpose costs more than it
        // saves, profiles/r1_gemm_sweep.txt).
        if (d.B_split && d.b_layout == 0 && d.batch == 1) return launch_gemm_fast_pd<WM, WN, FM, FN, BKF, 3>(d, st);
        if (d.b_layout == 0) return launch_gemm_fast_pd<WM, WN, FM, FN, BKF, 2>(d, st);
        return launch_gemm_fast_pd<WM, WN, FM, FN, BKF, 1>(d, st);
    }
    return launch_gemm_fast_pd<WM, WN, FM, FN, BKF, 0>(d, st);
}

}  // namespace

namespace {
long gemm_blocks(const cdetr_gemm_desc& d, int bm, int bn) { return (long)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn) * d.batch; }
int gemm_vec_a(const cdetr_gemm_desc& d) {
    return ((d.K & 3) == 0 && (d.lda & 3) == 0 && (d.sA & 3) == 0 && (d.sA2 & 3) == 0 && aligned16(d.A)) ? 1 : 0;
}
int gemm_vec_b(const cdetr_gemm_desc& d) {
    if (d.b_layout == 0)
        return ((((long)d.K * d.taps) & 3) == 0 && (d.ldb & 3) == 0 && (d.sB & 3) == 0 && (d.sB2 & 3) == 0 && aligned16(d.B)) ? 1 : 0;
    return ((d.N & 3) == 0 && (d.ldb & 3) == 0 && (d.sB & 3) == 0 && (d.sB2 & 3) == 0 && aligned16(d.B)) ? 1 : 0;
}
// the kernel classes of the default dispatch (shared by cdetr_gemm and cdetr_gemm_group)
bool gemm_is_direct(const cdetr_gemm_desc& d, int vecA, int vecB) {
    static const long direct_max = getenv("CDETR_DIRECT_MAX_BLOCKS") ? atol(getenv("CDETR_DIRECT_MAX_BLOCKS")) : 48;      // A/B knob
    return d.g.mode == CDETR_ROWS_DENSE && (gemm_blocks(d, 64, 64) <= direct_max || !(vecA && vecB && (d.K % 32) == 0)) && gemm_blocks(d, 64, 64) < 192;
}
bool gemm_is_fewrow_split(const cdetr_gemm_desc& d, int vecA, int vecB) {
    const char* e = cdetr_tune_env("CDETR_GEMM_FEWROW_SPLIT");      // A/B knob (tests)
    return !(e && atoi(e) == 0) && d.splitk_ws && d.batch == 1 && vecA && vecB && d.precision >= 1 && d.b_layout == 0 && (d.K % 64) == 0 && (long)d.K * d.taps >= 1024 &&
           d.M >= 256;
}
// bf16x3 with a long reduction is bound by L2->CU operand delivery (~8 TB/s measured, tools/split_sweep.py): a
// 128x128 tile shared by 16 waves of 32x32 halves that traffic at the same per-wave structure and occupancy.
bool gemm_is_f44(const cdetr_gemm_desc& d) {
    return d.precision >= 1 && d.b_layout == 0 && (long)d.K * d.taps >= 1024 && (d.N % 128) == 0 && gemm_blocks(d, 128, 128) >= 150;
}
// small grids (< 2 workgroups of 4 waves per CU): a 64x128 tile shared by 8 waves keeps the wave count and halves the
// A-operand traffic (tools/split_sweep.py: 6-11 % over 64x64 BK64 on the N = 256 encoder linears)
bool gemm_is_f24(const cdetr_gemm_desc& d) {
    return d.precision >= 1 && d.b_layout == 0 && (d.N % 128) == 0 && gemm_blocks(d, 64, 64) < 512;
}
}  // namespace
bool cdetr_gemm_dl_eligible(const cdetr_gemm_desc& d);      // igemm_dl.hip
namespace {
int check_gemm_desc(const cdetr_gemm_desc& d) {
    CDETR_CHECK_ARG(d.M >= 0 && d.N > 0 && d.K > 0 && d.taps > 0 && d.batch > 0, "cdetr_gemm: bad sizes M=%d N=%d K=%d taps=%d batch=%d",
                    d.M, d.N, d.K, d.taps, d.batch);
    CDETR_CHECK_ARG((d.A || d.A16) && d.B && (d.C || d.C16 || ((d.flags & CDETR_GEMM_C_GROUPS) && d.C16lo)), "cdetr_gemm: null A / B / output");
    // C == nullptr: the fp32 output is not wanted (an inner gradient of a bottleneck that only the next bf16 contraction reads): only the
    // direct-to-LDS kernel writes the bf16 twin alone -- the problem must be eligible for it (cdetr_gemm_dl_eligible)
    // (likewise A == NULL: the operand exists as its twin A16 only)
    CDETR_CHECK_ARG((d.C && d.A && !(d.flags & (CDETR_GEMM_A_GROUPS | CDETR_GEMM_C_GROUPS | CDETR_GEMM_RESID_GROUPS | CDETR_GEMM_GATE16_ONLY))) || cdetr_gemm_dl_eligible(d), "cdetr_gemm: A == NULL / C == NULL need A16 / C16 and direct-to-LDS-eligible operands (A16, B16 / B_split, K %% 64 == 0)");
    CDETR_CHECK_ARG(d.b_layout == 0 || d.b_layout == 1, "cdetr_gemm: b_layout %d", d.b_layout);
    if (d.g.mode == CDETR_ROWS_DENSE) {
        CDETR_CHECK_ARG(d.taps == 1, "cdetr_gemm: dense rows need taps == 1");
    } else {
        CDETR_CHECK_ARG(d.g.mode == CDETR_ROWS_CONV_FWD || d.g.mode == CDETR_ROWS_CONV_DGRAD, "cdetr_gemm: mode %d", d.g.mode);
        CDETR_CHECK_ARG(d.taps == d.g.kh * d.g.kw && d.g.stride >= 1 && d.g.dil >= 1, "cdetr_gemm: conv geometry mismatch");
        CDETR_CHECK_ARG(d.g.Hc > 0 && d.g.Wc > 0 && d.M % (d.g.Hc * d.g.Wc) == 0, "cdetr_gemm: M is not images*Hc*Wc");
    }
    if (d.b_layout == 1 && d.taps > 1) CDETR_CHECK_ARG(d.K % BK == 0, "cdetr_gemm: dgrad with taps needs K %% 16 == 0");
    if (!gemm_vec_a(d)) CDETR_CHECK_ARG(d.g.mode == CDETR_ROWS_DENSE, "cdetr_gemm: unaligned A only supported for dense rows");
    return CDETR_OK;
}
}  // namespace

// igemm_dl.hip: the direct-to-LDS tile kernel (operands pre-split in HBM)
bool cdetr_gemm_dl_eligible(const cdetr_gemm_desc& d);
int cdetr_gemm_dl_launch(const cdetr_gemm_desc& d, int tile, int stages, hipStream_t st);

namespace {
// Tile of the direct-to-LDS kernel for a problem: 0 = 128x128, 1 = 128x64, 2 = 64x128, 3 = 64x64 (rows x channels), or -1 = leave the
// problem to the register-staged kernels.  Measured on the backbone's shapes at two 800x800 images, cold operands (tools/dl_sweep.py,
// profiles/r3_dl_sweep.txt):
//   * plain bf16 (the data gradients: A = the bf16 twin of dY, B = the plain-bf16 weight image): the direct-to-LDS kernel wins on
//     every shape, 1.1-1.7x (sum 494 -> 386 us over the sweep); 64x64 tiles / 3-deep ring is the best or within 5 % of it everywhere
//     (3 workgroups per CU de-synchronise the store bursts of the epilogues), 128-row tiles only from K * taps >= 2048 on;
//   * split-bf16 x3 (the forward): the register-staged kernel reads fp32 rows = full 128-byte lines per k-tile, the planes
//     (64 B of hi + 64 B of lo per row and k-tile) are half lines -- the direct-to-LDS kernel wins where the epilogue dominates
//     (K <= 256: 1.1-1.2x) and loses on the long reductions (3x3, K >= 1024: 0.8-0.95x), -7 % over the sweep AFTER charging every
//     producer for the lo plane: not the default (the producers only write lo planes under ops.SPLIT_FWD).
// CDETR_GEMM_DL (read once): 0 = never, 1 = this rule (default), 100 + 10 * tile + stages = force one configuration wherever legal.
int gemm_dl_choice(const cdetr_gemm_desc& d, int& stages) {
    static const int mode = getenv("CDETR_GEMM_DL") ? atoi(getenv("CDETR_GEMM_DL")) : 1;
    if (mode == 0 || !cdetr_gemm_dl_eligible(d)) return -1;
    if (mode >= 100) {
        stages = mode % 10;
        return (mode / 10) % 10;
    }
    auto blocks = [&](int bm, int bn) { return (long)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn); };
    stages = 3;
    // split reduction (stages code 200 + 10 * slices + ring depth, igemm_dl.hip) -- CDETR_DL_SPLITK: 0 = never (default), 1 = this rule.
    // In the step the rule is neutral (three same-lease pairs, profiles/r5_dl_splitk.txt: the backbone's backward is throughput-bound on
    // two queues -- a shorter data-gradient launch gives its CUs to the weight gradients beside it, not to the chain): off.
    // Cold-operand sweep of the data gradients (profiles/r5_dl_splitk.txt): three slices on 64x128 tiles win on the 3x3 convolutions of
    // 50x50 maps (36-72 k-tiles per tile; 256 -> 256: 22.9 -> 19.5 us, 512 -> 512: 58.0 -> 45.5 us), the exchange (write-through partials,
    // arrival count, late epilogue operands: ~6 us of round trips) costs more than it hides on every 1x1 shape (K <= 2048: 8-32 k-tiles)
    auto split_3x3 = [&]() {
        static const int sm = getenv("CDETR_DL_SPLITK") ? atoi(getenv("CDETR_DL_SPLITK")) : 0;
        const long wgs = blocks(64, 128);
        const int nkt_all = (int)((long)d.K / (d.precision == 1 ? 32 : 64) * d.taps);
        return sm == 1 && d.taps == 9 && d.N % 128 == 0 && d.splitk_ws && wgs <= 384 && nkt_all >= 24 &&
               (long)SPLITK_COUNTERS * 4 + wgs * 3 * 64 * 128 * 4 <= d.splitk_ws_bytes;
    };
    if (!d.C) return blocks(64, 64) >= 512 && (long)d.K * d.taps >= 2048 && d.N % 128 == 0 ? 0 : 3;   // no fp32 output: only this kernel can run it
    if (blocks(64, 64) < 192) return -1;                                      // few tiles: the split-reduction forms of the register-staged kernels
    if (d.precision == 3) {
        if (split_3x3()) {
            stages = 233;
            return 2;
        }
        return ((long)d.K * d.taps >= 2048 && d.N % 128 == 0 && blocks(128, 128) >= 128) ? 0 : 3;
    }
    // split-bf16 x3 with planes given: where the sweep has it ahead -- the expanding 1x1 convolutions (K = planes <= 512, N = 4 K), whose
    // epilogue dominates; 64x128 tiles for the widest one
    if ((long)d.K * d.taps <= 512 && d.N >= 2 * d.K) return d.N >= 2048 ? 2 : 3;
    return -1;
}
}  // namespace

extern "C" int cdetr_gemm_dl(const cdetr_gemm_desc* dp, int32_t tile, int32_t stages, void* stream) {
    CDETR_CHECK_ARG(dp != nullptr, "cdetr_gemm_dl: null descriptor");
    if (int rcv = check_gemm_desc(*dp)) return rcv;
    if (dp->M == 0) return CDETR_OK;
    if (!cdetr_gemm_dl_eligible(*dp)) {
        cdetr_set_error("cdetr_gemm_dl: needs b_layout 0, batch 1, A16 (+ A16lo for precision 1), B_split, K %% %d == 0, 16-byte aligned operands",
                        dp->precision == 1 ? 32 : 64);
        return CDETR_ERR_UNSUPPORTED;
    }
    return cdetr_gemm_dl_launch(*dp, tile, stages, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int cdetr_gemm(const cdetr_gemm_desc* dp, void* stream) {
    CDETR_CHECK_ARG(dp != nullptr, "cdetr_gemm: null descriptor");
    cdetr_gemm_desc d = *dp;
    if (int rcv = check_gemm_desc(d)) return rcv;
    if (d.M == 0) return CDETR_OK;
    const int vecA = gemm_vec_a(d), vecB = gemm_vec_b(d);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // tile choice: the largest tile that still yields >= 1.5 waves of workgroups on 256 CUs
    auto blocks = [&](int bm, int bn) { return (long)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn) * d.batch; };
    // tuning knob (tools/gemm_sweep.py): CDETR_GEMM_VARIANT = 1..4 forces a fast tile, 5 the direct kernel, 6 the generic one
    const char* force_s = cdetr_tune_env("CDETR_GEMM_VARIANT");
    const int force = force_s ? atoi(force_s) : 0;
    const bool fast_ok = vecA && vecB && (d.K % 32) == 0;
    if (!d.C || !d.A || (d.flags & (CDETR_GEMM_A_GROUPS | CDETR_GEMM_C_GROUPS | CDETR_GEMM_RESID_GROUPS | CDETR_GEMM_GATE16_ONLY))) { // grouped operands, twin-only output / operand: the direct-to-LDS kernel is the only one that handles it (check_gemm_desc made sure it can)
        int stages = 3;
        const int tile = gemm_dl_choice(d, stages);
        return cdetr_gemm_dl_launch(d, tile >= 0 ? tile : 3, stages, st);
    }
    // CDETR_GEMM_VARIANT (tests / tools/gemm_sweep.py): 3 = 64x64 BK64, 4 = 64x64, 9 = 64x128 (8 waves), 10 = 128x128 (16 waves),
    // 13 = 96x128 (12 waves) -- the tiles of the default dispatch --, 5 = the few-row kernel, 6 = the generic kernel
    if (force == 9 && fast_ok) return launch_gemm_fast<2, 4, 1, 1, 32>(d, st);
    if (force == 10 && fast_ok) return launch_gemm_fast<4, 4, 1, 1, 32>(d, st);
    if (force == 13 && fast_ok && d.b_layout == 0) return launch_gemm_fast<3, 4, 1, 1, 32>(d, st);
    if (force == 3 && fast_ok && (d.K % 64) == 0) return launch_gemm_fast<2, 2, 1, 1, 64>(d, st);
    if (force == 4 && fast_ok) return launch_gemm_fast<2, 2, 1, 1, 32>(d, st);
    if (force == 14 && fast_ok && (d.K % 64) == 0 && d.precision >= 1 && d.b_layout == 0) return launch_gemm_fast_kg2(d, st);
    if (force == 16 && fast_ok && d.precision >= 1 && d.b_layout == 0) return launch_gemm_fast<2, 4, 2, 1, 32>(d, st);     // 128x128, 8 waves of 64x32
    if (force == 6) {
        if (d.N > 64 && d.M > 64) return launch_gemm<128, 128>(d, st, vecA, vecB);
        return launch_gemm<64, 64>(d, st, vecA, vecB);
    }
    if (!force) {
        int stages = 3;
        const int tile = gemm_dl_choice(d, stages);
        if (tile >= 0) return cdetr_gemm_dl_launch(d, tile, stages, st);
    }
    // few rows, long reduction (decoder FFN 1024 -> 256 on 600 rows): a 64x64 tile on two wave groups with the reduction cut across
    // 3-4 workgroups beats the one-wave-per-16x16-tile kernel (tools/splitk_sweep.py: 16.3 -> 10.6-12.6 us)
    // (kept at bf16x3 in the backward too: the few-row kernel they replace computes exact fp32 products whatever d.precision says, and
    // the decoder's gradient chain consists of nothing but few-row GEMMs -- tests/test_model_gpu.py trajectory test)
    if (!force && gemm_is_direct(d, vecA, vecB) && gemm_is_fewrow_split(d, vecA, vecB)) {
        if (d.precision >= 2) { d.precision = 1; d.A16 = nullptr; }
        return launch_gemm_fast_kg2(d, st);
    }
    if ((force == 5 && d.g.mode == CDETR_ROWS_DENSE) || gemm_is_direct(d, vecA, vecB)) {   // latency-bound: one wave per 16x16 tile
        const int tilesM = (d.M + 15) / 16, tilesN = (d.N + 15) / 16;
        dim3 grid(tilesM * tilesN, 1, d.batch);          // one workgroup (4 waves, K split 4 ways) per 16x16 tile
        const int cpw = (((d.K + 15) >> 4) + 3) >> 2;    // 16-wide k-chunks per wave
        auto go = [&](auto k0, auto k1) {
            if (d.b_layout == 0) hipLaunchKernelGGL(k0, grid, dim3(256), 0, st, d, tilesM, tilesN, vecA, vecB);
            else hipLaunchKernelGGL(k1, grid, dim3(256), 0, st, d, tilesM, tilesN, vecA, vecB);
        };
        if (cpw <= 4) go(igemm_direct_kernel<0, 4>, igemm_direct_kernel<1, 4>);
        else if (cpw <= 8) go(igemm_direct_kernel<0, 8>, igemm_direct_kernel<1, 8>);
        else go(igemm_direct_kernel<0, 16>, igemm_direct_kernel<1, 16>);
        return cdetr_launch_status("cdetr_gemm");
    }
    if (vecA && vecB && (d.K % 32) == 0) {   // fast path: tap-uniform k-tiles, no integer division in the loop
        // cost model: CUs run ceil(blocks / 256) "rounds" of a workgroup whose matrix-pipe time is ~ BM*BN; take the
        // cheapest tile, preferring the finer one on ties (better tail / latency hiding)
        // measured on MI355X (tools/gemm_sweep.py, profiles/r1_gemm_sweep.txt): 64x64 tiles at 4 workgroups/CU beat the
        // 128-wide tiles on every shape of this model (latency hiding by occupancy matters more than operand reuse at
        // the fp32-MFMA rate); BK = 64 only pays when the grid is too small to give every CU two workgroups.
        // bf16x3 with a long reduction is bound by L2->CU operand delivery (~8 TB/s measured, tools/split_sweep.py): a
        // 128x128 tile shared by 16 waves of 32x32 halves that traffic at the same per-wave structure and occupancy.
        if (gemm_is_f44(d)) {
            // 128x128 tiles that leave > 1/8 of the CUs idle in a single round: 96x128 tiles (12 waves) when those still fit one round --
            // 5000 x 512 outputs are 160 workgroups of 128 rows but 212 of 96 (each 3/4 of the work)
            // ... except the longest reductions (3x3 512 -> 512 at 50x50: 144 k-tiles): 128x128 on 8 waves of 64x32 with the reduction cut
            // three ways (480 workgroups) -- 112 -> 101 us bf16x3, 70 -> 65 us bf16 (tools/splitk_sweep.py SWEEP_BIG=1; larger per-wave tiles
            // lose on every other shape of the step, 64x64 per wave on all of them)
            if (d.splitk_ws && d.batch == 1 && blocks(128, 128) < 224 && (long)d.K * d.taps >= 4096) return launch_gemm_fast<2, 4, 2, 1, 32>(d, st);
            if (blocks(128, 128) < 224 && blocks(96, 128) <= 256) return launch_gemm_fast<3, 4, 1, 1, 32>(d, st);
            return launch_gemm_fast<4, 4, 1, 1, 32>(d, st);
        }
        // small grids (< 2 workgroups of 4 waves per CU): a 64x128 tile shared by 8 waves keeps the wave count and halves the
        // A-operand traffic (tools/split_sweep.py: 6-11 % over 64x64 BK64 on the N = 256 encoder linears)
        if (gemm_is_f24(d)) return launch_gemm_fast<2, 4, 1, 1, 32>(d, st);
        if ((d.K % 64) == 0 && blocks(64, 64) < 512) return launch_gemm_fast<2, 2, 1, 1, 64>(d, st);
        return launch_gemm_fast<2, 2, 1, 1, 32>(d, st);
    }
    if (d.N > 64 && d.M > 64 && blocks(128, 128) >= 384) return launch_gemm<128, 128>(d, st, vecA, vecB);
    if (d.M > 64 && blocks(128, 64) >= 384) return launch_gemm<128, 64>(d, st, vecA, vecB);
    return launch_gemm<64, 64>(d, st, vecA, vecB);
}

// n INDEPENDENT GEMMs submitted together.  Problems of the few-row class (igemm_direct_kernel, k-contiguous weight) and of the
// 64x128 / 8-wave class with a pre-split weight image run as grouped launches (<= GG_MAX problems per kernel); everything else goes
// through cdetr_gemm one by one.  The ~150 few-row GEMMs of a step take ~6 us each for well under a microsecond of work, and the
// N = 256 encoder projections launch 158 workgroups on 256 CUs: submitted together they share one launch and fill the chip.
extern "C" int cdetr_gemm_group(const cdetr_gemm_desc* descs, int32_t n, void* stream) {
    CDETR_CHECK_ARG(n >= 0 && (descs != nullptr || n == 0), "cdetr_gemm_group: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static const bool grouping = !(getenv("CDETR_GEMM_GROUP") && atoi(getenv("CDETR_GEMM_GROUP")) == 0) && !getenv("CDETR_GEMM_VARIANT");
    std::vector<int> cls[5];          // direct UB 4 / 8 / 16 (k-contiguous weight), fast 64x128 with a pre-split weight (bf16x3 | plain bf16)
    for (int i = 0; i < n; ++i) {
        const cdetr_gemm_desc& d = descs[i];
        if (int rcv = check_gemm_desc(d)) return rcv;
        if (d.M == 0) continue;
        const int vecA = gemm_vec_a(d), vecB = gemm_vec_b(d);
        int c = -1;
        if (grouping && gemm_is_direct(d, vecA, vecB)) {
            if (d.b_layout == 0) {
                const int cpw = (((d.K + 15) >> 4) + 3) >> 2;
                c = cpw <= 4 ? 0 : (cpw <= 8 ? 1 : 2);
            }
        } else if (grouping && vecA && vecB && (d.K % 32) == 0 && !gemm_is_f44(d) && gemm_is_f24(d) && d.B_split && d.batch == 1) {
            c = d.precision == 3 ? 4 : 3;
        }
        if (c >= 0) cls[c].push_back(i);
        else if (int rc1 = cdetr_gemm(&d, stream)) return rc1;
    }
    // one launch for a few-row (K <= 256) class next to ONE tile class, when everything fits a single argument table
    static const int mixed = getenv("CDETR_GEMM_MIXED") ? atoi(getenv("CDETR_GEMM_MIXED")) : 1;
    const int tc = !cls[3].empty() ? 3 : 4;
    if (mixed && !cls[0].empty() && (cls[3].empty() != cls[4].empty()) && cls[0].size() + cls[tc].size() <= (size_t)GG_MAX) {
        GemmGroupArgs g;
        int m = 0;
        g.blk0[0] = 0;
        for (int pass = 0; pass < 2; ++pass) {
            for (int idx : (pass == 0 ? cls[tc] : cls[0])) {
                GemmGroupItem& it = g.it[m];
                it.d = descs[idx];
                it.vecA = gemm_vec_a(it.d); it.vecB = gemm_vec_b(it.d); it.pad_ = pass == 0 ? 1 : 0;
                if (pass == 1) {
                    it.tilesM = (it.d.M + 15) / 16; it.tilesN = (it.d.N + 15) / 16;
                    it.nx = it.tilesM * it.tilesN;
                } else {
                    it.tilesM = (it.d.M + 63) / 64; it.tilesN = (it.d.N + 127) / 128;
                    it.nx = 8 * ((it.tilesM + 7) / 8) * it.tilesN;
                }
                g.blk0[m + 1] = g.blk0[m] + it.nx * it.d.batch;
                ++m;
            }
        }
        g.n = m;
        const int bytes = (2 * 64 * 36 + 2 * 128 * 36) * 4;
        if (tc == 3) {
            if (int rcl = raise_lds(igemm_mixed_group_kernel<3>, bytes, "cdetr_gemm_group")) return rcl;
            hipLaunchKernelGGL(igemm_mixed_group_kernel<3>, dim3(g.blk0[m]), dim3(512), bytes, st, g);
        } else {
            if (int rcl = raise_lds(igemm_mixed_group_kernel<1>, bytes, "cdetr_gemm_group")) return rcl;
            hipLaunchKernelGGL(igemm_mixed_group_kernel<1>, dim3(g.blk0[m]), dim3(512), bytes, st, g);
        }
        if (int rcl = cdetr_launch_status("cdetr_gemm_group")) return rcl;
        cls[0].clear();
        cls[tc].clear();
    }
    for (int c = 0; c < 5; ++c) {
        for (size_t c0 = 0; c0 < cls[c].size(); c0 += GG_MAX) {
            const int m = (int)std::min<size_t>(GG_MAX, cls[c].size() - c0);
            if (m == 1) { if (int rc1 = cdetr_gemm(&descs[cls[c][c0]], stream)) return rc1; continue; }
            GemmGroupArgs g;
            g.n = m; g.blk0[0] = 0;
            for (int k = 0; k < m; ++k) {
                GemmGroupItem& it = g.it[k];
                it.d = descs[cls[c][c0 + k]];
                it.vecA = gemm_vec_a(it.d); it.vecB = gemm_vec_b(it.d); it.pad_ = 0;
                if (c < 3) {
                    it.tilesM = (it.d.M + 15) / 16; it.tilesN = (it.d.N + 15) / 16;
                    it.nx = it.tilesM * it.tilesN;
                } else {
                    it.tilesM = (it.d.M + 63) / 64; it.tilesN = (it.d.N + 127) / 128;
                    it.nx = 8 * ((it.tilesM + 7) / 8) * it.tilesN;          // 8 XCD bands (see igemm_fast_body)
                }
                g.blk0[k + 1] = g.blk0[k] + it.nx * it.d.batch;
            }
            if (c == 0) hipLaunchKernelGGL((igemm_direct_group_kernel<0, 4>), dim3(g.blk0[m]), dim3(256), 0, st, g);
            else if (c == 1) hipLaunchKernelGGL((igemm_direct_group_kernel<0, 8>), dim3(g.blk0[m]), dim3(256), 0, st, g);
            else if (c == 2) hipLaunchKernelGGL((igemm_direct_group_kernel<0, 16>), dim3(g.blk0[m]), dim3(256), 0, st, g);
            else if (c == 3) {
                const int bytes = (2 * 64 * 36 + 2 * 128 * 36) * 4;
                if (int rcl = raise_lds(igemm_fast_group_kernel<2, 4, 1, 1, 0, 32, 3>, bytes, "cdetr_gemm_group")) return rcl;
                hipLaunchKernelGGL((igemm_fast_group_kernel<2, 4, 1, 1, 0, 32, 3>), dim3(g.blk0[m]), dim3(512), bytes, st, g);
            } else {                   // the backward's plain-bf16 products (data gradients of sibling projections)
                const int bytes = (2 * 64 * 36 + 2 * 128 * 36) * 4;
                if (int rcl = raise_lds(igemm_fast_group_kernel<2, 4, 1, 1, 0, 32, 3, 1>, bytes, "cdetr_gemm_group")) return rcl;
                hipLaunchKernelGGL((igemm_fast_group_kernel<2, 4, 1, 1, 0, 32, 3, 1>), dim3(g.blk0[m]), dim3(512), bytes, st, g);
            }
            if (int rcl = cdetr_launch_status("cdetr_gemm_group")) return rcl;
        }
    }
    return CDETR_OK;
}

namespace {
bool wgrad_has_twins(const cdetr_wgrad_desc& d);
bool wgrad_is_direct(const cdetr_wgrad_desc& d);
bool wgrad_is_fast(const cdetr_wgrad_desc& d);
int check_wgrad_desc(const cdetr_wgrad_desc& d) {
    CDETR_CHECK_ARG(d.P >= 0 && d.Nout > 0 && d.Cin > 0 && d.taps > 0 && d.batch > 0, "cdetr_wgrad: bad sizes");
    CDETR_CHECK_ARG((d.dY || d.dY16) && (d.X || d.X16) && d.dW, "cdetr_wgrad: null pointer");
    // dY == NULL: the gradient exists as its bf16 twin only (cdetr_gemm_desc.C == NULL upstream): the twin-fed tile kernel must be the one that runs
    // (X == NULL likewise: the activation exists as interleaved groups + its twin, there is no fp32 tensor to read)
    CDETR_CHECK_ARG((d.dY && d.X) || (wgrad_has_twins(d) && wgrad_is_fast(d) && !wgrad_is_direct(d)),
                    "cdetr_wgrad: dY == NULL / X == NULL need dY16 + X16, plain-bf16 products and a problem of the tile-kernel class");
    CDETR_CHECK_ARG(aligned16(d.dY) && aligned16(d.X) && (d.sY & 3) == 0 && (d.sX & 3) == 0 && (d.sY2 & 3) == 0 && (d.sX2 & 3) == 0, "cdetr_wgrad: dY/X must be 16-byte aligned");
    if (d.g.mode == CDETR_ROWS_DENSE) {
        CDETR_CHECK_ARG(d.taps == 1, "cdetr_wgrad: dense rows need taps == 1");
    } else {
        CDETR_CHECK_ARG(d.g.mode == CDETR_ROWS_CONV_FWD && d.taps == d.g.kh * d.g.kw, "cdetr_wgrad: geometry");
        CDETR_CHECK_ARG(d.P % (d.g.Hc * d.g.Wc) == 0, "cdetr_wgrad: P is not images*Hc*Wc");
    }
    return CDETR_OK;
}
// few-pixel problems (decoder, positional MLPs): pixel slices of <= 128 pixels (8 chunks: one round trip) per wave
void direct_wgrad_plan(const cdetr_wgrad_desc& d, int& tilesI, int& tilesJ, int& per, int& slices) {
    tilesI = (d.Nout + 15) / 16; tilesJ = (d.Cin + 15) / 16;
    slices = (d.P + 127) / 128;
    if (slices > 8) slices = 8;
    per = (d.P + slices - 1) / slices;
    per = ((per + 15) / 16) * 16;
    slices = (d.P + per - 1) / per;
}
// few-pixel problems (positional MLPs: P = B*(h+w) rows) take the one-wave-per-16x16-tile kernel; from 256 pixels on (the decoder's B*Q = 600
// rows) the transpose-read tile kernel is faster -- measured on the step: boundary 1024 / 512 / 256 / 64 pixels -> 11.71 / 11.63 / 11.60 /
// 11.60 ms (CDETR_WGRAD_DIRECT_MAXP moves it for A/B runs)
bool wgrad_is_direct(const cdetr_wgrad_desc& d) {
    static const int maxp = getenv("CDETR_WGRAD_DIRECT_MAXP") ? atoi(getenv("CDETR_WGRAD_DIRECT_MAXP")) : 256;
    return d.g.mode == CDETR_ROWS_DENSE && (d.P <= maxp || !((d.ldy & 3) == 0 && (d.ldx & 3) == 0 && (d.Nout & 3) == 0 && (d.Cin & 3) == 0)) && d.P <= 1024;
}
// bf16 twins usable (wgrad_tr16_kernel): plain-bf16 products requested, both twins given, 8-element granularity everywhere
bool wgrad_has_twins(const cdetr_wgrad_desc& d) {
    static const int on = getenv("CDETR_WGRAD_TWINS") ? atoi(getenv("CDETR_WGRAD_TWINS")) : 1;
    return on && d.precision == 3 && d.dY16 && d.X16 && !d.dbias && (d.Nout & 7) == 0 && (d.Cin & 7) == 0 && (d.ldy & 7) == 0 && (d.ldx & 7) == 0 &&
           (d.sY & 7) == 0 && (d.sX & 7) == 0 && (d.sY2 & 7) == 0 && (d.sX2 & 7) == 0 && aligned16(d.dY16) && aligned16(d.X16) && d.Nout >= 8 && d.Cin >= 8;
}
// 32-pixel blocks per staged k-tile of the twin-fed kernels (wgrad_tr16_body's KP): CDETR_WGRAD_KP = 1 / 2 / 4 (A/B knob, read per call by the tools)
int wgrad_kp() {
    static const int kp_env = getenv("CDETR_WGRAD_KP") ? atoi(getenv("CDETR_WGRAD_KP")) : 1;
    const char* e = cdetr_tune_env("CDETR_WGRAD_KP");
    const int kp = e ? atoi(e) : kp_env;      // (default 1: see profiles/r5_ab_wgrad_kp.txt)
    return kp >= 4 ? 4 : kp >= 2 ? 2 : 1;
}
bool wgrad_is_fast(const cdetr_wgrad_desc& d) { return (d.ldy & 3) == 0 && (d.ldx & 3) == 0 && (d.Nout & 3) == 0 && (d.Cin & 3) == 0; }
}  // namespace

extern "C" int cdetr_wgrad(const cdetr_wgrad_desc* dp, void* stream) {
    CDETR_CHECK_ARG(dp != nullptr, "cdetr_wgrad: null descriptor");
    cdetr_wgrad_desc d = *dp;
    if (int rcv = check_wgrad_desc(d)) return rcv;
    if (d.P == 0) return CDETR_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);

    if (wgrad_is_direct(d)) {
        int tilesI, tilesJ, per, slices;
        direct_wgrad_plan(d, tilesI, tilesJ, per, slices);
        dim3 grid((tilesI * tilesJ + 3) / 4, slices, d.batch);
        hipLaunchKernelGGL(wgrad_direct_kernel, grid, dim3(256), 0, st, d, tilesI, tilesJ, per, d.dbias);
        return cdetr_launch_status("cdetr_wgrad");
    }
    const bool fast = wgrad_is_fast(d);
    if (fast) {
        const int nktf = (d.P + 31) / 32;
        int rcf = CDETR_OK;
        auto launchf = [&](auto bi_c, auto bj_c) {
            constexpr int BI = decltype(bi_c)::value, BJ = decltype(bj_c)::value;
            const int tilesI = (d.Nout + BI - 1) / BI, tilesJ = (d.Cin + BJ - 1) / BJ;
            const long base = (long)tilesI * tilesJ * d.taps * d.batch;
            static const long target_env = getenv("CDETR_WGRAD_TARGET") ? atol(getenv("CDETR_WGRAD_TARGET")) : 0;
            // ~3 workgroups per CU; weights of <= 16 tiles pay slices x (atomic epilogue + pipeline fill) for little parallelism
            // gained: half as many slices measured 10-15 % faster there (5000x256x256, 20000x512x128)
            // (round 5: 384 for every problem -- beside the data-gradient chain fewer, longer workgroups disturb it less: profiles/r5_ab_wgrad.txt)
            // round 6: wg_target == 0 is again the rule for a launch that has the chip to itself (768; 384 for weights of <= 16
            // tiles); a caller that runs the launch BESIDE a data-gradient chain asks for 384 itself (engine.Trainer: ops.wgrad_flush(wg_target=))
            const long target = d.wg_target > 0 ? d.wg_target : (target_env ? target_env : (base <= 16 ? 384 : 768));
            long slices = (target + base - 1) / base;
            const long max_slices = (nktf + 3) / 4;                 // >= 4 k-tiles (128 pixels) per slice
            if (slices > max_slices) slices = max_slices;
            if (slices < 1) slices = 1;
            if (slices > 65535) slices = 65535;
            int per = (int)((nktf + slices - 1) / slices);
            slices = (nktf + per - 1) / per;
            const int bytes = (2 * 32 * (BI + 4) + 2 * 32 * (BJ + 4)) * 4;
            dim3 grid(tilesI * tilesJ * d.taps, (unsigned)slices, d.batch), block(256);
            // split-bf16: the LDS transpose-read kernel (wgrad_tr_kernel); fp32 MFMA: wgrad_fast_kernel
            if (wgrad_has_twins(d)) {
                auto go16 = [&](auto kp_c) {
                    constexpr int KP = decltype(kp_c)::value;
                    const int tbytes = 2 * ((BI + BJ) / 32) * (KP * 32 * 32 + 32) * 2;   // two buffers of hi planes (padded blocks)
                    if ((rcf = raise_lds(wgrad_tr16_kernel<BI, BJ, KP>, tbytes, "cdetr_wgrad"))) return;
                    static const int xcd16 = getenv("CDETR_WGRAD_XCD16") ? atoi(getenv("CDETR_WGRAD_XCD16")) : 1;
                    if (xcd16 && d.batch == 1 && slices >= 8 && max_slices >= 8) {      // XCD-aware slices: a multiple of 8 of them, 1-D grid
                        long s8 = std::min<long>((slices + 7) / 8 * 8, max_slices / 8 * 8);
                        if (s8 < 8) s8 = 8;
                        const int per8 = (int)((nktf + s8 - 1) / s8);
                        const int nx = tilesI * tilesJ * d.taps;
                        hipLaunchKernelGGL((wgrad_tr16_kernel<BI, BJ, KP>), dim3((unsigned)(nx * s8)), block, tbytes, st, d, tilesI, tilesJ, per8, nx);
                    } else
                        hipLaunchKernelGGL((wgrad_tr16_kernel<BI, BJ, KP>), grid, block, tbytes, st, d, tilesI, tilesJ, per, 0);
                };
                const int kp = wgrad_kp();
                if (kp == 4 && BI + BJ <= 192) go16(std::integral_constant<int, 4>{});
                else if (kp >= 2) go16(std::integral_constant<int, 2>{});
                else go16(std::integral_constant<int, 1>{});
            } else if (d.precision >= 1) {
                const int tbytes = 2 * 2 * (BI + BJ) * 32 * 2;
                if (d.precision == 2) {
                    if ((rcf = raise_lds(wgrad_tr_kernel<BI, BJ, 2>, tbytes, "cdetr_wgrad"))) return;
                    hipLaunchKernelGGL((wgrad_tr_kernel<BI, BJ, 2>), grid, block, tbytes, st, d, tilesI, tilesJ, per, d.dbias);
                } else if (d.precision == 3) {
                    if ((rcf = raise_lds(wgrad_tr_kernel<BI, BJ, 1>, tbytes, "cdetr_wgrad"))) return;
                    hipLaunchKernelGGL((wgrad_tr_kernel<BI, BJ, 1>), grid, block, tbytes, st, d, tilesI, tilesJ, per, d.dbias);
                } else {
                    if ((rcf = raise_lds(wgrad_tr_kernel<BI, BJ>, tbytes, "cdetr_wgrad"))) return;
                    hipLaunchKernelGGL((wgrad_tr_kernel<BI, BJ>), grid, block, tbytes, st, d, tilesI, tilesJ, per, d.dbias);
                }
            } else {
                if ((rcf = raise_lds(wgrad_fast_kernel<BI, BJ, 0>, bytes, "cdetr_wgrad"))) return;
                hipLaunchKernelGGL((wgrad_fast_kernel<BI, BJ, 0>), grid, block, bytes, st, d, tilesI, tilesJ, per, d.dbias);
            }
        };
        const char* wf = cdetr_tune_env("CDETR_WGRAD_VARIANT");   // tuning knob: 1 = 128x128, 2 = 128x64, 3 = 64x128, 4 = 64x64
        const int wforce = wf ? atoi(wf) : 0;
        if (wforce == 1) launchf(std::integral_constant<int, 128>{}, std::integral_constant<int, 128>{});
        else if (wforce == 2) launchf(std::integral_constant<int, 128>{}, std::integral_constant<int, 64>{});
        else if (wforce == 3) launchf(std::integral_constant<int, 64>{}, std::integral_constant<int, 128>{});
        else if (wforce == 4) launchf(std::integral_constant<int, 64>{}, std::integral_constant<int, 64>{});
        if (wforce) { if (rcf) return rcf; return cdetr_launch_status("cdetr_wgrad"); }
        // measured (tools/gemm_sweep.py wgrad, profiles/r1_gemm_sweep.txt): 64x64 tiles win for 1x1 / linear layers,
        // 128-wide tiles only for the 3x3 convolutions (9 taps = 9x more output tiles per k-slice)
        if (d.taps > 1 && d.Nout >= 512 && d.Cin >= 512)
            launchf(std::integral_constant<int, 128>{}, std::integral_constant<int, 128>{});
        else if (d.precision >= 1) {
            // transpose-read kernel (profiles/r1_gemm_sweep.txt, wgrad section): 64x128 once the weight has >= 1M elements, else 64x64
            if (d.taps == 1 && (long)d.Nout * d.Cin >= (1L << 20) && d.Cin >= 128)
                launchf(std::integral_constant<int, 64>{}, std::integral_constant<int, 128>{});
            else
                launchf(std::integral_constant<int, 64>{}, std::integral_constant<int, 64>{});
        } else if (d.taps > 1 && d.Nout >= 128)
            launchf(std::integral_constant<int, 128>{}, std::integral_constant<int, 64>{});
        else
            launchf(std::integral_constant<int, 64>{}, std::integral_constant<int, 64>{});
        if (rcf) return rcf;
        return cdetr_launch_status("cdetr_wgrad");
    }
    const int nkt = (d.P + BK - 1) / BK;
    if (d.dbias) {   // generic path: separate column-sum launch
        const int rows_per_block = 64;
        dim3 cg((d.Nout + 255) / 256, (d.P + rows_per_block - 1) / rows_per_block, 1);
        for (int zb = 0; zb < d.batch; ++zb)
            hipLaunchKernelGGL(colsum_kernel, cg, dim3(256), 0, st, d.dY + batch_off(zb, d.batch_inner, d.sY, d.sY2), (long)d.ldy, d.P, d.Nout, d.dbias,
                               rows_per_block);
    }
    auto launch = [&](auto bi_c, auto bj_c) {
        constexpr int BI = decltype(bi_c)::value, BJ = decltype(bj_c)::value;
        const int tilesI = (d.Nout + BI - 1) / BI, tilesJ = (d.Cin + BJ - 1) / BJ;
        const long base = (long)tilesI * tilesJ * d.taps * d.batch;
        long slices = (1024 + base - 1) / base;                 // aim at ~4 workgroups per CU
        const long max_slices = (nkt + 7) / 8;                  // >= 8 k-tiles (128 pixels) per slice
        if (slices > max_slices) slices = max_slices;
        if (slices < 1) slices = 1;
        if (slices > 65535) slices = 65535;
        int per = (int)((nkt + slices - 1) / slices);
        slices = (nkt + per - 1) / per;
        dim3 grid(tilesI * tilesJ * d.taps, (unsigned)slices, d.batch), block(256);
        hipLaunchKernelGGL((wgrad_kernel<BI, BJ>), grid, block, 0, st, d, tilesI, tilesJ, per);
    };
    if (d.Nout >= 128 && d.Cin >= 128)
        launch(std::integral_constant<int, 128>{}, std::integral_constant<int, 128>{});
    else if (d.Nout >= 128)
        launch(std::integral_constant<int, 128>{}, std::integral_constant<int, 64>{});
    else if (d.Cin >= 128)
        launch(std::integral_constant<int, 64>{}, std::integral_constant<int, 128>{});
    else
        launch(std::integral_constant<int, 64>{}, std::integral_constant<int, 64>{});
    return cdetr_launch_status("cdetr_wgrad");
}

// A batch of INDEPENDENT weight-gradient problems.  Problems that the 64x64 transpose-read kernel or the few-pixel kernel would
// take anyway are concatenated into grouped launches (wgrad_tr_group_kernel / wgrad_direct_group_kernel, <= WG_MAX problems
// each); everything else runs through cdetr_wgrad one by one.  A layer's weight gradients depend on nothing but their own dY / X
// and nothing but the optimizer consumes them, so the host queues them and submits them together: the ~100 few-pixel launches of
// a step are latency bound (~7 us each for microseconds of work) and the encoder's 16-tile problems cannot fill the chip alone.
extern "C" int cdetr_wgrad_group(const cdetr_wgrad_desc* descs, int32_t n, void* stream) {
    CDETR_CHECK_ARG(n >= 0 && (descs != nullptr || n == 0), "cdetr_wgrad_group: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static const int grouping = getenv("CDETR_WGRAD_GROUP") ? atoi(getenv("CDETR_WGRAD_GROUP")) : 1;
    const char* wfg = cdetr_tune_env("CDETR_WGRAD_VARIANT");
    const bool forced = wfg && atoi(wfg) != 0;
    std::vector<int> direct, tr64p, tr64t;                    // tr64t: the same class with bf16 twins (wgrad_tr16_group_kernel)
    for (int i = 0; i < n; ++i) {
        const cdetr_wgrad_desc& d = descs[i];
        if (int rcv = check_wgrad_desc(d)) return rcv;
        if (d.P == 0) continue;
        if (grouping && wgrad_is_direct(d)) direct.push_back(i);
        else if (grouping && wgrad_is_fast(d) && d.precision >= 1 && !forced &&
                 !(d.taps > 1 && d.Nout >= 512 && d.Cin >= 512) && !(d.taps == 1 && (long)d.Nout * d.Cin >= (1L << 20) && d.Cin >= 128))
            (wgrad_has_twins(d) ? tr64t : tr64p).push_back(i);      // = the shapes cdetr_wgrad gives to wgrad_tr_kernel<64, 64>
        else if (int rc1 = cdetr_wgrad(&d, stream)) return rc1;
    }
    for (size_t c0 = 0; c0 < direct.size(); c0 += WG_MAX) {
        const int m = (int)std::min<size_t>(WG_MAX, direct.size() - c0);
        if (m == 1) { if (int rc1 = cdetr_wgrad(&descs[direct[c0]], stream)) return rc1; continue; }
        WgradGroupArgs g;
        g.n = m; g.blk0[0] = 0;
        for (int k = 0; k < m; ++k) {
            WgradGroupItem& it = g.it[k];
            it.d = descs[direct[c0 + k]];
            int slices;
            direct_wgrad_plan(it.d, it.tilesI, it.tilesJ, it.per, slices);
            it.nx = (it.tilesI * it.tilesJ + 3) / 4; it.ny = slices; it.pad_ = 0;
            g.blk0[k + 1] = g.blk0[k] + it.nx * it.ny * it.d.batch;
        }
        hipLaunchKernelGGL(wgrad_direct_group_kernel, dim3(g.blk0[m]), dim3(256), 0, st, g);
        if (int rcl = cdetr_launch_status("cdetr_wgrad_group")) return rcl;
    }
    for (int twins = 0; twins < 2; ++twins) {
    const std::vector<int>& tr64 = twins ? tr64t : tr64p;
    for (size_t c0 = 0; c0 < tr64.size(); c0 += WG_MAX) {
        const int m = (int)std::min<size_t>(WG_MAX, tr64.size() - c0);
        if (m == 1) { if (int rc1 = cdetr_wgrad(&descs[tr64[c0]], stream)) return rc1; continue; }
        WgradGroupArgs g;
        g.n = m; g.blk0[0] = 0;
        // one common slice length (k-tiles per workgroup) for the whole launch: ~3 workgroups per CU in total, equal work each
        long work = 0;
        for (int k = 0; k < m; ++k) {
            const cdetr_wgrad_desc& d = descs[tr64[c0 + k]];
            work += (long)((d.Nout + 63) / 64) * ((d.Cin + 63) / 64) * d.taps * d.batch * ((d.P + 31) / 32);
        }
        // 768 for a launch alone on the chip; launches beside the data-gradient chain ask for 384 through wg_target (profiles/r5_ab_wgrad.txt)
        static const long gtarget = getenv("CDETR_WGRAD_GROUP_TARGET") ? atol(getenv("CDETR_WGRAD_GROUP_TARGET")) : 768;
        long want = 0;                                    // cdetr_wgrad_desc.wg_target: the largest request of the members (a launch alone on the chip)
        for (int k = 0; k < m; ++k) want = std::max<long>(want, descs[tr64[c0 + k]].wg_target);
        const long gt = want > 0 ? want : gtarget;
        long per_all = (work + gt - 1) / gt;
        if (per_all < 4) per_all = 4;
        for (int k = 0; k < m; ++k) {
            WgradGroupItem& it = g.it[k];
            it.d = descs[tr64[c0 + k]];
            const int nkt = (it.d.P + 31) / 32;
            it.tilesI = (it.d.Nout + 63) / 64; it.tilesJ = (it.d.Cin + 63) / 64;
            it.per = (int)std::min<long>(per_all, nkt);
            it.nx = it.tilesI * it.tilesJ * it.d.taps; it.ny = (nkt + it.per - 1) / it.per; it.pad_ = 0;
            static const int xcd16 = getenv("CDETR_WGRAD_XCD16") ? atoi(getenv("CDETR_WGRAD_XCD16")) : 1;
            if (twins && xcd16 && it.d.batch == 1 && it.ny >= 6 && (nkt + 3) / 4 >= 8) {      // XCD-aware slices: a multiple of 8 (see wgrad_tr16_kernel)
                it.ny = (it.ny + 7) / 8 * 8;
                it.per = (nkt + it.ny - 1) / it.ny;
                it.pad_ = 1;
            }
            g.blk0[k + 1] = g.blk0[k] + (it.nx * it.ny * it.d.batch + 7) / 8 * 8;
        }
        // one precision per grouped launch: the group's members come from one backward pass, the first member decides
        const int gprec = g.it[0].d.precision;
        if (twins) {
            const int kp = wgrad_kp();
            if (kp == 4) {
                if (int rcl = raise_lds(wgrad_tr16_group_kernel<64, 64, 4>, 2 * 4 * (4 * 32 * 32 + 32) * 2, "cdetr_wgrad_group")) return rcl;
                hipLaunchKernelGGL((wgrad_tr16_group_kernel<64, 64, 4>), dim3(g.blk0[m]), dim3(256), 2 * 4 * (4 * 32 * 32 + 32) * 2, st, g);
            } else if (kp == 2) hipLaunchKernelGGL((wgrad_tr16_group_kernel<64, 64, 2>), dim3(g.blk0[m]), dim3(256), 2 * 4 * (2 * 32 * 32 + 32) * 2, st, g);
            else hipLaunchKernelGGL((wgrad_tr16_group_kernel<64, 64>), dim3(g.blk0[m]), dim3(256), 2 * 4 * (32 * 32 + 32) * 2, st, g);
        }
        else if (gprec == 2) hipLaunchKernelGGL((wgrad_tr_group_kernel<64, 64, 2>), dim3(g.blk0[m]), dim3(256), 2 * 2 * (64 + 64) * 32 * 2, st, g);
        else if (gprec == 3) hipLaunchKernelGGL((wgrad_tr_group_kernel<64, 64, 1>), dim3(g.blk0[m]), dim3(256), 2 * 2 * (64 + 64) * 32 * 2, st, g);
        else hipLaunchKernelGGL((wgrad_tr_group_kernel<64, 64>), dim3(g.blk0[m]), dim3(256), 2 * 2 * (64 + 64) * 32 * 2, st, g);
        if (int rcl = cdetr_launch_status("cdetr_wgrad_group")) return rcl;
    }
    }
    return CDETR_OK;
}

extern "C" int cdetr_colsum(const float* X, int64_t ldx, int32_t M, int32_t N, float* out, void* stream) {
    CDETR_CHECK_ARG(X && out && M >= 0 && N > 0, "cdetr_colsum: bad args");
    if (M == 0) return CDETR_OK;
    const int rows_per_block = 64;
    dim3 grid((N + 255) / 256, (M + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), X, (long)ldx, M, N, out,
                       rows_per_block);
    return cdetr_launch_status("cdetr_colsum");
}

extern "C" int cdetr_maxpool3x3s2(const float* X, float* Y, int32_t Nimg, int32_t H, int32_t W, int32_t C, void* stream) {
    return cdetr_maxpool3x3s2_split(X, Y, nullptr, nullptr, Nimg, H, W, C, stream);
}

extern "C" int cdetr_maxpool3x3s2_split(const float* X, float* Y, void* Y16, void* Y16lo, int32_t Nimg, int32_t H, int32_t W, int32_t C,
                                        void* stream) {
    CDETR_CHECK_ARG(X && Y && Nimg > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && (Y16 || !Y16lo), "cdetr_maxpool: bad args");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long total = (long)Nimg * Ho * Wo * (C / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), X, Y,
                       reinterpret_cast<__bf16*>(Y16), reinterpret_cast<__bf16*>(Y16lo), Nimg, H, W, C, Ho, Wo);
    return cdetr_launch_status("cdetr_maxpool3x3s2");
}
