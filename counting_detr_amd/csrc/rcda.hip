// rcda.hip -- fused Row-Column Decoupled Attention core for gfx950.
//
// Reference math (A2/models/row_column_decoupled_attention.py:215-309), per (image n, head):
//   A_row = softmax_W(scale * q_row k_row^T)   [L,W]        A_col = softmax_H(scale * q_col k_col^T)   [L,H]
//   out[q,c] = sum_h sum_w A_col[q,h] A_row[q,w] V[h,w,c]
// The reference materialises T[q,w,c] = sum_h A_col[q,h] V[h,w,c] ([nh,L,W,32] fp32 = 128 MB per encoder call per image
// at 800x800) plus several permuted copies.  Here nothing but A_row/A_col (2 x 4 MB) ever reaches HBM:
//   forward : one workgroup = 128 queries (4 waves x 32) of one (n, head).  Phase 1 computes both logit matrices with
//             VALU FMAs (lanes 0-31 own a full row of S_row, lanes 32-63 a full row of S_col: no cross-lane reduction
//             in the softmax).  Phase 2 is one long MFMA chain per wave: for every key column w the A operand
//             P_w[q,h] = A_col[q,h] * A_row[q,w] is formed in registers and multiplied with the LDS-staged V[:,w,:] tile,
//             accumulating out[32 x 32] in 16 accumulator registers (v_mfma_f32_32x32x2_f32).
//   backward: G_w^T[h,q] = sum_c V[h,w,c] dOut[q,c] by MFMA (V tile as A operand, dOut^T loop-invariant in registers);
//             in this transposed layout dA_col accumulates element-wise and dA_row[q,w] = sum_h A_col[q,h] G_w[q,h] is an
//             in-lane reduction plus one cross-half add; the two softmax backward passes follow in-register.
//             dV[h,w,c] = sum_q A_col[q,h] A_row[q,w] dOut[q,c] is a second kernel (MFMA over the query axis).
// Head dim is fixed at 32 (E = 256, 8 heads in every shipped config).
#include "../../include/cdetr_hip.h"
#include "common.h"
#include <algorithm>
#include <type_traits>
#include <stdlib.h>

namespace {

// XCD-aware placement of a 2-D grid (x = tile of one (image, head), y = (image, head)): workgroup b of the launch runs on XCD b % 8, each
// with a private 4 MiB L2.  In dispatch order the x-tiles of ONE (image, head) land on all 8 XCDs, so every XCD fetches every head's V / K /
// attention rows from the fabric (PMC round 2: 4.5x the algorithmic bytes in the RCDA backward, 2x in the forward).  Remapped, XCD c owns
// the contiguous range [c * total / 8, (c + 1) * total / 8) of the row-major (y, x) index space -- whole (image, head) slabs, whose operands
// then come from the fabric once.  Identity when the grid is not a multiple of 8 workgroups.  Placement only affects speed.
__device__ __forceinline__ void xcd_slab(int& bx, int& by) {
    const int gx = gridDim.x, total = gx * gridDim.y;
    if (total & 7) return;
    const int lin = by * gx + bx;
    const int l2 = (lin & 7) * (total >> 3) + (lin >> 3);
    by = l2 / gx;
    bx = l2 - by * gx;
}


constexpr int D = 32;          // head dim
constexpr int QW = 32;         // queries per wave

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

struct FwdSmem {
    int sw, sh;          // row strides of the per-wave S_row / S_col tiles
    int off_srow, off_scol, off_k, off_v;
    int total;
};
__host__ __device__ inline FwdSmem fwd_smem(int H, int W, int NW) {
    FwdSmem s;
    const int Wp = (W + 3) & ~3, Hp = (H + 7) & ~7;
    s.sw = Wp + 1;
    s.sh = Hp + 4;
    s.off_srow = 0;
    s.off_scol = s.off_srow + NW * QW * s.sw;
    s.off_scol = (s.off_scol + 3) & ~3;
    s.off_k = s.off_scol + NW * QW * s.sh;
    const int kbytes = (W + H) * D;      // k_row + k_col tiles (phase 1)
    const int vbytes = 2 * Hp * D;       // double-buffered V tile (phase 2), overlays the k tiles
    s.off_v = s.off_k;
    s.total = s.off_k + (kbytes > vbytes ? kbytes : vbytes);
    return s;
}

// Copy `rows` x `cols` (cols % 4 == 0, 16-byte aligned rows, row stride = cols) from global into an LDS tile of QW rows (row stride
// lds_stride, any parity), zero-filling the rows >= rows_valid.  The global loads go out as batches of CH unconditional 16-byte loads
// per lane (clamped index, masked use): a load inside `if (row < rows_valid)` / a runtime-trip loop of load -> LDS store makes the
// wave wait for every round trip in turn (26 + 28 of them in the dS kernel's prologue before this was batched).
template <int CH>
__device__ __forceinline__ void stage_rows(const float* __restrict__ gsrc, int rows_valid, int cols, float* lds, int lds_stride, int lane) {
    const int cols4 = cols >> 2;
    const int total4 = QW * cols4;
    const int last4 = max(rows_valid * cols4 - 1, 0);
    for (int base = 0; base < total4; base += 64 * CH) {
        float4 t[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) t[u] = ld4(gsrc + 4 * (long)min(base + lane + 64 * u, last4));
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int i4 = base + lane + 64 * u;
            if (i4 < total4) {
                const int r = i4 / cols4, c = (i4 - r * cols4) * 4;
                const bool ok = r < rows_valid;
                float* dst = lds + r * lds_stride + c;
                dst[0] = ok ? t[u].x : 0.f; dst[1] = ok ? t[u].y : 0.f; dst[2] = ok ? t[u].z : 0.f; dst[3] = ok ? t[u].w : 0.f;
            }
        }
    }
}

// Two tensors staged together (the dS kernel's A_row and A_col rows): when each fits one batch, ALL loads of both are issued before the first LDS
// store -- one global round trip instead of two (round-6 probe: 5.2 us of a 55 us workgroup went into the two stage_rows calls) -- and the
// (row, column) of a slot advances incrementally instead of by a division per slot.
template <int CH>
__device__ __forceinline__ void stage_rows_pair(const float* __restrict__ g0, int cols0, float* lds0, int stride0,
                                                const float* __restrict__ g1, int cols1, float* lds1, int stride1, int rows_valid, int lane) {
    const int c40 = cols0 >> 2, c41 = cols1 >> 2;
    if (QW * c40 > 64 * CH || QW * c41 > 64 * CH) {
        stage_rows<CH>(g0, rows_valid, cols0, lds0, stride0, lane);
        stage_rows<CH>(g1, rows_valid, cols1, lds1, stride1, lane);
        return;
    }
    float4 t0[CH], t1[CH];
    const int last0 = max(rows_valid * c40 - 1, 0), last1 = max(rows_valid * c41 - 1, 0);
#pragma unroll
    for (int u = 0; u < CH; ++u) t0[u] = ld4(g0 + 4 * (long)min(lane + 64 * u, last0));
#pragma unroll
    for (int u = 0; u < CH; ++u) t1[u] = ld4(g1 + 4 * (long)min(lane + 64 * u, last1));
    auto put = [&](const float4 (&t)[CH], int cols4, float* lds, int stride) __attribute__((always_inline)) {
        const int dr = 64 / cols4, dc = 64 - dr * cols4;
        int r = lane / cols4, c4 = lane - r * cols4;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (r < QW) {
                const bool ok = r < rows_valid;
                float* dst = lds + r * stride + 4 * c4;
                dst[0] = ok ? t[u].x : 0.f; dst[1] = ok ? t[u].y : 0.f; dst[2] = ok ? t[u].z : 0.f; dst[3] = ok ? t[u].w : 0.f;
            }
            r += dr; c4 += dc;
            if (c4 >= cols4) { c4 -= cols4; ++r; }
        }
    };
    put(t0, c40, lds0, stride0);
    put(t1, c41, lds1, stride1);
}

// The reverse of stage_rows: `rows` x `cols` (cols % 4 == 0) from an LDS tile (row stride lds_stride, any parity) to a contiguous, 16-byte
// aligned global block, 16-byte stores, (row, column) advanced incrementally -- the per-element `idx / cols` of a flat loop costs ~40
// instructions each and was a third of the score phase.
__device__ __forceinline__ void save_rows(const float* lds, int lds_stride, float* __restrict__ gdst, int rows, int cols, int lane) {
    const int cols4 = cols >> 2;
    const int dr = 64 / cols4, dc = 64 - dr * cols4;        // one trip advances 64 float4 = dr rows + dc quads
    int r = lane / cols4, c4 = lane - r * cols4;
    const int total4 = rows * cols4;
    for (int i4 = lane; i4 < total4; i4 += 64) {
        const float* src = lds + r * lds_stride + 4 * c4;
        *reinterpret_cast<float4*>(gdst + 4 * (long)i4) = make_float4(src[0], src[1], src[2], src[3]);
        r += dr; c4 += dc;
        if (c4 >= cols4) { c4 -= cols4; ++r; }
    }
}

// Phases 0-1 shared by both forward kernels: stage the projected keys of (n, head), compute both logit matrices with VALU
// FMAs (lanes 0-31 own a full row of S_row, lanes 32-63 a full row of S_col: the softmax needs no cross-lane traffic),
// leave A_row / A_col in this wave's LDS tiles and save them for the backward pass.  Ends with the K tiles dead.
template <int NT>
__device__ __forceinline__ void rcda_scores(const cdetr_rcda_fwd_desc& d, const FwdSmem& sm, float* smem, float* Srow, float* Scol,
                                            int tid, int lane, int i32, int g, int n, int head, int qbase, int q, bool qvalid) {
    const int H = d.H, W = d.W, L = d.L, E = d.nh * D;
    const int Wp = (W + 3) & ~3, Hp = (H + 7) & ~7;
    float* Krow = smem + sm.off_k;             // [W][32]
    float* Kcol = Krow + W * D;                // [H][32]

    // ---- phase 0: stage the projected keys of this (n, head)
    {   // batches of 4 unconditional loads per thread (clamped key, masked store)
        const int nk8 = (W + H) * 8;
        for (int base = 0; base < nk8; base += 4 * NT) {
            float4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = min(base + tid + NT * u, nk8 - 1);
                const int key = idx >> 3, c4 = idx & 7;
                const float* src = (key < W) ? d.k_row + ((long)n * W + key) * E + head * D + c4 * 4
                                             : d.k_col + ((long)n * H + (key - W)) * E + head * D + c4 * 4;
                t[u] = ld4(src);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + tid + NT * u;
                if (idx < nk8) *reinterpret_cast<float4*>(Krow + (idx >> 3) * D + (idx & 7) * 4) = t[u];
            }
        }
    }
    // this lane's query vector: half 0 -> q_row, half 1 -> q_col
    float qv[D];
    {
        const float* qp = (g == 0 ? d.q_row : d.q_col) + ((long)n * L + (qvalid ? q : 0)) * E + head * D;
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            float4 t = ld4(qp + c4 * 4);
            if (!qvalid) t = make_float4(0.f, 0.f, 0.f, 0.f);
            qv[c4 * 4 + 0] = t.x; qv[c4 * 4 + 1] = t.y; qv[c4 * 4 + 2] = t.z; qv[c4 * 4 + 3] = t.w;
        }
    }
    __syncthreads();

    // ---- phase 1: logits + softmax; every lane owns one full row (no cross-lane traffic)
    {
        const int nkeys = (g == 0) ? W : H;
        const int npad = (g == 0) ? Wp : Hp;
        const float* Ks = (g == 0) ? Krow : Kcol;
        float* S = (g == 0) ? Srow + i32 * sm.sw : Scol + i32 * sm.sh;
        const uint8_t* mk = (g == 0) ? d.mask_row : d.mask_col;
        if (mk) mk += (long)n * nkeys;
        float mx = -INFINITY;
#pragma unroll 2
        for (int k = 0; k < nkeys; ++k) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int c4 = 0; c4 < 8; ++c4) {
                const float4 kv = *reinterpret_cast<const float4*>(Ks + k * D + c4 * 4);
                s0 = fmaf(qv[c4 * 4 + 0], kv.x, s0);
                s1 = fmaf(qv[c4 * 4 + 1], kv.y, s1);
                s2 = fmaf(qv[c4 * 4 + 2], kv.z, s2);
                s3 = fmaf(qv[c4 * 4 + 3], kv.w, s3);
            }
            float s = ((s0 + s1) + (s2 + s3)) * d.scale;
            if (mk && mk[k]) s = -INFINITY;
            S[k] = s;
            mx = fmaxf(mx, s);
        }
        float sum = 0.f;
#pragma unroll 4
        for (int k = 0; k < nkeys; ++k) {
            const float e = expf(S[k] - mx);
            S[k] = e;
            sum += e;
        }
        const float inv = 1.f / sum;
#pragma unroll 4
        for (int k = 0; k < nkeys; ++k) S[k] *= inv;
        for (int k = nkeys; k < npad; ++k) S[k] = 0.f;
    }
    __syncthreads();   // all waves done with Krow/Kcol (the V buffers overlay them)

    // ---- save A_row / A_col for the backward pass (coalesced copy of the wave's contiguous block)
    {
        const int nq = min(QW, L - qbase);   // may be <= 0 for tail waves
        if (nq > 0 && d.a_row != nullptr) {      // (a_row == a_col == NULL: inference, nothing saved for a backward pass)
            save_rows(Srow, sm.sw, d.a_row + (((long)n * d.nh + head) * L + qbase) * Wp, nq, Wp, lane);
            save_rows(Scol, sm.sh, d.a_col + (((long)n * d.nh + head) * L + qbase) * Hp, nq, Hp, lane);
        }
    }

}

// ------------------------------------------------------------------------------------------------ forward
template <int NF, int NW, int PREC>   // NF = 32-row key fragments along H (H <= 32*NF); NW = waves (x32 queries) per workgroup; PREC 0 = fp32 MFMA, 1 = split-bf16 x3
__global__ __launch_bounds__(64 * NW) void rcda_fwd_kernel(const cdetr_rcda_fwd_desc d) {
    constexpr int NT = 64 * NW, QB = QW * NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KH8 = NF * 4;
    const int H = d.H, W = d.W, L = d.L, E = d.nh * D;
    const int Hp = (H + 7) & ~7;
    const FwdSmem sm = fwd_smem(H, W, NW);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i32 = lane & 31, g = lane >> 5;
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    xcd_slab(bx_, by_);
    const int n = by_ / d.nh, head = by_ % d.nh;
    const int qbase = bx_ * QB + wid * QW;
    const int q = qbase + i32;
    const bool qvalid = q < L;

    float* Srow = smem + sm.off_srow + wid * QW * sm.sw;
    float* Scol = smem + sm.off_scol + wid * QW * sm.sh;
    rcda_scores<NT>(d, sm, smem, Srow, Scol, tid, lane, i32, g, n, head, qbase, q, qvalid);

    // ---- phase 2: out = sum_w (A_col * A_row[:,w]) . V[:,w,:]
    float* Vs = smem + sm.off_v;   // [2][Hp][32]
    // zero the padded key rows of both buffers once
    for (int idx = tid; idx < 2 * (Hp - H) * D; idx += NT) {
        const int b = idx / ((Hp - H) * D), r = idx - b * (Hp - H) * D;
        Vs[b * Hp * D + H * D + r] = 0.f;
    }
    float4 acol[KH8];
#pragma unroll
    for (int kk = 0; kk < KH8; ++kk)
        acol[kk] = (kk * 8 < Hp) ? *reinterpret_cast<const float4*>(Scol + i32 * sm.sh + kk * 8 + g * 4)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);

    const float* vbase = d.v + (long)n * H * W * E + head * D;   // + (h*W + w)*E + c
    constexpr int VSLOTS = (NF * 256 + NT - 1) / NT;   // float4 per thread per tile: H*8 <= 256*NF
    float4 rv[VSLOTS];
    auto vfetch = [&](int w) {
#pragma unroll
        for (int s = 0; s < VSLOTS; ++s) {
            const int idx = tid + NT * s;
            const int h = idx >> 3, c4 = idx & 7;
            rv[s] = (h < H) ? ld4(vbase + ((long)h * W + w) * E + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto vstash = [&](int buf) {
#pragma unroll
        for (int s = 0; s < VSLOTS; ++s) {
            const int idx = tid + NT * s;
            const int h = idx >> 3, c4 = idx & 7;
            if (h < H) *reinterpret_cast<float4*>(Vs + buf * Hp * D + h * D + c4 * 4) = rv[s];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    vfetch(0);
    vstash(0);
    __syncthreads();
    for (int w = 0; w < W; ++w) {
        const int buf = w & 1;
        if (w + 1 < W) vfetch(w + 1);
        const float arow = Srow[i32 * sm.sw + w];
        const float* vb = Vs + buf * Hp * D + (g * 4) * D + i32;
        if (PREC == 0) {
#pragma unroll
            for (int kk = 0; kk < KH8; ++kk) {
                if (kk * 8 < Hp) {
                    const float p0 = acol[kk].x * arow, p1 = acol[kk].y * arow, p2 = acol[kk].z * arow, p3 = acol[kk].w * arow;
                    const float b0 = vb[(kk * 8 + 0) * D], b1 = vb[(kk * 8 + 1) * D], b2 = vb[(kk * 8 + 2) * D],
                                b3 = vb[(kk * 8 + 3) * D];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p0, b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p1, b1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p2, b2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p3, b3, acc, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int kp = 0; kp < KH8 / 2; ++kp) {          // 16 key rows per step: chunks 2kp and 2kp+1
                if (kp * 16 < Hp) {
                    const int k0 = 2 * kp, k1 = 2 * kp + 1;
                    const bool has1 = k1 * 8 < Hp;           // Hp is a multiple of 8, not of 16 (acol[k1] is zero then)
                    const float pa[8] = {acol[k0].x * arow, acol[k0].y * arow, acol[k0].z * arow, acol[k0].w * arow,
                                         acol[k1].x * arow, acol[k1].y * arow, acol[k1].z * arow, acol[k1].w * arow};
                    float vv[8];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        vv[t] = vb[(k0 * 8 + t) * D];
                        vv[4 + t] = has1 ? vb[(k1 * 8 + t) * D] : 0.f;
                    }
                    bf16x8 ah, al, bh, bl;
                    split_bf16x8(pa, ah, al);
                    split_bf16x8(vv, bh, bl);
                    acc = mfma_bf16x3(ah, al, bh, bl, acc);
                }
            }
        }
        if (w + 1 < W) vstash(buf ^ 1);
        __syncthreads();
    }
    mfma_drain(acc);
    // C layout: col = lane&31 (= channel), row = (r&3) + 8*(r>>2) + 4*g (= query within the wave)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qq = qbase + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (qq < L) d.out[((long)n * L + qq) * E + head * D + i32] = acc[r];
    }
}

// ------------------------------------------------------------------------------------------------ forward, two-step form
// split-bf16 only, W <= 96 (KS <= 6 steps of 16 key columns).  The contraction is evaluated as the reference factorises it
//     T_h[q,c] = sum_w A_row[q,w] V[h,w,c]          out[q,c] = sum_h A_col[q,h] T_h[q,c]
// but transposed, so that everything that is constant over h stays in registers:
//   * T_h^T[c,q] = sum_w V_h^T[c,w] A_row^T[w,q] is one short MFMA chain per h (KS = ceil(W/16) k-steps of 3 bf16 MFMAs).  Its
//     B operand (lane = query, 8 consecutive w) is A_row itself: split into bf16 hi/lo ONCE per kernel and kept in 8*KS
//     registers; the A operand is the V_h tile, transposed and split once per workgroup while it is staged into LDS
//     (4 w x 2 c register blocks, ds_write_b64), read back with two ds_read_b128 per k-step and no VALU work;
//   * in the transposed accumulator layout a query is a LANE, so out^T[c,q] += A_col[q,h] * T_h^T[c,q] is 16 FMAs with one
//     per-lane scalar (the form with A_col*A_row as the MFMA operand needs a multiply + bf16 split per element per h);
//   * V tiles are prefetched PD = 4 iterations ahead (8 registers per tile per thread): an iteration is never bound by
//     the global-load latency, which is what limited the one-tile-ahead pipeline of rcda_fwd_kernel.
constexpr int RCDA_RG_DEFAULT = 1;       // key rows per barrier of the two-step forward (see rcda_fwd2_kernel, RG)
constexpr int RCDA_WS_COUNTERS = 4096;   // int32 arrival counters at the head of cdetr_rcda_fwd_desc.ws (= SPLITK_COUNTERS of igemm.hip: one scratch serves both)
constexpr int KSTR = 36;      // LDS row stride (floats) of the projected keys in rcda_scores_mfma: 36 / 4 odd -> conflict-free ds_read_b128

// Score phase on the matrix pipe (split-bf16, H <= 32 TH, W <= 32 TW; TW, TH <= 3): S^T = K Q^T per 32-key tile -- A = the projected keys (row = key,
// k-slot j of step s <-> channel 16s + 8g + j, read from LDS), B = this wave's 32 projected queries (column = query i32, same channel
// slots, straight from global).  The accumulator of a tile holds, for query i32, the logits of keys 32t + (r&3) + 8(r>>2) + 4g: the
// softmax over keys is an in-lane reduction over registers plus one exchange between the two lane halves.  Replaces the VALU loop of
// rcda_scores (50 keys x (8 LDS reads + 32 FMAs) per lane: 9 of the forward kernel's 43 us at the decoder shape).
// Leaves A_row / A_col in the wave's LDS tiles and saves them, like rcda_scores; ends with the key tiles dead.
template <int NT, int TW, int TH, bool PROBE = false>
__device__ __forceinline__ void rcda_scores_mfma(const cdetr_rcda_fwd_desc& d, const FwdSmem& sm, float* smem, float* Srow, float* Scol,
                                                 int tid, int lane, int i32, int g, int n, int head, int qbase, int q, bool qvalid,
                                                 bool save = true, unsigned long long* stamps = nullptr) {
    // PROBE: stamps[0..5] = keys + queries landed (before the barrier) | barrier passed | row side done | column side done | barrier passed | maps saved
    auto stamp = [&](int i) __attribute__((always_inline)) {
        if constexpr (PROBE) { stamps[i] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    };
    const int H = d.H, W = d.W, L = d.L, E = d.nh * D;
    const int Wp = (W + 3) & ~3, Hp = (H + 7) & ~7;
    float* Krow = smem + sm.off_k;             // [W][KSTR]
    float* Kcol = Krow + W * KSTR;             // [H][KSTR]
    {   // stage the projected keys of (n, head): batches of 4 unconditional loads per thread (clamped key, masked store)
        const int nk8 = (W + H) * 8;
        for (int base = 0; base < nk8; base += 4 * NT) {
            float4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = min(base + tid + NT * u, nk8 - 1);
                const int key = idx >> 3, c4 = idx & 7;
                const float* src = (key < W) ? d.k_row + ((long)n * W + key) * E + head * D + c4 * 4
                                             : d.k_col + ((long)n * H + (key - W)) * E + head * D + c4 * 4;
                t[u] = ld4(src);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + tid + NT * u;
                if (idx < nk8) *reinterpret_cast<float4*>(Krow + (idx >> 3) * KSTR + (idx & 7) * 4) = t[u];
            }
        }
    }
    // B operands: this lane's query, channels 16s + 8g .. + 7 of q_row and of q_col
    bf16x8 qh[2][2], ql[2][2];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const float* qp = (side == 0 ? d.q_row : d.q_col) + ((long)n * L + (qvalid ? q : 0)) * E + head * D + 8 * g;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            float4 t0 = ld4(qp + 16 * st), t1 = ld4(qp + 16 * st + 4);
            if (!qvalid) { t0 = make_float4(0.f, 0.f, 0.f, 0.f); t1 = t0; }
            const float x[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
            split_bf16x8(x, qh[side][st], ql[side][st]);
        }
    }
    // key-padding masks as bit sets (one ballot per 64 keys)
    unsigned long long mrow[2] = {0ull, 0ull}, mcol[2] = {0ull, 0ull};      // word 1: keys 64..127 (TW / TH == 3: maps wider than 64)
    if (d.mask_row) {
        const uint8_t* mr = d.mask_row + (long)n * W;
        const uint8_t* mc = d.mask_col + (long)n * H;
        mrow[0] = __ballot(lane < W && mr[min(lane, W - 1)] != 0);
        mcol[0] = __ballot(lane < H && mc[min(lane, H - 1)] != 0);
        if (TW > 2) mrow[1] = __ballot(lane + 64 < W && mr[min(lane + 64, W - 1)] != 0);
        if (TH > 2) mcol[1] = __ballot(lane + 64 < H && mc[min(lane + 64, H - 1)] != 0);
    }
    if constexpr (PROBE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    stamp(0);
    __syncthreads();
    stamp(1);

#pragma unroll
    for (int side = 0; side < 2; ++side) {
        constexpr int TMAX = TW > TH ? TW : TH;
        const int nt = side == 0 ? TW : TH;
        const int nkeys = side == 0 ? W : H, npad = side == 0 ? Wp : Hp;
        const float* Ks = side == 0 ? Krow : Kcol;
        const unsigned long long* mbits = side == 0 ? mrow : mcol;
        float* S = side == 0 ? Srow + i32 * sm.sw : Scol + i32 * sm.sh;
        float sv[TMAX][16];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t >= nt) break;
            const int kc = min(32 * t + i32, nkeys - 1);
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const float4 t0 = *reinterpret_cast<const float4*>(Ks + kc * KSTR + 16 * st + 8 * g);
                const float4 t1 = *reinterpret_cast<const float4*>(Ks + kc * KSTR + 16 * st + 8 * g + 4);
                const float x[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                bf16x8 ah, al;
                split_bf16x8(x, ah, al);
                acc = mfma_bf16x3(ah, al, qh[side][st], ql[side][st], acc);
            }
            mfma_drain(acc);
            // "this key does not count" as ONE 32-bit set per lane and tile: bit p <-> key 32t + 4g + p (p = (r&3) + 8(r>>2) is a compile-time
            // constant per register), = the padding-mask bits of the tile shifted by 4g, OR every key >= nkeys.  The per-element form
            // `key >= nkeys || (mask >> key) & 1` compiled to a 64-bit shift and two exec-mask regions per logit (round-6 probe: 3.4 us per
            // side of the score phase, 7000 cycles for 32 logits per lane); with the set it is an AND, a compare and a select.
            const unsigned chunk = (unsigned)(mbits[t >> 1] >> (32 * (t & 1)));       // wave-uniform
            const int lim = nkeys - 32 * t - 4 * g;                                  // keys of this lane with p >= lim are out of range
            const unsigned dead = (chunk >> (4 * g)) | (lim >= 32 ? 0u : (lim <= 0 ? 0xffffffffu : (0xffffffffu << lim)));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                constexpr unsigned one = 1u;
                const unsigned bit = one << ((r & 3) + 8 * (r >> 2));
                const float v = (dead & bit) ? -INFINITY : acc[r] * d.scale;
                sv[t][r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = xhalf_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t >= nt) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __expf(sv[t][r] - mx);
                sv[t][r] = e;
                sum += e;
            }
        }
        sum = xhalf_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t >= nt) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * g;
                // keys in [nkeys, npad) carry exp(-inf) = 0; keys >= npad go to the row's spare slot (the strides sw = Wp + 1 / sh = Hp + 4
                // leave index npad unused): an unconditional store instead of one exec-mask region per logit
                S[min(key, npad)] = sv[t][r] * inv;
            }
        }
        stamp(2 + side);
    }
    __syncthreads();   // all waves done with the key tiles (the V buffers overlay them); the LDS rows are visible to the whole wave
    stamp(4);

    {
        const int nq = min(QW, L - qbase);   // may be <= 0 for tail waves
        if (nq > 0 && save && d.a_row != nullptr) {
            save_rows(Srow, sm.sw, d.a_row + (((long)n * d.nh + head) * L + qbase) * Wp, nq, Wp, lane);
            save_rows(Scol, sm.sh, d.a_col + (((long)n * d.nh + head) * L + qbase) * Hp, nq, Hp, lane);
        }
    }
    stamp(5);
}

struct Fwd2Smem { int off_v, vts, total; };
__host__ __device__ inline Fwd2Smem fwd2_smem(const FwdSmem& sm, int H, int W, int KS, int RG = 1) {
    Fwd2Smem s;
    s.off_v = sm.off_k;
    s.vts = 32 * KS + 8;                                   // bf16 per V^T row: [hi 16KS | lo 16KS | pad 8] -> odd multiple of 16 bytes
    const int vfloats = 2 * RG * 32 * s.vts / 2;           // two buffers (RG = 2: two PAIRS of buffers) of 32 channel rows
    const int kfloats = (W + H) * KSTR;                    // key tiles of the score phase, padded rows (see rcda_scores_mfma)
    s.total = sm.off_k + (kfloats > vfloats ? kfloats : vfloats);
    return s;
}

// PROBE (tools/rcda_probe.py, CDETR_RCDA_PROBE=1; never dispatched otherwise): every wave writes 16 x uint64 of s_memtime readings into
// cdetr_rcda_fwd_desc.ws -- [kernel start, score phase done, main loop start, main loop end, output stored, cycles spent waiting at the
// per-key-row barrier, cycles in the MFMA + accumulate section, cycles in fetch + stash] -- so that "barrier cadence" is a number.
// RG = 2 (round 6): TWO key rows per workgroup barrier -- four LDS tile buffers (two pairs), the rows' MFMA chains T_h and T_h+1 are
// independent and interleave, half as many barriers.  The phase probe of the RG = 1 loop at the encoder shape (profiles/r6_rcda_probe.txt):
// per key row ~650 cycles in the MFMA + accumulate section (12 dependent MFMAs = 384 cycles of issue), ~390 in fetch + stash, ~370 waiting
// at the barrier.
template <int NW, int KS, int TH = 2, bool PROBE = false, int RG = 1>   // KS = 16-column steps of the key axis W (W <= 16 KS <= 96); TH = 32-row tiles of H in the MFMA score phase
__global__ __launch_bounds__(64 * NW) void rcda_fwd2_kernel(const cdetr_rcda_fwd_desc d) {
    unsigned long long pt[5] = {0, 0, 0, 0, 0}, pacc[3] = {0, 0, 0};
    auto now = [&]() __attribute__((always_inline)) -> unsigned long long {
        if constexpr (PROBE) { const unsigned long long t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return t; }
        return 0ull;
    };
    pt[0] = now();
    constexpr int NT = 64 * NW, QB = QW * NW;
    constexpr int PD = 4;                                  // V tiles in flight
    constexpr int NBLK = 16 * 4 * KS;                      // 4w x 2c blocks of one tile: (4 KS w-groups) x 16 channel pairs
    constexpr int VB = (NBLK + NT - 1) / NT;               // blocks per thread
    constexpr int VTS = 32 * KS + 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = d.H, W = d.W, L = d.L, E = d.nh * D;
    const FwdSmem sm = fwd_smem(H, W, NW);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i32 = lane & 31, g = lane >> 5;
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    xcd_slab(bx_, by_);
    const int n = by_ / d.nh, head = by_ % d.nh;
    const int qbase = bx_ * QB + wid * QW;
    const int q = qbase + i32;
    const bool qvalid = q < L;
    float* Srow = smem + sm.off_srow + wid * QW * sm.sw;
    float* Scol = smem + sm.off_scol + wid * QW * sm.sh;
    // gridDim.z > 1: the key rows are cut into gridDim.z slices, one workgroup each (the decoder's 2 x 8 x 300 queries are 48 workgroups
    // stepping through 50 barrier-separated iterations on a 256-CU chip; sliced four ways they are 192 stepping through 13).  Every slice
    // recomputes the two softmaxes (a few dozen MFMAs); slice 0 saves them.  The partial outputs meet in cdetr_rcda_fwd_desc.ws (below).
    const int hs = gridDim.z, hz = blockIdx.z;
    const int hper = (H + hs - 1) / hs;
    const int hb = hz * hper, he = min(H, hb + hper);
    // ---- V staging: block (wg, cp) = rows w0 = 4wg .. +3, channels 2cp, 2cp+1
    __bf16* VT = reinterpret_cast<__bf16*>(smem + sm.off_k);            // [2][32][VTS]
    // Loads are UNCONDITIONAL (a predicated load sits in its own basic block and the compiler then drains vmcnt every
    // iteration, which serialises the prefetch ring): rows w >= W are clamped to W-1 -- their products vanish because the
    // A_row operand is zero there -- and tiles h >= H re-read tile H-1 and are weighted with A_col = 0.
    const float* vsrc[VB];          // -> this thread's block of the NEXT tile to fetch
    long vrow[VB][4];               // clamped row offsets (floats) of the block's 4 key columns
    int vdst[VB];
#pragma unroll
    for (int b = 0; b < VB; ++b) {
        const int blk = tid + NT * b;
        const int wg = (blk < NBLK) ? (blk >> 4) : 0, cp = blk & 15;
        vdst[b] = (blk < NBLK) ? (2 * cp) * VTS + 4 * wg : -1;
        vsrc[b] = d.v + ((long)n * H + min(hb, H - 1)) * W * E + head * D + 2 * cp;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) vrow[b][kk] = (long)min(4 * wg + kk, W - 1) * E;
    }
    float rv[PD][VB][8];                                    // [kk] -> (x = channel 2cp, y = 2cp+1) as plain scalars
    const long tileE = (long)W * E;
    int hf = min(hb, H - 1);                                // next tile to fetch; vsrc[] points at it
    auto vfetch = [&](float (&r)[VB][8]) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < VB; ++b) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float2 t = *reinterpret_cast<const float2*>(vsrc[b] + vrow[b][kk]);
                r[b][2 * kk] = t.x;
                r[b][2 * kk + 1] = t.y;
            }
        }
        ++hf;
        if (hf < H) {               // wave-uniform
#pragma unroll
            for (int b = 0; b < VB; ++b) vsrc[b] += tileE;
        }
    };
    auto vstash = [&](const float (&r)[VB][8], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < VB; ++b) {
            if (vdst[b] < 0) continue;
            __bf16* dst = VT + buf * 32 * VTS + vdst[b];
            stash_split4(dst, 16 * KS, r[b][0], r[b][2], r[b][4], r[b][6]);
            stash_split4(dst + VTS, 16 * KS, r[b][1], r[b][3], r[b][5], r[b][7]);
        }
    };

    unsigned long long sst[6] = {0, 0, 0, 0, 0, 0};
    if (H <= 32 * TH) rcda_scores_mfma<NT, (KS + 1) / 2, TH, PROBE>(d, sm, smem, Srow, Scol, tid, lane, i32, g, n, head, qbase, q, qvalid, hz == 0, sst);
    else rcda_scores<NT>(d, sm, smem, Srow, Scol, tid, lane, i32, g, n, head, qbase, q, qvalid);
    pt[1] = now();

    // ---- hoisted B operand: this lane's A_row row, k-slot j of step s <-> w = 16s + 8g + j
    bf16x8 bh[KS], bl[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int w = 16 * s + 8 * g + j;
            x[j] = (w < W) ? Srow[i32 * sm.sw + w] : 0.f;
        }
        split_bf16x8(x, bh[s], bl[s]);
    }

    float outT[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) outT[r] = 0.f;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    vfetch(rv[0]); vfetch(rv[1]); vfetch(rv[2]); vfetch(rv[3]);
    vstash(rv[0], 0);          // the K tiles this overlays are dead: rcda_scores ended with a barrier
    if constexpr (RG == 2) vstash(rv[1], 1);
    __syncthreads();
    // one iteration; U = h mod PD is a compile-time constant so the register ring is statically indexed
    pt[2] = now();
    auto step = [&](auto U, int h) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        const unsigned long long s0 = now();
        vfetch(rv[u]);                                                   // tile h + PD; set u was stashed one iteration ago
        const unsigned long long s1 = now();
        const float acolh = (h < he) ? Scol[i32 * sm.sh + h] : 0.f;
        const __bf16* vt = VT + (u & 1) * 32 * VTS + i32 * VTS + 8 * g;
        f32x16 T = zero;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(vt + 16 * s);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(vt + 16 * KS + 16 * s);
            T = mfma_bf16x3(ah, al, bh[s], bl[s], T);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) outT[r] = fmaf(acolh, T[r], outT[r]);
        unsigned long long s2 = 0;
        if constexpr (PROBE) { asm volatile("" :: "v"(outT[0]), "v"(outT[15])); s2 = now(); }
        vstash(rv[(u + 1) % PD], (u + 1) & 1);                          // tile h+1 -> the buffer tile h-1 used
        const unsigned long long s3 = now();
        __syncthreads();
        if constexpr (PROBE) {
            const unsigned long long s4 = now();
            pacc[0] += s4 - s3; pacc[1] += s2 - s1; pacc[2] += (s1 - s0) + (s3 - s2);
        }
    };
    // RG == 2: pair P = (h / 2) mod 2 owns LDS buffers 2P, 2P + 1 and register sets 2P, 2P + 1
    auto step2 = [&](auto P_, int h) __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value, Q = 1 - P;
        const unsigned long long s0 = now();
        vfetch(rv[2 * P]);                                               // tiles h + 4, h + 5; the sets were stashed one iteration ago
        vfetch(rv[2 * P + 1]);
        const unsigned long long s1 = now();
        const float acol0 = (h < he) ? Scol[i32 * sm.sh + h] : 0.f;
        const float acol1 = (h + 1 < he) ? Scol[i32 * sm.sh + h + 1] : 0.f;
        const __bf16* vt0 = VT + (2 * P) * 32 * VTS + i32 * VTS + 8 * g;
        const __bf16* vt1 = vt0 + 32 * VTS;
        f32x16 T0 = zero, T1 = zero;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 ah0 = *reinterpret_cast<const bf16x8*>(vt0 + 16 * s);
            const bf16x8 al0 = *reinterpret_cast<const bf16x8*>(vt0 + 16 * KS + 16 * s);
            const bf16x8 ah1 = *reinterpret_cast<const bf16x8*>(vt1 + 16 * s);
            const bf16x8 al1 = *reinterpret_cast<const bf16x8*>(vt1 + 16 * KS + 16 * s);
            T0 = mfma_bf16x3(ah0, al0, bh[s], bl[s], T0);
            T1 = mfma_bf16x3(ah1, al1, bh[s], bl[s], T1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) outT[r] = fmaf(acol1, T1[r], fmaf(acol0, T0[r], outT[r]));
        unsigned long long s2 = 0;
        if constexpr (PROBE) { asm volatile("" :: "v"(outT[0]), "v"(outT[15])); s2 = now(); }
        vstash(rv[2 * Q], 2 * Q);                                        // tiles h + 2, h + 3 -> the pair tiles h - 2, h - 1 used
        vstash(rv[2 * Q + 1], 2 * Q + 1);
        const unsigned long long s3 = now();
        __syncthreads();
        if constexpr (PROBE) {
            const unsigned long long s4 = now();
            pacc[0] += s4 - s3; pacc[1] += s2 - s1; pacc[2] += (s1 - s0) + (s3 - s2);
        }
    };
    if constexpr (RG == 2) {
        for (int h0 = hb; h0 < he; h0 += 4) {
            step2(std::integral_constant<int, 0>{}, h0);
            step2(std::integral_constant<int, 1>{}, h0 + 2);
        }
    } else {
        for (int h0 = hb; h0 < he; h0 += PD) {
            step(std::integral_constant<int, 0>{}, h0);
            step(std::integral_constant<int, 1>{}, h0 + 1);
            step(std::integral_constant<int, 2>{}, h0 + 2);
            step(std::integral_constant<int, 3>{}, h0 + 3);
        }
    }
    pt[3] = now();
    if (hs > 1) {
        // The slices' partial sums meet in the scratch the split-reduction GEMMs use (same layout: arrival counters, then partials; same
        // protocol, igemm.hip): park, count in, and the slice that arrives last adds all of them IN SLICE ORDER -- the same sum whichever
        // slice that is.  Relaxed device-scope atomic stores / loads carry the data across the XCDs' private L2s without a fence.
        int* cnt = reinterpret_cast<int*>(d.ws);
        float* wsp = reinterpret_cast<float*>(cnt + RCDA_WS_COUNTERS);
        const int tile = blockIdx.y * gridDim.x + blockIdx.x;
        float* mine = wsp + ((long)tile * hs + hz) * 16 * NT + tid;
#pragma unroll
        for (int r = 0; r < 16; ++r) __hip_atomic_store(mine + r * NT, outT[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem + sm.off_k);          // the V tiles are dead
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (old == hs - 1) ? 1 : 0;
            if (last) __hip_atomic_store(cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // counters stay zero between launches
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0) return;
        float sum[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[r] = 0.f;
        for (int z = 0; z < hs; ++z) {
            if (z == hz) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sum[r] += outT[r];
            } else {
                const float* other = wsp + ((long)tile * hs + z) * 16 * NT + tid;
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = __hip_atomic_load(other + r * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int r = 0; r < 16; ++r) sum[r] += t[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) outT[r] = sum[r];
    }
    // out^T layout: lane = query i32, register r = channel (r&3) + 8*(r>>2) + 4g
    if (qvalid) {
        float* op = d.out + ((long)n * L + q) * E + head * D + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(op + 8 * j) = make_float4(outT[4 * j], outT[4 * j + 1], outT[4 * j + 2], outT[4 * j + 3]);
    }
    if constexpr (PROBE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pt[4] = now();
        if (lane == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(d.ws) + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wid) * 16;
            o[0] = pt[0]; o[1] = pt[1]; o[2] = pt[2]; o[3] = pt[3]; o[4] = pt[4]; o[5] = pacc[0]; o[6] = pacc[1]; o[7] = pacc[2];
#pragma unroll
            for (int i = 0; i < 6; ++i) o[8 + i] = sst[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward (dS)
// LDS plan of the dS kernel (56 KB at 50x50, 4 waves -> two workgroups per CU):
//   U: per-wave [32][su] slices holding A_col at the start and the ds_col / A_row staging at the end; while the main loop
//      runs (A_col lives in registers) the same memory is the double-buffered V tile shared by the workgroup;
//   R: per-wave [32][sw] A_row; element (q, w) is overwritten by dA_row[q, w] as soon as column w has been consumed.
// A/B knob of the fused key gradients (rcda_bwd_body): 1 = every wave adds its partial tiles to global itself (rounds 3-5)
__device__ __constant__ int g_rcda_dk_per_wave = 0;
__device__ __forceinline__ bool rcda_dk_per_wave() { return g_rcda_dk_per_wave != 0; }
constexpr int BWD_CG = 2;     // key columns staged and consumed per workgroup barrier of the dS kernel
struct BwdSmem {
    int sw, sh, su, off_u, off_r, off_kk, total;
};
__host__ __device__ inline BwdSmem bwd_smem(int H, int W, int NF, int NW, bool with_k = false) {
    BwdSmem s;
    const int Wp = (W + 3) & ~3, Hp = (H + 7) & ~7;
    s.sw = Wp + 1;
    s.sh = Hp + 1;
    s.su = s.sw > s.sh ? s.sw : s.sh;
    s.off_u = 0;
    int u = NW * QW * s.su;
    const int v = 2 * BWD_CG * (32 * NF) * 36;
    if (v > u) u = v;
    s.off_r = (u + 3) & ~3;
    s.total = s.off_r + NW * QW * s.sw;
    s.off_kk = (s.total + 3) & ~3;                  // projected keys of (n, head) for the fused query gradients: [W][32] | [H][32]
    if (with_k) s.total = s.off_kk + (W + H) * D;
    return s;
}

// TERMS: bf16 MFMAs per product in the split-bf16 paths (3, or 1 = plain bf16: cdetr_rcda_bwd_desc.precision 3)
// PROBE (tools/rcda_probe.py bwd, CDETR_RCDA_PROBE; never dispatched otherwise): cdetr_rcda_bwd_desc.ds_row is the stamp buffer (16 x uint64 per
// wave of s_memtime readings: start | attention rows staged, A_col in registers | main loop over the key columns done | both softmax backward
// passes | keys in LDS | query gradients stored | key gradients added), nothing is saved through ds_row / ds_col.
template <int NF, int NW, int PREC, int TERMS = 3, bool PROBE = false>
__device__ __forceinline__ void rcda_bwd_body(const cdetr_rcda_bwd_desc& d, const int bx, const int by) {
    unsigned long long pst[7] = {0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int i) __attribute__((always_inline)) {
        if constexpr (PROBE) { pst[i] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    };
    auto flush_probe = [&]() __attribute__((always_inline)) {
        if constexpr (PROBE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if ((threadIdx.x & 63) == 0) {
                unsigned long long* o = reinterpret_cast<unsigned long long*>(d.ds_row) + ((long)(by * gridDim.x + bx) * NW + (threadIdx.x >> 6)) * 16;
#pragma unroll
                for (int i = 0; i < 7; ++i) o[i] = pst[i];
            }
        }
    };
    stamp(0);
    constexpr int NT = 64 * NW, QB = QW * NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int VS = 36;                 // V tile row stride (floats): 36/4 odd -> conflict-free ds_read_b128
    constexpr int HR = 32 * NF;            // V tile rows (zero beyond H)
    const int H = d.H, W = d.W, L = d.L, E = d.nh * D;
    const int Wp = (W + 3) & ~3, Hp = (H + 7) & ~7;
    const BwdSmem sm = bwd_smem(H, W, NF, NW, d.dq_row != nullptr);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i32 = lane & 31, g = lane >> 5;
    const int n = by / d.nh, head = by % d.nh;
    const int qbase = bx * QB + wid * QW;
    const int q = qbase + i32;
    const bool qvalid = q < L;
    const int nq = min(QW, L - qbase);

    // projected keys of (n, head) for the fused query gradients: requested now, parked in registers through the main loop and written
    // to LDS at the end (no exposed round trip there); larger key sets are staged late
    constexpr int KP = 4;
    const int nk8 = (W + H) * 8;
    const bool kpre_ok = d.dq_row != nullptr && nk8 <= KP * NT;
    float4 kpre[KP];
#pragma unroll
    for (int u = 0; u < KP; ++u) kpre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kpre_ok) {
#pragma unroll
        for (int u = 0; u < KP; ++u) {
            const int idx = min(tid + NT * u, nk8 - 1);
            const int key = idx >> 3, c4 = idx & 7;
            const float* src = (key < W) ? d.k_row + ((long)n * W + key) * E + head * D + c4 * 4
                                         : d.k_col + ((long)n * H + (key - W)) * E + head * D + c4 * 4;
            kpre[u] = ld4(src);
        }
    }

    float* Acol = smem + sm.off_u + wid * QW * sm.su;        // this wave's U slice, [32][sh] view
    float* Arow = smem + sm.off_r + wid * QW * sm.sw;        // [32][sw]; turns into dA_row column by column
    float* dArow = Arow;
    float* Vs = smem + sm.off_u;                             // [2][CG][HR][36], overlays U during the main loop

    // dOut^T fragment (B operand, loop invariant): lane (j = query, g) holds dOut[q][8kk + 4g + s]; requested before the attention rows so that
    // the two fetches share one round trip
    float dob[4][4];
    float4 dobt[4];
    {
        const float* dp = d.d_out + ((long)n * L + (qvalid ? q : 0)) * E + head * D;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) dobt[kk] = ld4(dp + kk * 8 + g * 4);
    }
    // ---- load the saved attention rows of this wave (coalesced), zero for tail queries
    // A_col goes straight into the transposed register layout it is used in (element (f, r) <-> key row h = 32f + (r&3) + 8(r>>2) + 4g of query
    // i32: four consecutive h = one 16-byte load); rounds 1-5 staged it through LDS (64 scalar stores + 32 scalar reads per lane + two barriers)
    float4 act[NF][4];
    {
        const float* ac = d.a_col + (((long)n * d.nh + head) * L + (qvalid ? q : 0)) * Hp;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) act[f][j] = ld4(ac + min(32 * f + 8 * j + 4 * g, Hp - 4));
    }
    {
        const int qb = min(qbase, L - 1), nqv = max(nq, 0);      // tail waves (qbase >= L) stage zeros from an in-bounds address
        stage_rows<8>(d.a_row + (((long)n * d.nh + head) * L + qb) * Wp, nqv, Wp, Arow, sm.sw, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        if (!qvalid) dobt[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
        dob[kk][0] = dobt[kk].x; dob[kk][1] = dobt[kk].y; dob[kk][2] = dobt[kk].z; dob[kk][3] = dobt[kk].w;
    }
    bf16x8 dobh[2], dobl[2];      // loop-invariant split of the dOut^T fragment (split-bf16 mode)
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
        const float x[8] = {dob[2 * kp][0], dob[2 * kp][1], dob[2 * kp][2], dob[2 * kp][3],
                            dob[2 * kp + 1][0], dob[2 * kp + 1][1], dob[2 * kp + 1][2], dob[2 * kp + 1][3]};
        split_bf16x8(x, dobh[kp], dobl[kp]);
    }
    // A_col in the transposed accumulator layout: element (f, r) <-> key row h = 32f + (r&3) + 8(r>>2) + 4g, query i32
    float acolT[NF][16], dacolT[NF][16];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = qvalid && (32 * f + 8 * j + 4 * g < Hp);
            acolT[f][4 * j + 0] = ok ? act[f][j].x : 0.f;
            acolT[f][4 * j + 1] = ok ? act[f][j].y : 0.f;
            acolT[f][4 * j + 2] = ok ? act[f][j].z : 0.f;
            acolT[f][4 * j + 3] = ok ? act[f][j].w : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) dacolT[f][4 * j + k] = 0.f;
        }
    // (no workgroup barrier here: the A_row slices are private to their waves, nobody reads the U region before the first V tile is staged,
    // and the barrier that follows that staging orders the rest)
    stamp(1);

    // ---- main loop over key columns w.  V[:, w, :] tiles run through a ring of PD register sets (unconditional loads: rows
    // h >= H re-read row H-1 -- A_col is zero there -- and columns w >= W re-read column W-1, weighted by A_row = 0) and are
    // staged once per workgroup; in split-bf16 mode the tile is split while it is staged, in the slot order the dOut^T
    // fragment uses (k-slot j of lane group g at step kp <-> channel 16kp + 8(j>>2) + 4g + (j&3)).
    constexpr int PD = 2;      // register sets in flight
    constexpr int CG = BWD_CG; // key columns per set = per workgroup barrier
    constexpr int VSLOTS = (NF * 256 + NT - 1) / NT;
    const float* vsrc[VSLOTS];
    int vdst[VSLOTS];
#pragma unroll
    for (int s = 0; s < VSLOTS; ++s) {
        const int idx = tid + NT * s;
        const int h = idx >> 3, c4 = idx & 7;
        vsrc[s] = d.v + (long)n * H * W * E + head * D + (long)min(h, H - 1) * W * E + c4 * 4;
        if (PREC == 1) {
            const int kp = c4 >> 2, qd = c4 & 3;
            vdst[s] = (h < HR) ? h * (2 * VS) + kp * 16 + (qd & 1) * 8 + (qd >> 1) * 4 : -1;     // bf16 units
        } else {
            vdst[s] = (h < HR) ? h * VS + c4 * 4 : -1;                                            // float units
        }
    }
    float4 rv[PD][CG][VSLOTS];
    int wf = 0;                                             // next column to fetch; vsrc[] points at it
    auto vfetch = [&](float4 (&r)[CG][VSLOTS]) __attribute__((always_inline)) {
#pragma unroll
        for (int cc = 0; cc < CG; ++cc) {
#pragma unroll
            for (int s = 0; s < VSLOTS; ++s) r[cc][s] = ld4(vsrc[s]);
            ++wf;
            if (wf < W) {
#pragma unroll
                for (int s = 0; s < VSLOTS; ++s) vsrc[s] += E;
            }
        }
    };
    auto vstash = [&](const float4 (&r)[CG][VSLOTS], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int cc = 0; cc < CG; ++cc)
#pragma unroll
            for (int s = 0; s < VSLOTS; ++s) {
                if (vdst[s] < 0) continue;
                float* tile = Vs + (buf * CG + cc) * HR * VS;
                if (PREC == 1) stash_split4(reinterpret_cast<__bf16*>(tile) + vdst[s], 32, r[cc][s].x, r[cc][s].y, r[cc][s].z, r[cc][s].w);
                else *reinterpret_cast<float4*>(tile + vdst[s]) = r[cc][s];
            }
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    vfetch(rv[0]); vfetch(rv[1]);
    vstash(rv[0], 0);
    __syncthreads();
    auto step = [&](auto U, int w0s) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        const int buf = u & 1;
        vfetch(rv[u]);                                                   // the set after next; set u was staged one step ago
#if CDETR_RCDA_UNROLL_CC
#pragma unroll
#else
#pragma clang loop unroll(disable)
#endif
      for (int cc = 0; cc < CG; ++cc) {       // rolled on purpose: unrolled, the compiler keeps CG accumulator sets live (380 VGPRs in round 2; round 6, -DCDETR_RCDA_UNROLL_CC=1:
                                              // 256 VGPRs + 136 bytes of scratch in the 5-wave kernel, 84 bytes without the parked keys -- two waves share a SIMD there)
        const int w = w0s + cc;
        const float* tile = Vs + (buf * CG + cc) * HR * VS;
        const float arow = Arow[i32 * sm.sw + w];                        // zero for W <= w < Wp
        float part4[4] = {0.f, 0.f, 0.f, 0.f};      // four partial sums: one accumulator is a chain of 16 NF dependent FMAs per key column
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            f32x16 gt = zero16;
            if (PREC == 0) {
                const float* va = tile + (32 * f + i32) * VS + g * 4;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const float4 a = *reinterpret_cast<const float4*>(va + kk * 8);
                    gt = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, dob[kk][0], gt, 0, 0, 0);
                    gt = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, dob[kk][1], gt, 0, 0, 0);
                    gt = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, dob[kk][2], gt, 0, 0, 0);
                    gt = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, dob[kk][3], gt, 0, 0, 0);
                }
            } else {
                const __bf16* va = reinterpret_cast<const __bf16*>(tile) + (32 * f + i32) * (2 * VS) + g * 8;
#pragma unroll
                for (int kp = 0; kp < 2; ++kp) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(va + kp * 16);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(va + 32 + kp * 16);
                    gt = mfma_bf16_terms<TERMS>(ah, al, dobh[kp], dobl[kp], gt);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                part4[r & 3] = fmaf(acolT[f][r], gt[r], part4[r & 3]);
                dacolT[f][r] = fmaf(arow, gt[r], dacolT[f][r]);
            }
        }
        float part = (part4[0] + part4[1]) + (part4[2] + part4[3]);
        part = xhalf_sum(part);
        if (g == 0) dArow[i32 * sm.sw + w] = part;
      }
        vstash(rv[(u + 1) % PD], buf ^ 1);
        __syncthreads();
    };
    for (int w0 = 0; w0 < W; w0 += PD * CG) {     // w runs to Wp - 1 at most (Wp % 4 == 0): Arow / dArow rows hold Wp (+1) entries
        step(std::integral_constant<int, 0>{}, w0);
        step(std::integral_constant<int, 1>{}, w0 + CG);
    }

    stamp(2);
    // The loop ended with a barrier: the V buffers are dead and every wave owns its U slice again.
    auto wave_sync = [&]() {     // same-wave LDS hand-off: order the writes above before the reads below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // ---- softmax backward, row attention: A_row comes back from HBM (coalesced, L2-hot) into the U slice
    {
        float* Ar = Acol;          // [32][sw] view of the slice
        stage_rows<8>(d.a_row + (((long)n * d.nh + head) * L + min(qbase, L - 1)) * Wp, max(nq, 0), Wp, Ar, sm.sw, lane);
        wave_sync();
        float dot = 0.f;
#pragma unroll 8
        for (int w = g; w < W; w += 2) dot = fmaf(Ar[i32 * sm.sw + w], dArow[i32 * sm.sw + w], dot);      // (unrolled: the LDS reads of 8 steps fly together)
        dot = xhalf_sum(dot);
        for (int w0 = g; w0 < Wp; w0 += 16) {        // batches of 8: all reads of a batch before its writes (the two tiles may alias as far as the compiler knows)
            float a8[8], b8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int wc = min(w0 + 2 * j, Wp - 1);
                a8[j] = Ar[i32 * sm.sw + wc];
                b8[j] = dArow[i32 * sm.sw + wc];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int w = w0 + 2 * j;
                dArow[i32 * sm.sw + min(w, Wp)] = (w < W) ? d.scale * a8[j] * (b8[j] - dot) : 0.f;      // w >= Wp: the row's spare slot (sw = Wp + 1)
            }
        }
        wave_sync();
        if (!PROBE && nq > 0 && d.ds_row) save_rows(dArow, sm.sw, d.ds_row + (((long)n * d.nh + head) * L + qbase) * Wp, nq, Wp, lane);
        wave_sync();               // the slice is rewritten below
    }
    // ---- softmax backward, column attention (registers), staged through the U slice for a coalesced store
    {
        float dot = 0.f;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) dot = fmaf(acolT[f][r], dacolT[f][r], dot);
        dot = xhalf_sum(dot);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int h = 32 * f + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (h < Hp) Acol[i32 * sm.sh + h] = d.scale * acolT[f][r] * (dacolT[f][r] - dot);
            }
        wave_sync();
        if (!PROBE && nq > 0 && d.ds_col) save_rows(Acol, sm.sh, d.ds_col + (((long)n * d.nh + head) * L + qbase) * Hp, nq, Hp, lane);
    }
    stamp(3);
    // ---- fused query gradients: dq_row[q][:] = sum_w dS_row[q][w] k_row[w][:], dq_col likewise.  dS_row / dS_col are still in this
    // wave's LDS tiles; the projected keys of (n, head) are staged once per workgroup.  Lane (query i32, half g) owns channels 16g..16g+15.
    // the projected queries of the fused key gradients (used after the query gradients below): requested now, so that their round trip hides
    // behind the key staging and the dq products
    float qv[2][2][8];                                    // [side][step][j]: q[qbase + 16 step + 8g + j][head*32 + i32]
    if (PREC == 1 && d.dq_row && d.dk_row) {
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int qq = qbase + 16 * st + 8 * g + j;
                    qv[side][st][j] = (side == 0 ? d.q_row : d.q_col)[((long)n * L + min(max(qq, 0), L - 1)) * E + head * D + i32];
                }
    }
    if (d.dq_row) {
        float* Kk = smem + sm.off_kk;
        if (kpre_ok) {
#pragma unroll
            for (int u = 0; u < KP; ++u) {
                const int idx = tid + NT * u;
                if (idx < nk8) *reinterpret_cast<float4*>(Kk + (idx >> 3) * D + (idx & 7) * 4) = kpre[u];
            }
        } else
        for (int base = 0; base < nk8; base += 4 * NT) {       // batches of 4 unconditional loads per thread (clamped key, masked store)
            float4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = min(base + tid + NT * u, nk8 - 1);
                const int key = idx >> 3, c4 = idx & 7;
                const float* src = (key < W) ? d.k_row + ((long)n * W + key) * E + head * D + c4 * 4
                                             : d.k_col + ((long)n * H + (key - W)) * E + head * D + c4 * 4;
                t[u] = ld4(src);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + tid + NT * u;
                if (idx < nk8) *reinterpret_cast<float4*>(Kk + (idx >> 3) * D + (idx & 7) * 4) = t[u];
            }
        }
        __syncthreads();
        stamp(4);
        if constexpr (PREC == 1) {
            // dq^T[c][q] = sum_w k^T[c][w] dS^T[w][q] on the matrix pipe: A = k^T (row = channel i32, k-slot j of step s <-> key 16s + 8g + j),
            // B = dS^T (column = query i32, same key slots, read from this lane's own LDS row); keys past the end are masked to zero
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int nkeys = side == 0 ? W : H;
                const float* S = side == 0 ? dArow + i32 * sm.sw : Acol + i32 * sm.sh;
                const float* Ks = Kk + (side == 0 ? 0 : W * D) + i32;
                f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const int nks = (nkeys + 15) >> 4;
                for (int st = 0; st < nks; ++st) {
                    float a[8], b[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int w = 16 * st + 8 * g + j;
                        const int wc = min(w, nkeys - 1);
                        const float av = Ks[wc * D], bv = S[wc];
                        a[j] = (w < nkeys) ? av : 0.f;
                        b[j] = (w < nkeys) ? bv : 0.f;
                    }
                    bf16x8 ah, al, bh, bl;
                    split_bf16x8(a, ah, al);
                    split_bf16x8(b, bh, bl);
                    acc = mfma_bf16_terms<TERMS>(ah, al, bh, bl, acc);
                }
                mfma_drain(acc);
                if (qvalid) {      // lane = query i32; register r = channel (r & 3) + 8 (r >> 2) + 4 g
                    float* dst = (side == 0 ? d.dq_row : d.dq_col) + ((long)n * L + q) * E + head * D + 4 * g;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<float4*>(dst + 8 * j) = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
                }
            }
            stamp(5);
            if (d.dk_row) {
                // ---- fused key gradients: dk[w][c] += sum_q dS[q][w] q[q][c] over this workgroup's queries.  Per wave: A = dS^T (row = key,
                // k-slot j of step s <-> query 16s + 8g + j, column reads of the wave's LDS tile), B = the projected queries straight from
                // global (lane = channel i32: 128-byte coalesced rows); the waves' partial [keys][32] tiles are summed through LDS (plain stores + one
                // pass of reads: ds_add_f32 ran at a few hundred cycles per wave instruction when that was tried) and added to global once per workgroup.
                // Round 6: the waves' partial [keys][32] tiles meet in LDS (each wave parks its two tiles in its own, by then dead, dS slices) and
                // the workgroup adds ONE tile per side to global: NW x fewer atomics (the probe put 11 us of the encoder-shape workgroup's 55 us
                // into this phase: 80 waves per (image, head) adding to the same 6400 addresses).  Maps with more than 128 keys per side keep the
                // per-wave atomics.
                const bool wg_reduce = (W <= 128 && H <= 128) && !rcda_dk_per_wave();
                constexpr int NTL = 4;
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const int nkeys = side == 0 ? W : H;
                    const float* S = side == 0 ? dArow : Acol;        // [32 queries][stride]
                    float* Sw = side == 0 ? dArow : Acol;
                    const int sstr = side == 0 ? sm.sw : sm.sh;
                    float* acc_g = (side == 0 ? d.dk_row + (long)n * W * E : d.dk_col + (long)n * H * E) + head * D + i32;
                    bf16x8 bh[2], bl[2];
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        float b[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) b[j] = (qbase + 16 * st + 8 * g + j < L) ? qv[side][st][j] : 0.f;
                        split_bf16x8(b, bh[st], bl[st]);
                    }
                    const int ntile = (nkeys + 31) >> 5;
                    auto tile = [&](int tl) __attribute__((always_inline)) -> f32x16 {
                        const int wkey = 32 * tl + i32, wc = min(wkey, nkeys - 1);
                        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int st = 0; st < 2; ++st) {
                            float a[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const float av = S[(16 * st + 8 * g + j) * sstr + wc];      // rows of tail queries hold zeros
                                a[j] = (wkey < nkeys) ? av : 0.f;
                            }
                            bf16x8 ah, al;
                            split_bf16x8(a, ah, al);
                            acc = mfma_bf16_terms<TERMS>(ah, al, bh[st], bl[st], acc);
                        }
                        mfma_drain(acc);
                        return acc;
                    };
                    if (wg_reduce) {
                        f32x16 accs[NTL];
#pragma unroll
                        for (int tl = 0; tl < NTL; ++tl)
                            if (tl < ntile) accs[tl] = tile(tl);
                        wave_sync();                                   // every read of this wave's dS slice is done: it becomes the partial tile [key][32]
#pragma unroll
                        for (int tl = 0; tl < NTL; ++tl)
                            if (tl < ntile) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int wk = 32 * tl + (r & 3) + 8 * (r >> 2) + 4 * g;
                                    Sw[min(wk, nkeys) * D + i32] = accs[tl][r];      // keys past the end: row `nkeys` of the slice (never read; QW (stride) >= (nkeys + 1) 32)
                                }
                            }
                    } else {
                        for (int tl = 0; tl < ntile; ++tl) {
                            const f32x16 acc = tile(tl);
                            // accumulator: row = key 32 tl + (r & 3) + 8 (r >> 2) + 4 g, column = channel i32
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int wk = 32 * tl + (r & 3) + 8 * (r >> 2) + 4 * g;
                                if (wk < nkeys) atomicAdd(acc_g + (long)wk * E, acc[r]);     // 128-byte rows per half-wave
                            }
                        }
                    }
                }
                if (wg_reduce) {
                    __syncthreads();
                    const int nrow = W * D, ntot = (W + H) * D;
                    for (int e = tid; e < ntot; e += NT) {
                        const bool rowside = e < nrow;
                        const int idx = rowside ? e : e - nrow;
                        const float* base = rowside ? smem + sm.off_r + idx : smem + sm.off_u + idx;
                        const int wstr = rowside ? QW * sm.sw : QW * sm.su;
                        float sum = 0.f;
#pragma unroll
                        for (int wv = 0; wv < NW; ++wv) sum += base[wv * wstr];
                        const int key = idx >> 5, ch = idx & 31;
                        float* dst = rowside ? d.dk_row + ((long)n * W + key) * E + head * D + ch : d.dk_col + ((long)n * H + key) * E + head * D + ch;
                        atomicAdd(dst, sum);
                    }
                }
            }
            stamp(6);
            flush_probe();
            return;
        }
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int nkeys = side == 0 ? W : H;
            const float* S = side == 0 ? dArow + i32 * sm.sw : Acol + i32 * sm.sh;
            const float* Ks = Kk + (side == 0 ? 0 : W * D) + 16 * g;
            float acc[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll 2
            for (int k = 0; k < nkeys; ++k) {
                const float a = S[k];
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const float4 kv = *reinterpret_cast<const float4*>(Ks + k * D + c4 * 4);
                    acc[c4 * 4 + 0] = fmaf(a, kv.x, acc[c4 * 4 + 0]);
                    acc[c4 * 4 + 1] = fmaf(a, kv.y, acc[c4 * 4 + 1]);
                    acc[c4 * 4 + 2] = fmaf(a, kv.z, acc[c4 * 4 + 2]);
                    acc[c4 * 4 + 3] = fmaf(a, kv.w, acc[c4 * 4 + 3]);
                }
            }
            if (qvalid) {
                float* dst = (side == 0 ? d.dq_row : d.dq_col) + ((long)n * L + q) * E + head * D + 16 * g;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4)
                    *reinterpret_cast<float4*>(dst + c4 * 4) = make_float4(acc[c4 * 4], acc[c4 * 4 + 1], acc[c4 * 4 + 2], acc[c4 * 4 + 3]);
            }
        }
    }
}

template <int NF, int NW, int PREC, int TERMS = 3, bool PROBE = false>
__global__ __launch_bounds__(64 * NW) void rcda_bwd_kernel(const cdetr_rcda_bwd_desc d) {
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    xcd_slab(bx_, by_);
    rcda_bwd_body<NF, NW, PREC, TERMS, PROBE>(d, bx_, by_);
}

// ------------------------------------------------------------------------------------------------ backward (dV)
// dV[h,w,c] += sum_q A_col[q,h] A_row[q,w] dOut[q,c].  Workgroup = (4 consecutive w (one per wave), (n,head), q-slice).
template <int NF, int PREC>
__global__ __launch_bounds__(256) void rcda_dv_kernel(const cdetr_rcda_bwd_desc d, const int q_per_slice) {
    constexpr int QT = 64;               // queries per LDS tile
    constexpr int HR = 32 * NF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = d.H, W = d.W, L = d.L, E = d.nh * D;
    const int Wp = (W + 3) & ~3, Hp = (H + 7) & ~7;
    float* Ac = smem;                    // [QT][HR]   (zero for h >= Hp)
    float* Ar = Ac + QT * HR;            // [QT][Wp]
    float* Do = Ar + QT * Wp;            // [QT][32]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i32 = lane & 31, g = lane >> 5;
    const int n = blockIdx.y / d.nh, head = blockIdx.y % d.nh;
    const int w = blockIdx.x * 4 + wid;
    const bool wvalid = w < W;
    const int qs = blockIdx.z * q_per_slice;
    const int qe = min(L, qs + q_per_slice);
    const float* gac = d.a_col + ((long)n * d.nh + head) * L * Hp;
    const float* gar = d.a_row + ((long)n * d.nh + head) * L * Wp;
    const float* gdo = d.d_out + (long)n * L * E + head * D;

    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    for (int q0 = qs; q0 < qe; q0 += QT) {
        const int nq = min(QT, qe - q0);
        __syncthreads();
        for (int idx = tid; idx < QT * HR; idx += 256) {
            const int r = idx / HR, c = idx - r * HR;
            Ac[idx] = (r < nq && c < Hp) ? gac[(long)(q0 + r) * Hp + c] : 0.f;
        }
        for (int idx = tid; idx < QT * Wp; idx += 256) {
            const int r = idx / Wp;
            Ar[idx] = (r < nq) ? gar[(long)q0 * Wp + idx] : 0.f;
        }
        for (int idx = tid; idx < QT * 8; idx += 256) {
            const int r = idx >> 3, c4 = idx & 7;
            *reinterpret_cast<float4*>(Do + r * D + c4 * 4) =
                (r < nq) ? ld4(gdo + (long)(q0 + r) * E + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        if (wvalid && PREC == 0) {
#pragma unroll 2
            for (int kk = 0; kk < QT / 8; ++kk) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int qq = kk * 8 + g * 4 + s;
                    const float ar = Ar[qq * Wp + w];
                    const float b = Do[qq * D + i32];
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const float a = Ac[qq * HR + 32 * f + i32] * ar;
                        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[f], 0, 0, 0);
                    }
                }
            }
        }
        if (wvalid && PREC == 1) {
#pragma unroll 2
            for (int kp = 0; kp < QT / 16; ++kp) {
                float arv[8], bv[8];
                int qs[8];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    qs[t] = (2 * kp) * 8 + g * 4 + t;
                    qs[4 + t] = (2 * kp + 1) * 8 + g * 4 + t;
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    arv[t] = Ar[qs[t] * Wp + w];
                    bv[t] = Do[qs[t] * D + i32];
                }
                bf16x8 bh, bl;
                split_bf16x8(bv, bh, bl);
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    float av[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) av[t] = Ac[qs[t] * HR + 32 * f + i32] * arv[t];
                    bf16x8 ah, al;
                    split_bf16x8(av, ah, al);
                    acc[f] = mfma_bf16x3(ah, al, bh, bl, acc[f]);
                }
            }
        }
    }
    mfma_drain(acc);
    if (!wvalid) return;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int h = 32 * f + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (h < H) atomicAdd(d.d_v + (((long)n * H + h) * W + w) * E + head * D + i32, acc[f][r]);
        }
}

// ------------------------------------------------------------------------------------------------ backward (dV), two-step form
// split-bf16 only, any W: chunks of <= 64 key columns.   dV[h,w,c] = sum_q A_row[q,w] * (A_col[q,h] dOut[q,c])
// Workgroup = 8 waves = 8 key rows h (one per wave) of one (n, head) over a slice of the queries; the reduction runs over q:
//   A operand  A_row^T[w, q]  (rows w: two 32-row fragments) -- the same for every h: transposed and split into bf16 hi/lo
//              ONCE per workgroup while the q-tile is staged (4q x 4w register blocks -> ds_write_b64);
//   B operand  A_col[q,h] * dOut[q,c]  (lane = channel, 8 consecutive q): 8 multiplies + one packed split per k-step, fed
//              from the fp32 tiles dOut^T[c, q] and A_col^T[h, q] with four ds_read_b128.
// One k-step = 6 MFMAs against ~30 VALU instructions: matrix-pipe bound, where the A_col*A_row-operand form of
// rcda_dv_kernel needs ~76 VALU per 6 MFMAs and reloads every tile synchronously.  Every thread has at most one staging
// role (a_row block / dOut block / a_col quad); q-tiles run through two register sets with unconditional loads.
struct Dv2Smem { int art, dot, act, tile, total; };
__host__ __device__ inline Dv2Smem dv2_smem() {
    Dv2Smem s;
    s.art = 0;                       // A_row^T  [64 w][hi 64 | lo 64 | pad 8] bf16  = 64 * 68 floats
    s.dot = 64 * 68;                 // dOut^T   [32 c][64 q + 4] fp32
    s.act = s.dot + 32 * 68;         // A_col^T  [8 h][64 q + 4] fp32
    s.tile = s.act + 8 * 68;
    s.total = 2 * s.tile;
    return s;
}

template <int TERMS = 3>
__device__ __forceinline__ void rcda_dv2_body(const cdetr_rcda_bwd_desc& d, const int q_per_slice, const int bx, const int by, const int bz) {
    constexpr int QT = 64, ARS = 136;                   // queries per tile; bf16 per A_row^T row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Dv2Smem sm = dv2_smem();
    const int H = d.H, W = d.W, L = d.L, E = d.nh * D;
    const int Wp = (W + 3) & ~3, Hp = (H + 7) & ~7;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i32 = lane & 31, g = lane >> 5;
    const int n = by / d.nh, head = by % d.nh;
    // bx = (key-column chunk, key-row group): maps wider than 64 columns (round 6: FSCD-LVIS images up to 1333 wide -> W = 84) are cut
    // into ceil(Wp / 64) chunks of equal width (a multiple of 4 columns), one workgroup each; a chunk is the 64-row A_row^T tile below
    const int nhg = (H + 7) >> 3;
    const int wchunk = bx / nhg;
    const int nwc = (Wp + 63) >> 6;
    const int wper = (((Wp + nwc - 1) / nwc) + 3) & ~3;
    const int wbase = wchunk * wper;
    const int Wc = min(wper, Wp - wbase);               // columns of this chunk (<= 64, % 4 == 0)
    const int h0 = (bx - wchunk * nhg) * 8, h = h0 + wid;
    const int qs = bz * q_per_slice;
    const int qe = min(L, qs + q_per_slice);
    const int ntile = (qe - qs + QT - 1) / QT;
    if (ntile <= 0) return;

    // ---- staging roles
    const int nrb = 16 * (Wc >> 2);                     // a_row blocks: 16 q-groups x Wc/4 w-quads (<= 256)
    int role, q4, x4;                                   // q4 = q-group (4 queries), x4 = w-quad / c-quad / h-quad
    if (tid < nrb) { role = 0; q4 = tid & 15; x4 = tid >> 4; }
    else if (tid >= 256 && tid < 384) { role = 1; q4 = (tid - 256) & 15; x4 = (tid - 256) >> 4; }
    else if (tid >= 384 && tid < 416) { role = 2; q4 = (tid - 384) & 15; x4 = (tid - 384) >> 4; }     // 2 h-quads
    else { role = 3; q4 = 0; x4 = 0; }
    const float* src;                                   // row (qs + 4 q4), this thread's quad; advanced by one tile per fetch
    long rstride;
    if (role == 0) { src = d.a_row + (((long)n * d.nh + head) * L) * Wp + wbase + x4 * 4; rstride = Wp; }
    else if (role == 1) { src = d.d_out + (long)n * L * E + head * D + x4 * 4; rstride = E; }
    else { src = d.a_col + (((long)n * d.nh + head) * L) * Hp + min(h0 + x4 * 4, Hp - 4); rstride = Hp; }
    float4 rs[2][4];
    unsigned rmask[2] = {0, 0};                         // bit kk: query row kk of the block is inside the slice
    int tf = 0;                                         // next tile to fetch
    auto fetch = [&](float4 (&r)[4], unsigned& m) __attribute__((always_inline)) {
        const int qb = qs + min(tf, ntile - 1) * QT + q4 * 4;        // surplus fetches re-read the last tile (never staged)
        m = 0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int q = qb + kk;
            r[kk] = ld4(src + (long)min(q, L - 1) * rstride);
            m |= (q < qe ? 1u : 0u) << kk;
        }
        ++tf;
    };
    auto stash = [&](const float4 (&r0)[4], unsigned m, int buf) __attribute__((always_inline)) {
        float4 r[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) r[kk] = ((m >> kk) & 1u) ? r0[kk] : make_float4(0.f, 0.f, 0.f, 0.f);
        float* base = smem + buf * sm.tile;
        if (role == 0) {            // 4q x 4w -> four rows w, 4 consecutive q each, split
            __bf16* dst = reinterpret_cast<__bf16*>(base + sm.art) + (x4 * 4) * ARS + q4 * 4;
            stash_split4(dst, 64, r[0].x, r[1].x, r[2].x, r[3].x);
            stash_split4(dst + ARS, 64, r[0].y, r[1].y, r[2].y, r[3].y);
            stash_split4(dst + 2 * ARS, 64, r[0].z, r[1].z, r[2].z, r[3].z);
            stash_split4(dst + 3 * ARS, 64, r[0].w, r[1].w, r[2].w, r[3].w);
        } else if (role == 1 || role == 2) {   // 4q x 4c (or 4h) -> four rows, fp32
            float* dst = base + (role == 1 ? sm.dot : sm.act) + (x4 * 4) * 68 + q4 * 4;
            *reinterpret_cast<float4*>(dst) = make_float4(r[0].x, r[1].x, r[2].x, r[3].x);
            *reinterpret_cast<float4*>(dst + 68) = make_float4(r[0].y, r[1].y, r[2].y, r[3].y);
            *reinterpret_cast<float4*>(dst + 2 * 68) = make_float4(r[0].z, r[1].z, r[2].z, r[3].z);
            *reinterpret_cast<float4*>(dst + 3 * 68) = make_float4(r[0].w, r[1].w, r[2].w, r[3].w);
        }
    };
    // rows w >= Wc of A_row^T are never written: zero them once in both buffers
    for (int idx = tid; idx < 2 * (64 - Wc) * (ARS / 2); idx += 512) {
        const int b = idx / ((64 - Wc) * (ARS / 2)), r = idx - b * (64 - Wc) * (ARS / 2);
        smem[b * sm.tile + sm.art + Wc * (ARS / 2) + r] = 0.f;
    }

    f32x16 acc[2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float* base = smem + buf * sm.tile;
        const __bf16* art = reinterpret_cast<const __bf16*>(base + sm.art) + i32 * ARS + 8 * g;
        const float* dot = base + sm.dot + i32 * 68 + 8 * g;
        const float* act = base + sm.act + wid * 68 + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float4 d0 = *reinterpret_cast<const float4*>(dot + 16 * ks), d1 = *reinterpret_cast<const float4*>(dot + 16 * ks + 4);
            const float4 a0 = *reinterpret_cast<const float4*>(act + 16 * ks), a1 = *reinterpret_cast<const float4*>(act + 16 * ks + 4);
            const float x[8] = {d0.x * a0.x, d0.y * a0.y, d0.z * a0.z, d0.w * a0.w, d1.x * a1.x, d1.y * a1.y, d1.z * a1.z, d1.w * a1.w};
            bf16x8 bh, bl;
            split_bf16x8(x, bh, bl);
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(art + 32 * f * ARS + 16 * ks);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(art + 32 * f * ARS + 64 + 16 * ks);
                acc[f] = mfma_bf16_terms<TERMS>(ah, al, bh, bl, acc[f]);
            }
        }
    };

    fetch(rs[0], rmask[0]);
    fetch(rs[1], rmask[1]);
    __syncthreads();               // zero fill above
    stash(rs[0], rmask[0], 0);
    __syncthreads();
    int t = 0;
    for (; t + 1 < ntile; t += 2) {
        fetch(rs[0], rmask[0]);            // tile t+2
        compute(0);
        stash(rs[1], rmask[1], 1);
        __syncthreads();
        fetch(rs[1], rmask[1]);            // tile t+3
        compute(1);
        stash(rs[0], rmask[0], 0);
        __syncthreads();
    }
    if (t < ntile) compute(0);
    mfma_drain(acc);
    if (h >= H) return;
    // accumulator layout: row = key column w = 32f + (r&3) + 8(r>>2) + 4g, column = channel i32
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int wl = 32 * f + (r & 3) + 8 * (r >> 2) + 4 * g, w = wbase + wl;
            if (wl < Wc && w < W) atomicAdd(d.d_v + (((long)n * H + h) * W + w) * E + head * D + i32, acc[f][r]);
        }
}

template <int TERMS = 3>
__global__ __launch_bounds__(512) void rcda_dv2_kernel(const cdetr_rcda_bwd_desc d, const int q_per_slice) {
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    xcd_slab(bx_, by_);                  // (every z-plane of the grid is a multiple of 8 workgroups or the map is the identity)
    rcda_dv2_body<TERMS>(d, q_per_slice, bx_, by_, blockIdx.z);
}

// dS / dq / dk (rcda_bwd_body) and dV (rcda_dv2_body) of one attention in ONE launch: the two read the same saved attention maps and d_out
// and write disjoint outputs, so their workgroups share the chip instead of following each other (encoder: 61 + 33 us as two launches).
// blockIdx.x < nqb: a query block of the dS kernel (its 64 * NW threads; the surplus waves of the 512-thread block leave at once);
// the rest: (key-row group, query slice) of the dV kernel.
template <int NF, int NW, int TERMS>
__global__ __launch_bounds__(512) void rcda_bwd_all_kernel(const cdetr_rcda_bwd_desc d, const int nqb, const int hgroups, const int q_per_slice) {
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    xcd_slab(bx_, by_);
    if (bx_ < nqb) {
        if ((int)threadIdx.x >= 64 * NW) return;
        rcda_bwd_body<NF, NW, 1, TERMS>(d, bx_, by_);
    } else {
        const int t = bx_ - nqb;
        rcda_dv2_body<TERMS>(d, q_per_slice, t % hgroups, by_, t / hgroups);
    }
}

template <typename F>
int set_smem(F func, int bytes, const char* what) {
    if (bytes > 160 * 1024) {
        cdetr_set_error("%s: needs %d bytes of LDS (> 160 KiB): feature map too large for this tiling", what, bytes);
        return CDETR_ERR_UNSUPPORTED;
    }
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(func), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) {
            cdetr_set_error("%s: hipFuncSetAttribute(%d): %s", what, bytes, hipGetErrorString(e));
            return CDETR_ERR_LAUNCH;
        }
    }
    return CDETR_OK;
}

template <int NF, int NW>
int launch_rcda_fwd(const cdetr_rcda_fwd_desc& d, hipStream_t st) {
    const FwdSmem sm = fwd_smem(d.H, d.W, NW);
    const int bytes = sm.total * 4;
    int rc;
    dim3 grid((d.L + QW * NW - 1) / (QW * NW), d.N * d.nh), block(64 * NW);
    if (d.precision == 1) {
        if ((rc = set_smem(rcda_fwd_kernel<NF, NW, 1>, bytes, "cdetr_rcda_fwd"))) return rc;
        hipLaunchKernelGGL((rcda_fwd_kernel<NF, NW, 1>), grid, block, bytes, st, d);
    } else {
        if ((rc = set_smem(rcda_fwd_kernel<NF, NW, 0>, bytes, "cdetr_rcda_fwd"))) return rc;
        hipLaunchKernelGGL((rcda_fwd_kernel<NF, NW, 0>), grid, block, bytes, st, d);
    }
    return cdetr_launch_status("cdetr_rcda_fwd");
}
// hgroups > 0: the two-step dV of the same attention (hgroups x slices workgroups, `per` queries per slice) rides in the same launch
template <int NF, int NW>
int launch_rcda_bwd(const cdetr_rcda_bwd_desc& d, hipStream_t st, int hgroups = 0, int slices = 0, int per = 0) {
    const BwdSmem sm = bwd_smem(d.H, d.W, NF, NW, d.dq_row != nullptr);
    int bytes = sm.total * 4;
    int rc;
    dim3 grid((d.L + QW * NW - 1) / (QW * NW), d.N * d.nh), block(64 * NW);
    if (hgroups > 0) {
        bytes = std::max(bytes, dv2_smem().total * 4);
        const int nqb = grid.x;
        dim3 g2(nqb + hgroups * slices, d.N * d.nh);
        if (d.precision == 3) {
            if ((rc = set_smem(rcda_bwd_all_kernel<NF, NW, 1>, bytes, "cdetr_rcda_bwd"))) return rc;
            hipLaunchKernelGGL((rcda_bwd_all_kernel<NF, NW, 1>), g2, dim3(512), bytes, st, d, nqb, hgroups, per);
        } else {
            if ((rc = set_smem(rcda_bwd_all_kernel<NF, NW, 3>, bytes, "cdetr_rcda_bwd"))) return rc;
            hipLaunchKernelGGL((rcda_bwd_all_kernel<NF, NW, 3>), g2, dim3(512), bytes, st, d, nqb, hgroups, per);
        }
        return cdetr_launch_status("cdetr_rcda_bwd(dS+dV)");
    }
    if (d.precision == 3) {          // plain-bf16 products (the backward's arithmetic)
        if ((rc = set_smem(rcda_bwd_kernel<NF, NW, 1, 1>, bytes, "cdetr_rcda_bwd"))) return rc;
        hipLaunchKernelGGL((rcda_bwd_kernel<NF, NW, 1, 1>), grid, block, bytes, st, d);
    } else if (d.precision >= 1) {
        if ((rc = set_smem(rcda_bwd_kernel<NF, NW, 1>, bytes, "cdetr_rcda_bwd"))) return rc;
        hipLaunchKernelGGL((rcda_bwd_kernel<NF, NW, 1>), grid, block, bytes, st, d);
    } else {
        if ((rc = set_smem(rcda_bwd_kernel<NF, NW, 0>, bytes, "cdetr_rcda_bwd"))) return rc;
        hipLaunchKernelGGL((rcda_bwd_kernel<NF, NW, 0>), grid, block, bytes, st, d);
    }
    return cdetr_launch_status("cdetr_rcda_bwd(dS)");
}
// waves per workgroup: 4 (128 queries share every staged V tile).  2-wave workgroups used to pay for the short decoder query
// sets; with the grouped / prefetched tile loops they no longer do (tools/rcda_bench.py: dS 66 vs 73 us, fwd 50 vs 52 us).
// Key-row slices of the two-step forward (gridDim.z of rcda_fwd2_kernel): enough workgroups that every CU holds two (their
// barrier-separated iterations then overlap), each slice keeping >= 8 key rows; 1 when the caller gave no scratch.
inline int fwd2_slices(const cdetr_rcda_fwd_desc& d, int base, int nt, int lds_bytes) {
    if (!d.ws || base > RCDA_WS_COUNTERS) return 1;
    int hs;
    const char* f = cdetr_tune_env("CDETR_RCDA_HS");
    if (f) hs = atoi(f);
    else {
        const int per_cu = lds_bytes > 0 ? std::max(1, std::min(4, (160 * 1024) / lds_bytes)) : 1;
        (void)per_cu;
        hs = 256 / base;                         // as many slices as keep the grid on one workgroup per CU (round 6, after the score phase got cheaper:
                                                 // 48 workgroups -> 5 slices 18.9 us, 4: 20.2, 6: 22.2; 80 -> 3 and 112 -> 2 as before, tools/rcda_slices.py)
    }
    hs = std::min(std::min(hs, 8), d.H / 8);
    const long avail = (long)d.ws_bytes - (long)RCDA_WS_COUNTERS * 4;
    while (hs > 1 && (long)base * hs * 16 * nt * 4 > avail) --hs;
    return hs < 1 ? 1 : hs;
}

inline int pick_nw(int L, int NH) {
    (void)L; (void)NH;
    const char* f = cdetr_tune_env("CDETR_RCDA_NW");
    if (f) return atoi(f) == 2 ? 2 : 4;
    return 4;
}

}  // namespace

extern "C" int cdetr_rcda_fwd(const cdetr_rcda_fwd_desc* dp, void* stream) {
    CDETR_CHECK_ARG(dp != nullptr, "cdetr_rcda_fwd: null descriptor");
    const cdetr_rcda_fwd_desc d = *dp;
    CDETR_CHECK_ARG(d.N > 0 && d.L > 0 && d.H > 0 && d.W > 0 && d.nh > 0, "cdetr_rcda_fwd: bad sizes");
    CDETR_CHECK_ARG(d.H <= 128 && d.W <= 1024, "cdetr_rcda_fwd: H must be <= 128 (got %d)", d.H);
    CDETR_CHECK_ARG(d.q_row && d.q_col && d.k_row && d.k_col && d.v && d.out, "cdetr_rcda_fwd: null pointer");
    CDETR_CHECK_ARG((d.a_row != nullptr) == (d.a_col != nullptr), "cdetr_rcda_fwd: a_row and a_col are saved together or not at all (both NULL: inference)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nw = pick_nw(d.L, d.N * d.nh);
    static const int use_v2 = getenv("CDETR_RCDA_FWD2") ? atoi(getenv("CDETR_RCDA_FWD2")) : 1;
    static const int wide = getenv("CDETR_RCDA_WIDE") ? atoi(getenv("CDETR_RCDA_WIDE")) : 1;       // 0: rounds 1-5 (W <= 64 only), for A/B
    if (use_v2 && d.precision == 1 && d.W <= (wide ? 96 : 64)) {      // two-step form (see rcda_fwd2_kernel); W <= 96 = an 800 x 1333 image (FSCD-LVIS) at stride 16
        const int ks = (d.W + 15) / 16;
        // key rows per barrier (rcda_fwd2_kernel<..., RG>): CDETR_RCDA_RG = 1 | 2
        static const int rg_env = getenv("CDETR_RCDA_RG") ? atoi(getenv("CDETR_RCDA_RG")) : RCDA_RG_DEFAULT;
        const char* rg_t = cdetr_tune_env("CDETR_RCDA_RG");
        const int rg = ((rg_t ? atoi(rg_t) : rg_env) == 2 && ks == 4 && d.H <= 64) ? 2 : 1;
        auto go = [&](auto kern, int NWv, int KSv = 0, int RGv = 1) -> int {     // KSv: the kernel's KS when it is not ceil(W / 16)
            const FwdSmem sm = fwd_smem(d.H, d.W, NWv);
            const int bytes = fwd2_smem(sm, d.H, d.W, KSv ? KSv : ks, RGv).total * 4;
            int rc;
            if ((rc = set_smem(kern, bytes, "cdetr_rcda_fwd"))) return rc;
            dim3 grid((d.L + QW * NWv - 1) / (QW * NWv), d.N * d.nh), block(64 * NWv);
            grid.z = fwd2_slices(d, (int)(grid.x * grid.y), 64 * NWv, bytes);
            hipLaunchKernelGGL(kern, grid, block, bytes, st, d);
            return cdetr_launch_status("cdetr_rcda_fwd");
        };
        // 5-wave workgroups (160 queries) when that lands the grid on <= one workgroup per CU and 4 waves do not: the encoder's
        // 2 x 8 x 2500 queries are 320 workgroups of 128 (64 CUs get two) but exactly 256 of 160
        static const int probe = getenv("CDETR_RCDA_PROBE") ? atoi(getenv("CDETR_RCDA_PROBE")) : 0;
        if (probe && d.ws && ks == 4 && d.H <= 64) {     // tools/rcda_probe.py: unsliced launch, stamps into d.ws
            auto gop = [&](auto kern, int NWv) -> int {
                const FwdSmem sm = fwd_smem(d.H, d.W, NWv);
                const int bytes = fwd2_smem(sm, d.H, d.W, ks, rg).total * 4;
                int rc;
                if ((rc = set_smem(kern, bytes, "cdetr_rcda_fwd"))) return rc;
                dim3 grid((d.L + QW * NWv - 1) / (QW * NWv), d.N * d.nh), block(64 * NWv);
                hipLaunchKernelGGL(kern, grid, block, bytes, st, d);
                return cdetr_launch_status("cdetr_rcda_fwd(probe)");
            };
            if (rg == 2) return probe == 5 ? gop(rcda_fwd2_kernel<5, 4, 2, true, 2>, 5) : gop(rcda_fwd2_kernel<4, 4, 2, true, 2>, 4);
            return probe == 5 ? gop(rcda_fwd2_kernel<5, 4, 2, true>, 5) : gop(rcda_fwd2_kernel<4, 4, 2, true>, 4);
        }
        static const int nw5_env = getenv("CDETR_RCDA_NW5") ? atoi(getenv("CDETR_RCDA_NW5")) : 1;
        const char* nw5_t = cdetr_tune_env("CDETR_RCDA_NW5");
        const int nw5 = nw5_t ? atoi(nw5_t) : nw5_env;
        const long wg4 = (long)((d.L + QW * 4 - 1) / (QW * 4)) * d.N * d.nh, wg5 = (long)((d.L + QW * 5 - 1) / (QW * 5)) * d.N * d.nh;
        // (round 6, measured and dropped: 6 / 7 / 8 waves per workgroup at the encoder shape -- two waves per SIMD so that one wave's MFMAs overlap the
        // other's VALU / LDS work, fewer busy CUs: 51.6 / 53.0 / 54.5 us against 49.5 for the 5-wave form; profiles/r6_rcda_probe.txt)
        if (nw5 && nw == 4 && ks == 4 && wg4 > 256 && wg4 <= 512 && wg5 <= 256)
            return rg == 2 ? go(rcda_fwd2_kernel<5, 4, 2, false, 2>, 5, 0, 2) : go(rcda_fwd2_kernel<5, 4>, 5);
        if (rg == 2 && nw == 4) return go(rcda_fwd2_kernel<4, 4, 2, false, 2>, 4, 0, 2);
        if (ks > 4 || (wide && d.H > 64)) {         // wide / tall maps (round 6): a third 32-key tile in the score phase
            if (d.H > 64 && d.H <= 96) return ks <= 4 ? go(rcda_fwd2_kernel<4, 4, 3>, 4, 4) : go(rcda_fwd2_kernel<4, 6, 3>, 4, 6);
            if (ks <= 4) return go(rcda_fwd2_kernel<4, 4>, 4, 4);        // H > 96: VALU score phase
            return ks == 5 ? go(rcda_fwd2_kernel<4, 5>, 4) : go(rcda_fwd2_kernel<4, 6>, 4);
        }
        if (nw == 4) {
            if (ks == 1) return go(rcda_fwd2_kernel<4, 1>, 4);
            if (ks == 2) return go(rcda_fwd2_kernel<4, 2>, 4);
            if (ks == 3) return go(rcda_fwd2_kernel<4, 3>, 4);
            return go(rcda_fwd2_kernel<4, 4>, 4);
        }
        if (ks == 1) return go(rcda_fwd2_kernel<2, 1>, 2);
        if (ks == 2) return go(rcda_fwd2_kernel<2, 2>, 2);
        if (ks == 3) return go(rcda_fwd2_kernel<2, 3>, 2);
        return go(rcda_fwd2_kernel<2, 4>, 2);
    }
    if (d.H <= 32) return nw == 4 ? launch_rcda_fwd<1, 4>(d, st) : launch_rcda_fwd<1, 2>(d, st);
    if (d.H <= 64) return nw == 4 ? launch_rcda_fwd<2, 4>(d, st) : launch_rcda_fwd<2, 2>(d, st);
    return nw == 4 ? launch_rcda_fwd<4, 4>(d, st) : launch_rcda_fwd<4, 2>(d, st);
}

extern "C" int cdetr_rcda_bwd(const cdetr_rcda_bwd_desc* dp, void* stream) {
    CDETR_CHECK_ARG(dp != nullptr, "cdetr_rcda_bwd: null descriptor");
    const cdetr_rcda_bwd_desc d = *dp;
    CDETR_CHECK_ARG(d.N > 0 && d.L > 0 && d.H > 0 && d.W > 0 && d.nh > 0, "cdetr_rcda_bwd: bad sizes");
    CDETR_CHECK_ARG(d.H <= 128 && d.W <= 1024, "cdetr_rcda_bwd: H must be <= 128 (got %d)", d.H);
    CDETR_CHECK_ARG(d.d_out && d.a_row && d.a_col && d.v && d.d_v, "cdetr_rcda_bwd: null pointer");
    CDETR_CHECK_ARG((d.ds_row && d.ds_col) || (!d.ds_row && !d.ds_col && d.dk_row),
                    "cdetr_rcda_bwd: ds_row / ds_col may only be omitted (both) when the fused q / k gradients are requested");
    CDETR_CHECK_ARG(!d.dq_row || (d.dq_col && d.k_row && d.k_col), "cdetr_rcda_bwd: dq_row needs dq_col, k_row and k_col");
    CDETR_CHECK_ARG(!d.dk_row || (d.dq_row && d.dk_col && d.q_row && d.q_col && d.precision >= 1),
                    "cdetr_rcda_bwd: dk_row needs dk_col, q_row, q_col, the dq_* outputs and a split-bf16 precision (1 or 3)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int NF = d.H <= 32 ? 1 : (d.H <= 64 ? 2 : 4);
    const int Wp = (d.W + 3) & ~3;
    const int nw = pick_nw(d.L, d.N * d.nh);
    int rc;
    // the two-step dV (rcda_dv2_kernel) rides in the dS launch (rcda_bwd_all_kernel) unless CDETR_RCDA_MERGE=0
    static const int use_dv2 = getenv("CDETR_RCDA_DV2") ? atoi(getenv("CDETR_RCDA_DV2")) : 1;
    static const int merge = getenv("CDETR_RCDA_MERGE") ? atoi(getenv("CDETR_RCDA_MERGE")) : 1;
    static const int wide = getenv("CDETR_RCDA_WIDE") ? atoi(getenv("CDETR_RCDA_WIDE")) : 1;
    const bool dv2 = use_dv2 && d.precision >= 1 && (wide || d.W <= 64);
    int hgroups = 0, slices = 0, per = 0;
    if (dv2) {
        hgroups = ((d.H + 7) / 8) * ((Wp + 63) / 64);                // (key-row groups) x (chunks of <= 64 key columns), see rcda_dv2_body
        const long base = (long)hgroups * d.N * d.nh;
        // query slices of the dV workgroups.  Round 6 (after the dS kernel got shorter): 2 slices at the encoder shape (224 workgroups: dS + dV 82-83 us;
        // 4 slices, the rounds 2-5 rule: 85-87; 1 slice: 94-96) and 1 slice when the launch is shared with a short dS grid (decoder: 37.8-38.5 us
        // against 40.0); CDETR_RCDA_DV2_TARGET overrides
        static const int dv2_target_env = getenv("CDETR_RCDA_DV2_TARGET") ? atoi(getenv("CDETR_RCDA_DV2_TARGET")) : 0;
        const long ds_wgs0 = (long)((d.L + QW * nw - 1) / (QW * nw)) * d.N * d.nh;
        const int dv2_target = dv2_target_env ? dv2_target_env : (ds_wgs0 <= 128 ? 112 : 224);
        slices = (int)((dv2_target + base - 1) / base);
        const int max_slices = (d.L + 255) / 256;                    // >= 4 q-tiles per slice
        if (slices > max_slices) slices = max_slices;
        if (slices < 1) slices = 1;
        per = (d.L + slices - 1) / slices;
        per = ((per + 63) / 64) * 64;
        slices = (d.L + per - 1) / per;
    }
    // (Rounds 2-5 merged only when the dS grid left most CUs free -- decoder: 48 workgroups --, because at the encoder shape 61 + 33 us became
    // 117 us with the kernels of that time.  The H > 64 kernels need more than the 256 registers a 512-thread block leaves a wave: never merged.)
    const long ds_wgs = (long)((d.L + QW * nw - 1) / (QW * nw)) * d.N * d.nh;
    // Round 6: merged at EVERY shape the two-step dV covers.  With the dS body 13 % shorter and the dV workgroups on two query slices the shared
    // launch now wins at the encoder shape too (dS + dV 82 -> 76.4-77.1 us; 50 x 84 keys, L = 4200: 275 -> 176 us; step -0.02...-0.09 ms in three
    // same-lease pairs, profiles/r6_ab_rcda_slices.txt) -- the 117 us of round 2 was measured with the round-2 kernels.  CDETR_RCDA_MERGE=3: the old
    // rule (merge only when the dS grid is <= 128 workgroups), 0: never.
    const int mh = (dv2 && NF < 4 && (merge == 1 || merge == 2 || (merge == 3 && ds_wgs <= 128))) ? hgroups : 0;
    static const int dk_per_wave = getenv("CDETR_RCDA_DK_PER_WAVE") ? atoi(getenv("CDETR_RCDA_DK_PER_WAVE")) : 0;      // A/B only
    static bool dk_pushed = false;
    if (dk_per_wave && !dk_pushed) {
        const int one = 1;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rcda_dk_per_wave), &one, sizeof(int));
        dk_pushed = true;
    }
    static const int probe = getenv("CDETR_RCDA_PROBE") ? atoi(getenv("CDETR_RCDA_PROBE")) : 0;
    if (probe && NF == 2 && d.precision == 3 && d.ds_row && d.dk_row) {      // tools/rcda_probe.py: the dS kernel alone, stamps into ds_row
        const int NWp = probe == 5 ? 5 : 4;
        const BwdSmem sm = bwd_smem(d.H, d.W, 2, NWp, true);
        dim3 grid((d.L + QW * NWp - 1) / (QW * NWp), d.N * d.nh), block(64 * NWp);
        if (NWp == 5) {
            if ((rc = set_smem(rcda_bwd_kernel<2, 5, 1, 1, true>, sm.total * 4, "cdetr_rcda_bwd"))) return rc;
            hipLaunchKernelGGL((rcda_bwd_kernel<2, 5, 1, 1, true>), grid, block, sm.total * 4, st, d);
        } else {
            if ((rc = set_smem(rcda_bwd_kernel<2, 4, 1, 1, true>, sm.total * 4, "cdetr_rcda_bwd"))) return rc;
            hipLaunchKernelGGL((rcda_bwd_kernel<2, 4, 1, 1, true>), grid, block, sm.total * 4, st, d);
        }
        return cdetr_launch_status("cdetr_rcda_bwd(probe)");
    }
    if (NF == 1) rc = nw == 4 ? launch_rcda_bwd<1, 4>(d, st, mh, slices, per) : launch_rcda_bwd<1, 2>(d, st, mh, slices, per);
    else if (NF == 2) {
        // 5-wave workgroups when they put the grid on <= one workgroup per CU and 4-wave ones do not (see cdetr_rcda_fwd)
        static const int nw5 = getenv("CDETR_RCDA_NW5") ? atoi(getenv("CDETR_RCDA_NW5")) : 1;
        const long wg4 = (long)((d.L + QW * 4 - 1) / (QW * 4)) * d.N * d.nh, wg5 = (long)((d.L + QW * 5 - 1) / (QW * 5)) * d.N * d.nh;
        if (nw5 && nw == 4 && wg4 > 256 && wg4 <= 512 && wg5 <= 256) rc = launch_rcda_bwd<2, 5>(d, st, mh, slices, per);
        else rc = nw == 4 ? launch_rcda_bwd<2, 4>(d, st, mh, slices, per) : launch_rcda_bwd<2, 2>(d, st, mh, slices, per);
    }
    else rc = nw == 4 ? launch_rcda_bwd<4, 4>(d, st, mh, slices, per) : launch_rcda_bwd<4, 2>(d, st, mh, slices, per);
    if (rc || mh) return rc;
    if (dv2) {   // two-step dV as its own launch
        const int bytes = dv2_smem().total * 4;
        if (d.precision == 3) {
            if ((rc = set_smem(rcda_dv2_kernel<1>, bytes, "cdetr_rcda_bwd(dV)"))) return rc;
            hipLaunchKernelGGL(rcda_dv2_kernel<1>, dim3(hgroups, d.N * d.nh, slices), dim3(512), bytes, st, d, per);
        } else {
            if ((rc = set_smem(rcda_dv2_kernel<3>, bytes, "cdetr_rcda_bwd(dV)"))) return rc;
            hipLaunchKernelGGL(rcda_dv2_kernel<3>, dim3(hgroups, d.N * d.nh, slices), dim3(512), bytes, st, d, per);
        }
        return cdetr_launch_status("cdetr_rcda_bwd(dV)");
    }
    {   // dV kernel
        const int bytes = (64 * 32 * NF + 64 * Wp + 64 * 32) * 4;
        const int wgroups = (d.W + 3) / 4;
        long base = (long)wgroups * d.N * d.nh;
        int slices = (int)((1024 + base - 1) / base);
        const int max_slices = (d.L + 127) / 128;
        if (slices > max_slices) slices = max_slices;
        if (slices < 1) slices = 1;
        int per = (d.L + slices - 1) / slices;
        per = ((per + 63) / 64) * 64;
        slices = (d.L + per - 1) / per;
        dim3 grid(wgroups, d.N * d.nh, slices), block(256);
        auto dv = [&](auto kern) {
            if ((rc = set_smem(kern, bytes, "cdetr_rcda_bwd(dV)"))) return;
            hipLaunchKernelGGL(kern, grid, block, bytes, st, d, per);
        };
        rc = CDETR_OK;
        if (d.precision >= 1) {
            if (NF == 1) dv(rcda_dv_kernel<1, 1>); else if (NF == 2) dv(rcda_dv_kernel<2, 1>); else dv(rcda_dv_kernel<4, 1>);
        } else {
            if (NF == 1) dv(rcda_dv_kernel<1, 0>); else if (NF == 2) dv(rcda_dv_kernel<2, 0>); else dv(rcda_dv_kernel<4, 0>);
        }
        if (rc) return rc;
        if ((rc = cdetr_launch_status("cdetr_rcda_bwd(dV)"))) return rc;
    }
    return CDETR_OK;
}
