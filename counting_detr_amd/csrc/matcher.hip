// matcher.hip -- Hungarian matching on the device.
//
// cdetr_match_cost : the reference's matching cost (A2/models/matcher.py:222-242) per image only (the reference
//                    also computes -- and throws away -- the cross-image blocks), fp32, same expression and
//                    summation order: C = 5*L1 + 2*(pos - neg) + 2*(-GIoU); contraction of mul+add is disabled
//                    in this file so every product is rounded like the reference's separate torch ops.
// cdetr_lsap       : exact rectangular linear sum assignment, the algorithm of scipy.optimize.linear_sum_assignment
//                    (Crouse 2016: shortest augmenting paths, float64 duals, tall matrices transposed, ties broken in
//                    favour of a not-yet-assigned column, scan order = the `remaining` list with swap-removal) so the
//                    returned indices are those scipy returns -- including on ties.  One workgroup per image; all solver
//                    state lives in LDS; the column scan is lane-parallel with a wave-shuffle arg-min whose comparator
//                    reproduces the sequential tie rule.  Removes the reference's `.cpu()` pipeline drain
//                    (A2/models/matcher.py:243) so the whole train step stays on the device and is graph-capturable.
#include "../../include/cdetr_hip.h"
#include "common.h"
#include <stdlib.h>

#pragma clang fp contract(off)

namespace {

__global__ __launch_bounds__(256) void match_cost_kernel(const float* __restrict__ logits, int ncls,
                                                         const float* __restrict__ boxes, const float* __restrict__ tgt,
                                                         const int* __restrict__ tgt_off, const int64_t* __restrict__ cost_off,
                                                         int Q, float w_class, float w_bbox, float w_giou,
                                                         float* __restrict__ cost) {
    const int b = blockIdx.y;
    const int t0 = tgt_off[b], T = tgt_off[b + 1] - t0;
    const long total = (long)Q * T;
    const bool transpose = T < Q;   // solver layout: rows = the shorter side
    float* out = cost + cost_off[b];
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        // enumerate in OUTPUT order so stores coalesce
        int q, t;
        if (transpose) { t = (int)(idx / Q); q = (int)(idx - (long)t * Q); }
        else { q = (int)(idx / T); t = (int)(idx - (long)q * T); }
        const float x = logits[((long)b * Q + q) * ncls + 0];   // tgt_ids are all 0 (A2/data/fsc147.py:84)
        const float p = 1.f / (1.f + expf(-x));
        const float neg = (0.75f * (p * p)) * (-logf((1.f - p) + 1e-8f));
        const float pos = (0.25f * ((1.f - p) * (1.f - p))) * (-logf(p + 1e-8f));
        const float cost_class = pos - neg;
        const float* ob = boxes + ((long)b * Q + q) * 4;
        const float* tb = tgt + ((long)t0 + t) * 4;
        const float ocx = ob[0], ocy = ob[1], ow = ob[2], oh = ob[3];
        const float tcx = tb[0], tcy = tb[1], tw = tb[2], th = tb[3];
        const float cost_bbox = ((fabsf(ocx - tcx) + fabsf(ocy - tcy)) + fabsf(ow - tw)) + fabsf(oh - th);
        const float ox0 = ocx - 0.5f * ow, oy0 = ocy - 0.5f * oh, ox1 = ocx + 0.5f * ow, oy1 = ocy + 0.5f * oh;
        const float tx0 = tcx - 0.5f * tw, ty0 = tcy - 0.5f * th, tx1 = tcx + 0.5f * tw, ty1 = tcy + 0.5f * th;
        const float a1 = (ox1 - ox0) * (oy1 - oy0), a2 = (tx1 - tx0) * (ty1 - ty0);
        const float iw = fmaxf(fminf(ox1, tx1) - fmaxf(ox0, tx0), 0.f), ih = fmaxf(fminf(oy1, ty1) - fmaxf(oy0, ty0), 0.f);
        const float inter = iw * ih;
        const float uni = (a1 + a2) - inter;
        const float iou = inter / uni;
        const float ew = fmaxf(fmaxf(ox1, tx1) - fminf(ox0, tx0), 0.f), eh = fmaxf(fmaxf(oy1, ty1) - fminf(oy0, ty0), 0.f);
        const float area = ew * eh;
        const float giou = iou - (area - uni) / area;
        const float c = (w_bbox * cost_bbox + w_class * cost_class) + w_giou * (-giou);
        out[idx] = c;
    }
}

struct Cand {          // arg-min candidate with the sequential scan's tie rule
    double val;
    int it;            // position in `remaining`
    int unassigned;    // row4col[j] == -1
};
__device__ __forceinline__ Cand better(const Cand& a, const Cand& b) {
    // sequential rule: strictly smaller value wins; among equal values the LAST unassigned column wins, and if none
    // is unassigned the FIRST column wins.
    if (a.it < 0) return b;
    if (b.it < 0) return a;
    if (a.val < b.val) return a;
    if (b.val < a.val) return b;
    if (a.unassigned != b.unassigned) return a.unassigned ? a : b;
    if (a.unassigned) return (a.it > b.it) ? a : b;
    return (a.it < b.it) ? a : b;
}
__device__ __forceinline__ Cand shfl_xor_cand(const Cand& c, int m) {
    Cand r;
    r.val = __shfl_xor(c.val, m, 64);
    r.it = __shfl_xor(c.it, m, 64);
    r.unassigned = __shfl_xor(c.unassigned, m, 64);
    return r;
}

// One workgroup (NT threads) per image.
template <int NT>
__global__ __launch_bounds__(NT) void lsap_kernel(const float* __restrict__ cost_all, const int64_t* __restrict__ cost_off,
                                                  const int* __restrict__ tgt_off, int Q, int Mmax, int64_t* __restrict__ idx_i,
                                                  int64_t* __restrict__ idx_j, int* __restrict__ status, int nc_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __builtin_amdgcn_s_setprio(3);         // (critical-path kernel that may share its SIMDs with another stream's throughput work)
    const int b = blockIdx.x;
    const int T = tgt_off[b + 1] - tgt_off[b];
    const bool transpose = T < Q;
    const int nr = transpose ? T : Q;      // rows of the solver problem (the shorter side)
    const int nc = transpose ? Q : T;
    const float* cost = cost_all + cost_off[b];
    const int tid = threadIdx.x;
    int64_t* oi = idx_i + (long)b * Mmax;
    int64_t* oj = idx_j + (long)b * Mmax;
    for (int i = tid; i < Mmax; i += NT) { oi[i] = 0; oj[i] = 0; }     // rows are valid (in range) on every exit path; the caller does not pre-fill
    __syncthreads();
    if (nr == 0) { if (tid == 0) status[b] = 0; return; }

    // LDS carve (nc_cap >= nc >= nr): v, spc, u f64 | path, row4col, remaining, col4row i32 | SC, SR u8
    double* v = reinterpret_cast<double*>(lds);
    double* spc = v + nc_cap;
    double* u = spc + nc_cap;
    int* path = reinterpret_cast<int*>(u + nc_cap);
    int* row4col = path + nc_cap;
    int* remaining = row4col + nc_cap;
    int* col4row = remaining + nc_cap;
    unsigned char* SC = reinterpret_cast<unsigned char*>(col4row + nc_cap);
    unsigned char* SR = SC + nc_cap;
    __shared__ Cand wave_best[NT / 64];
    __shared__ int s_i, s_sink, s_num_remaining, s_bad;
    __shared__ double s_min;

    if (tid == 0) s_bad = 0;
    for (int j = tid; j < nc; j += NT) { v[j] = 0.0; path[j] = -1; row4col[j] = -1; }
    for (int i = tid; i < nr; i += NT) { u[i] = 0.0; col4row[i] = -1; }
    __syncthreads();
    // validity scan (scipy raises ValueError on NaN / -inf entries)
    {
        int bad = 0;
        for (long k = tid; k < (long)nr * nc; k += NT) {
            const float c = cost[k];
            if (c != c || c == -INFINITY) bad = 1;
        }
        if (bad) atomicOr(&s_bad, 1);
    }
    __syncthreads();
    if (s_bad) { if (tid == 0) status[b] = 2; return; }

    for (int cur = 0; cur < nr; ++cur) {
        // ---- augmenting_path(cur)
        for (int j = tid; j < nc; j += NT) { spc[j] = INFINITY; SC[j] = 0; remaining[j] = nc - j - 1; }
        for (int i = tid; i < nr; i += NT) SR[i] = 0;
        if (tid == 0) { s_i = cur; s_sink = -1; s_num_remaining = nc; s_min = 0.0; }
        __syncthreads();
        while (true) {
            const int i = s_i;
            const int num_remaining = s_num_remaining;
            const double min_val = s_min;
            const double ui = u[i];
            const float* crow = cost + (long)i * nc;
            Cand best;
            best.val = INFINITY; best.it = -1; best.unassigned = 0;
            for (int it = tid; it < num_remaining; it += NT) {
                const int j = remaining[it];
                const double r = ((min_val + (double)crow[j]) - ui) - v[j];
                double s = spc[j];
                if (r < s) { path[j] = i; spc[j] = r; s = r; }
                Cand c;
                c.val = s; c.it = it; c.unassigned = (row4col[j] == -1);
                best = better(best, c);
            }
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) best = better(best, shfl_xor_cand(best, m));
            if (NT > 64) {
                if ((tid & 63) == 0) wave_best[tid >> 6] = best;
                __syncthreads();
                if (tid == 0) {
                    Cand bb = wave_best[0];
                    for (int wv = 1; wv < NT / 64; ++wv) bb = better(bb, wave_best[wv]);
                    wave_best[0] = bb;
                }
                __syncthreads();
                best = wave_best[0];
            }
            // every thread now holds the same `best`
            if (tid == 0) {
                SR[i] = 1;
                if (best.it < 0 || best.val == INFINITY) {
                    s_sink = -2;      // infeasible
                } else {
                    const int j = remaining[best.it];
                    s_min = best.val;
                    if (row4col[j] == -1) s_sink = j;
                    else s_i = row4col[j];
                    SC[j] = 1;
                    remaining[best.it] = remaining[num_remaining - 1];
                    s_num_remaining = num_remaining - 1;
                }
            }
            __syncthreads();
            if (s_sink != -1) break;
        }
        const int sink = s_sink;
        if (sink == -2) { if (tid == 0) status[b] = 1; return; }
        const double min_val = s_min;
        // ---- dual update
        for (int i = tid; i < nr; i += NT) {
            if (i == cur) u[i] += min_val;
            else if (SR[i]) u[i] += min_val - spc[col4row[i]];
        }
        for (int j = tid; j < nc; j += NT)
            if (SC[j]) v[j] -= min_val - spc[j];
        __syncthreads();
        // ---- augment along the path (sequential, short)
        if (tid == 0) {
            int j = sink;
            while (true) {
                const int i = path[j];
                row4col[j] = i;
                const int t = col4row[i];
                col4row[i] = j;
                j = t;
                if (i == cur) break;
            }
        }
        __syncthreads();
    }
    // ---- write (query, target) pairs with ascending query index
    if (!transpose) {
        for (int i = tid; i < nr; i += NT) { oi[i] = i; oj[i] = col4row[i]; }
    } else {
        // rows are targets, columns are queries: walk the queries in order, emit the assigned ones (stable, ascending)
        // rank of query j = number of assigned queries with a smaller index
        for (int j = tid; j < nc; j += NT) remaining[j] = (row4col[j] != -1) ? 1 : 0;
        __syncthreads();
        if (tid == 0) {   // nc <= a few thousand: a serial prefix sum is cheap and simple
            int run = 0;
            for (int j = 0; j < nc; ++j) { const int f = remaining[j]; remaining[j] = run; run += f; }
        }
        __syncthreads();
        for (int j = tid; j < nc; j += NT)
            if (row4col[j] != -1) { oi[remaining[j]] = j; oj[remaining[j]] = row4col[j]; }
    }
    if (tid == 0) status[b] = 0;
}


// ------------------------------------------------------------------------------------------------ single-wave solver
// Same algorithm, tie rule and scan order as lsap_kernel, organised for latency: ONE wavefront per image, no barriers.
// Lane l owns columns l, l+64, ... (<= CPL per lane): their dual v, shortest-path cost, predecessor and position in
// scipy's `remaining` list live in REGISTERS; rows' duals / assignments, row4col, `remaining` and -- when it fits -- the
// whole fp32 cost matrix live in LDS.  One inner iteration = CPL fused updates per lane + a 6-step arg-min over the wave.
// The SR/SC bookkeeping of the reference algorithm collapses to a per-lane bit mask: the visited rows (other than the
// current one) are exactly row4col[j] of the visited non-sink columns.
// The iteration is one dependent chain (next row <- arg-min <- scan <- cost row), so its latency is the kernel's time:
//   * the arg-min runs on DPP lane permutes (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror, row_bcast 15 / 31: ~10 VALU
//     per step) instead of 24 ds_bpermute round trips through the LDS pipe; only (value, key) are reduced -- keys are unique, so
//     the owner of the winning pair identifies itself and its column / row4col entry are fetched with v_readlane;
//   * row4col of the owned columns is mirrored in registers (refreshed after each augmentation), the tie keys are cached and
//     only recomputed when a column's position in `remaining` changes.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_min_f64(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const double o = __hiloint2double(ohi, olo);
    return (o < x) ? o : x;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_min_u32(unsigned x) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, ROW_MASK, 0xf, false);
    return o < x ? o : x;
}
// lexicographic minimum of (val, key) over the wave, returned in every lane: first the minimum value (DPP butterfly inside the
// rows of 16, row_bcast across them, lane 63 holds the result), then the minimum key among the lanes that hold that value
template <bool DPP>
__device__ __forceinline__ void wave_lexmin(double& val, unsigned& key) {
    if constexpr (!DPP) {          // A/B reference: butterfly over ds_bpermute
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            const double ov = __shfl_xor(val, m, 64);
            const unsigned ok = (unsigned)__shfl_xor((int)key, m, 64);
            const bool take = (ov < val) | ((ov == val) & (ok < key));
            val = take ? ov : val;
            key = take ? ok : key;
        }
        return;
    }
    double m = val;
    m = dpp_min_f64<0xB1, 0xf>(m);       // quad_perm [1,0,3,2]
    m = dpp_min_f64<0x4E, 0xf>(m);       // quad_perm [2,3,0,1]
    m = dpp_min_f64<0x141, 0xf>(m);      // row_half_mirror: 8 lanes
    m = dpp_min_f64<0x140, 0xf>(m);      // row_mirror: 16 lanes
    m = dpp_min_f64<0x142, 0xa>(m);      // row_bcast:15 into rows 1 and 3
    m = dpp_min_f64<0x143, 0xc>(m);      // row_bcast:31 into rows 2 and 3
    const double gmin = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m), 63), __builtin_amdgcn_readlane(__double2loint(m), 63));
    unsigned k = (val == gmin) ? key : 0xffffffffu;
    k = dpp_min_u32<0xB1, 0xf>(k);
    k = dpp_min_u32<0x4E, 0xf>(k);
    k = dpp_min_u32<0x141, 0xf>(k);
    k = dpp_min_u32<0x140, 0xf>(k);
    k = dpp_min_u32<0x142, 0xa>(k);
    k = dpp_min_u32<0x143, 0xc>(k);
    key = (unsigned)__builtin_amdgcn_readlane((int)k, 63);
    val = gmin;
}

template <int CPL, bool COST_LDS, bool DPP = true>
__global__ __launch_bounds__(64) void lsap_wave_kernel(const float* __restrict__ cost_all, const int64_t* __restrict__ cost_off,
                                                       const int* __restrict__ tgt_off, int Q, int Mmax,
                                                       int64_t* __restrict__ idx_i, int64_t* __restrict__ idx_j,
                                                       int* __restrict__ status, int prio) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // one wave on the step's critical path, possibly sharing its SIMD with throughput work of another stream (the trainer runs the next
    // batch's frozen stage beside the solve): highest wave priority at the instruction arbiter
    if (prio) __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x;
    const int T = tgt_off[b + 1] - tgt_off[b];
    const bool transpose = T < Q;
    const int nr = transpose ? T : Q;
    const int nc = transpose ? Q : T;
    const float* cost_g = cost_all + cost_off[b];
    const int lane = threadIdx.x;
    int64_t* oi = idx_i + (long)b * Mmax;
    int64_t* oj = idx_j + (long)b * Mmax;
    for (int i = lane; i < Mmax; i += 64) { oi[i] = 0; oj[i] = 0; }    // rows are valid (in range) on every exit path; the caller does not pre-fill
    if (nr == 0) { if (lane == 0) status[b] = 0; return; }

    // LDS carve: u[nr] f64 | col4row[nr] | row4col[nc] | remaining[nc] | path[nc] | cost[nr*nc] f32 (optional)
    double* u = reinterpret_cast<double*>(lds);
    int* col4row = reinterpret_cast<int*>(u + nr);
    int* row4col = col4row + nr;
    int* remaining = row4col + nc;
    int* pathl = remaining + nc;
    float* cost_l = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(pathl + nc) + 15) & ~(uintptr_t)15);   // 16-byte aligned (the host adds 64 B of slack)
    const float* cost = COST_LDS ? cost_l : cost_g;

    // cost matrix -> LDS + validity scan.  One wave has to cover the whole L2 round trip per batch of loads, so the batch is made
    // large: 16 unconditional loads in flight per lane (clamped index, masked use) -- 16 B each when the block is 16-byte aligned.
    // (A plain one-element-per-trip loop spent ~200 us here for a 300 x 120 matrix: more than the assignment itself.)
    int bad = 0;
    const long ntot = (long)nr * nc;
    constexpr int LU = 16;
    if ((reinterpret_cast<uintptr_t>(cost_g) & 15) == 0) {
        const long n4 = ntot >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(cost_g);
        for (long k0 = 0; k0 < n4; k0 += 64 * LU) {
            float4 t[LU];
#pragma unroll
            for (int q = 0; q < LU; ++q) t[q] = g4[min(k0 + lane + 64 * q, n4 - 1)];
#pragma unroll
            for (int q = 0; q < LU; ++q) {
                const long k = k0 + lane + 64 * q;
                if (k < n4) {
                    const float4 c = t[q];
                    if (c.x != c.x || c.x == -INFINITY || c.y != c.y || c.y == -INFINITY || c.z != c.z || c.z == -INFINITY ||
                        c.w != c.w || c.w == -INFINITY) bad = 1;
                    if (COST_LDS) *reinterpret_cast<float4*>(cost_l + 4 * k) = c;
                }
            }
        }
        for (long k = (n4 << 2) + lane; k < ntot; k += 64) {
            const float c = cost_g[k];
            if (c != c || c == -INFINITY) bad = 1;
            if (COST_LDS) cost_l[k] = c;
        }
    } else {
        for (long k0 = 0; k0 < ntot; k0 += 64 * LU) {
            float t[LU];
#pragma unroll
            for (int q = 0; q < LU; ++q) t[q] = cost_g[min(k0 + lane + 64 * q, ntot - 1)];
#pragma unroll
            for (int q = 0; q < LU; ++q) {
                const long k = k0 + lane + 64 * q;
                if (k < ntot) {
                    if (t[q] != t[q] || t[q] == -INFINITY) bad = 1;
                    if (COST_LDS) cost_l[k] = t[q];
                }
            }
        }
    }
    if (__any(bad)) { if (lane == 0) status[b] = 2; return; }
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = lane; j < nc; j += 64) row4col[j] = -1;

    double v[CPL], spc[CPL];
    int path[CPL], r4c[CPL];                 // r4c: register mirror of row4col for the owned columns
    unsigned key[CPL];                       // tie key of the owned columns (see below), valid while the column is unvisited
    unsigned valid = 0;                      // bit c: column lane + 64c exists
#pragma unroll
    for (int c = 0; c < CPL; ++c) { v[c] = 0.0; path[c] = -1; r4c[c] = -1; if (lane + 64 * c < nc) valid |= 1u << c; }

    // The scan's tie rule (strictly smaller value wins; among equal values the LAST unassigned column, else the FIRST
    // column -- positions refer to the `remaining` array) is folded into ONE 32-bit key to minimise next to the value:
    //   unassigned: key = 0x7fffffff - it      assigned: key = 0x80000000 + it      (no candidate: 0xffffffff)
    for (int cur = 0; cur < nr; ++cur) {
        // ---- fast path: the FIRST scan step of this row (identical arithmetic and tie key) already ends on an unassigned column.  Then
        // the augmenting path is the single edge (cur, j*): u[cur] += min, no column dual moves (the only visited column is the sink:
        // delta = min - min = 0), and the row's search state (`remaining`, shortest-path registers, predecessor array) is never looked at
        // again -- so none of it is set up.  With more columns than rows most rows end here (T = 120 targets on Q = 300 queries: ~80 %);
        // a row whose nearest column is taken falls through to the full search below, which starts from scratch.
        {
            const float* crow0 = cost + (long)cur * nc;
            const double ui0 = u[cur];
            const double min0 = 0.0;
            double bval = INFINITY;
            unsigned bkey = 0xffffffffu;
            int best_c = 0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const bool a = (valid >> c) & 1u;
                const int jj = a ? lane + 64 * c : 0;
                const double r = ((min0 + (double)crow0[jj]) - ui0) - v[c];
                const double s = (a & (r < INFINITY)) ? r : INFINITY;              // the column's shortest-path value after the first step
                const unsigned p0 = (unsigned)(nc - 1 - (lane + 64 * c));
                const unsigned k = (r4c[c] == -1) ? (0x7fffffffu - p0) : (0x80000000u + p0);
                const bool better = a & ((s < bval) | ((s == bval) & (k < bkey)));
                bval = better ? s : bval;
                bkey = better ? k : bkey;
                best_c = better ? c : best_c;
            }
            const double my_val = bval;
            const unsigned my_key = bkey;
            wave_lexmin<DPP>(bval, bkey);
            if (bkey != 0xffffffffu && bval != INFINITY && (bkey & 0x80000000u) == 0u) {
                const bool mine = (my_key == bkey) && (my_val == bval);
                const int wl = __builtin_ctzll(__ballot(mine));
                const int jstar = __builtin_amdgcn_readlane(lane + 64 * best_c, wl);
                if (lane == 0) {
                    u[cur] = ui0 + bval;
                    row4col[jstar] = cur;
                    col4row[cur] = jstar;
                }
#pragma unroll
                for (int c = 0; c < CPL; ++c) r4c[c] = (mine && best_c == c) ? cur : r4c[c];
                continue;
            }
        }
        unsigned sc = 0;                       // visited-column mask of this lane
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            spc[c] = INFINITY;
            const unsigned p0 = (unsigned)(nc - 1 - (lane + 64 * c));
            key[c] = (r4c[c] == -1) ? (0x7fffffffu - p0) : (0x80000000u + p0);
        }
        for (int it = lane; it < nc; it += 64) remaining[it] = nc - 1 - it;
        int i = cur, num_remaining = nc, sink = -1;
        double min_val = 0.0;
        while (true) {
            const int last = num_remaining - 1;
            const int jlast = *reinterpret_cast<volatile int*>(remaining + last);   // needed only after the arg-min; volatile keeps the
                                                                                   // read up here, off the critical path
            const float* crow = cost + (long)i * nc;
            const double ui = u[i];
            double bval = INFINITY;
            unsigned bkey = 0xffffffffu;
            int best_c = 0;
            const unsigned act = valid & ~sc;
            // branch-free scan (selects, no exec-mask regions): columns past nc read a clamped address and are masked out
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const bool a = (act >> c) & 1u;
                const int jj = ((valid >> c) & 1u) ? lane + 64 * c : 0;
                const double r = ((min_val + (double)crow[jj]) - ui) - v[c];
                const bool upd = a & (r < spc[c]);
                spc[c] = upd ? r : spc[c];
                path[c] = upd ? i : path[c];
                const bool better = a & ((spc[c] < bval) | ((spc[c] == bval) & (key[c] < bkey)));
                bval = better ? spc[c] : bval;
                bkey = better ? key[c] : bkey;
                best_c = better ? c : best_c;
            }
            const double my_val = bval;
            const unsigned my_key = bkey;
            wave_lexmin<DPP>(bval, bkey);
            if (bkey == 0xffffffffu || bval == INFINITY) { sink = -2; break; }
            min_val = bval;
            // keys are unique among candidates: exactly one lane owns the winning pair
            const bool mine = (my_key == bkey) && (my_val == bval);
            const int wl = __builtin_ctzll(__ballot(mine));
            int my_r = r4c[0];
#pragma unroll
            for (int c = 1; c < CPL; ++c) my_r = (best_c == c) ? r4c[c] : my_r;
            const int jstar = __builtin_amdgcn_readlane(lane + 64 * best_c, wl);
            const int inext = __builtin_amdgcn_readlane(my_r, wl);
            const bool j_unassigned = (bkey & 0x80000000u) == 0u;
            const int best_it = j_unassigned ? (int)(0x7fffffffu - bkey) : (int)(bkey - 0x80000000u);
            sc |= mine ? (1u << best_c) : 0u;             // visited-column bit on the owner lane
            // swap-removal from `remaining` (position best_it): the column that sat last moves to best_it
            if (lane == 0) remaining[best_it] = jlast;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const unsigned nk = (r4c[c] == -1) ? (0x7fffffffu - (unsigned)best_it) : (0x80000000u + (unsigned)best_it);
                key[c] = (lane + 64 * c == jlast) ? nk : key[c];
            }
            num_remaining = last;
            if (j_unassigned) { sink = jstar; break; }
            i = inext;
        }
        if (sink == -2) { if (lane == 0) status[b] = 1; return; }
        // ---- dual update: rows visited (other than cur) = row4col[j] of the visited non-sink columns
        if (lane == 0) u[cur] += min_val;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = lane + 64 * c;
            if ((sc >> c) & 1u) {
                const double delta = min_val - spc[c];
                v[c] -= delta;
                if (j != sink) u[r4c[c]] += delta;
            }
            if (j < nc) pathl[j] = path[c];
        }
        // ---- augment (short sequential walk)
        if (lane == 0) {
            int j = sink;
            while (true) {
                const int ii = pathl[j];
                row4col[j] = ii;
                const int t = col4row[ii];
                col4row[ii] = j;
                j = t;
                if (ii == cur) break;
            }
        }
        // refresh the register mirror (same wave: the LDS writes above are ordered by program order)
#pragma unroll
        for (int c = 0; c < CPL; ++c)
            if ((valid >> c) & 1u) r4c[c] = row4col[lane + 64 * c];
    }
    // ---- (query, target) pairs with ascending query index
    if (!transpose) {
        for (int i = lane; i < nr; i += 64) { oi[i] = i; oj[i] = col4row[i]; }
    } else {
        int base = 0;
        for (int j0 = 0; j0 < nc; j0 += 64) {
            const int j = j0 + lane;
            const int r = (j < nc) ? row4col[j] : -1;
            const unsigned long long m = __ballot(r != -1);
            if (r != -1) {
                const int rank = base + __popcll(m & ((1ull << lane) - 1ull));
                oi[rank] = j;
                oj[rank] = r;
            }
            base += __popcll(m);
        }
    }
    if (lane == 0) status[b] = 0;
}

// ------------------------------------------------------------------------------------------------ multi-wave solver (crowded images)
// More than 1024 columns (FSC-147 holds up to 3731 objects per image, A2/data/fsc147.py:80-84; the reference hands these matrices to scipy,
// A2/models/matcher.py:243-247): the single wave's register budget is exhausted and the cost matrix (Q x T fp32: 2.5-4.5 MB) no longer fits
// the LDS.  lsap_kernel<256> (indirect `remaining` reads, f64 state in LDS, four barriers and a one-thread section per scan step) took
// 5.2 / 6.9 ms at T = (37, 2100) / (3000, 3731) -- as long as the rest of the step.  This kernel is lsap_wave_kernel spread over the 16 waves of
// ONE workgroup: thread t owns columns t, t + 1024, ... (CPL per thread) with their dual, shortest-path value, predecessor, row4col mirror
// and tie key in REGISTERS; a scan step is CPL fused updates per thread, the DPP arg-min inside each wave, ONE barrier, and a 16-slot
// LDS arg-min that every thread evaluates for itself (double-buffered slots: no second barrier).  Same arithmetic (f64 expression order),
// same tie key (scipy's sequential rule folded into one integer) -> the same assignment, bit for bit.  The cost row of the step comes
// from L2 / HBM: that round trip is what a step costs now.
struct WgSlot { double val; unsigned key; int jstar; int rnext; int pad_; };

template <int CPL>
__global__ __launch_bounds__(1024) void lsap_wg_kernel(const float* __restrict__ cost_all, const int64_t* __restrict__ cost_off,
                                                       const int* __restrict__ tgt_off, int Q, int Mmax, int64_t* __restrict__ idx_i,
                                                       int64_t* __restrict__ idx_j, int* __restrict__ status) {
    constexpr int NT = 1024, NW = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x;
    const int T = tgt_off[b + 1] - tgt_off[b];
    const bool transpose = T < Q;
    const int nr = transpose ? T : Q;
    const int nc = transpose ? Q : T;
    const float* __restrict__ cost = cost_all + cost_off[b];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int64_t* oi = idx_i + (long)b * Mmax;
    int64_t* oj = idx_j + (long)b * Mmax;
    for (int i = tid; i < Mmax; i += NT) { oi[i] = 0; oj[i] = 0; }
    if (nr == 0) { if (tid == 0) status[b] = 0; return; }

    // LDS carve: slots[2][16] | u[nr] f64 | col4row[nr] | row4col[nc] | remaining[nc] | path[nc]
    WgSlot* slots = reinterpret_cast<WgSlot*>(lds);
    double* u = reinterpret_cast<double*>(slots + 2 * NW);
    int* col4row = reinterpret_cast<int*>(u + nr);
    int* row4col = col4row + nr;
    int* remaining = row4col + nc;
    int* pathl = remaining + nc;
    __shared__ int s_bad;
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < nr; i += NT) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = tid; j < nc; j += NT) row4col[j] = -1;
    __syncthreads();
    {       // validity scan (scipy raises ValueError on NaN / -inf entries); 8 loads in flight per thread
        int bad = 0;
        const long ntot = (long)nr * nc;
        for (long k0 = 0; k0 < ntot; k0 += (long)NT * 8) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = cost[min(k0 + tid + (long)NT * q, ntot - 1)];
#pragma unroll
            for (int q = 0; q < 8; ++q) bad |= (t[q] != t[q] || t[q] == -INFINITY) ? 1 : 0;
        }
        if (bad) atomicOr(&s_bad, 1);
    }
    __syncthreads();
    if (s_bad) { if (tid == 0) status[b] = 2; return; }

    double v[CPL], spc[CPL];
    int path[CPL], r4c[CPL];
    unsigned key[CPL];
    unsigned valid = 0;
#pragma unroll
    for (int c = 0; c < CPL; ++c) { v[c] = 0.0; spc[c] = INFINITY; path[c] = -1; r4c[c] = -1; key[c] = 0xffffffffu; if (tid + NT * c < nc) valid |= 1u << c; }
    int parity = 0;

    // workgroup-wide lexicographic arg-min of the per-thread candidates (bval, bkey; best_c = the owning column slot).  Returns the winner's
    // value / key in (bval, bkey), its column in jstar and row4col[jstar] in rnext; `mine` = this thread owns the winning column.
    auto wg_lexmin = [&](double& bval, unsigned& bkey, int best_c, int& jstar, int& rnext, bool& mine) __attribute__((always_inline)) {
        const double my_val = bval;
        const unsigned my_key = bkey;
        wave_lexmin<true>(bval, bkey);
        const bool wmine = (my_key == bkey) && (my_val == bval) && (bkey != 0xffffffffu);
        const unsigned long long bal = __ballot(wmine);
        const int wl = bal ? __builtin_ctzll(bal) : 0;
        int my_r = r4c[0];
#pragma unroll
        for (int c = 1; c < CPL; ++c) my_r = (best_c == c) ? r4c[c] : my_r;
        const int wj = __builtin_amdgcn_readlane(tid + NT * best_c, wl);
        const int wr = __builtin_amdgcn_readlane(my_r, wl);
        WgSlot* sl = slots + parity * NW;
        if (lane == 0) { sl[wv].val = bval; sl[wv].key = bkey; sl[wv].jstar = wj; sl[wv].rnext = wr; }
        __syncthreads();
        double gv = sl[0].val;
        unsigned gk = sl[0].key;
        int gj = sl[0].jstar, gr = sl[0].rnext;
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) {
            const double ov = sl[w2].val;
            const unsigned ok = sl[w2].key;
            const bool take = (ov < gv) | ((ov == gv) & (ok < gk));
            gv = take ? ov : gv;
            gk = take ? ok : gk;
            gj = take ? sl[w2].jstar : gj;
            gr = take ? sl[w2].rnext : gr;
        }
        parity ^= 1;
        mine = wmine && (my_key == gk) && (my_val == gv);
        bval = gv; bkey = gk; jstar = gj; rnext = gr;
    };

    for (int cur = 0; cur < nr; ++cur) {
        // ---- fast path (as lsap_wave_kernel): the first scan step of the row already ends on an unassigned column -> the augmenting path
        // is the single edge (cur, j*), u[cur] += min, no column dual moves, no search state is set up
        {
            const float* crow0 = cost + (long)cur * nc;
            const double ui0 = u[cur];
            double bval = INFINITY;
            unsigned bkey = 0xffffffffu;
            int best_c = 0;
            float cv[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) cv[c] = crow0[((valid >> c) & 1u) ? tid + NT * c : 0];
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const bool a = (valid >> c) & 1u;
                const double r = ((0.0 + (double)cv[c]) - ui0) - v[c];
                const double s = (a & (r < INFINITY)) ? r : INFINITY;
                const unsigned p0 = (unsigned)(nc - 1 - (tid + NT * c));
                const unsigned k = (r4c[c] == -1) ? (0x7fffffffu - p0) : (0x80000000u + p0);
                const bool better = a & ((s < bval) | ((s == bval) & (k < bkey)));
                bval = better ? s : bval;
                bkey = better ? k : bkey;
                best_c = better ? c : best_c;
            }
            int jstar, rnext;
            bool mine;
            wg_lexmin(bval, bkey, best_c, jstar, rnext, mine);
            if (bkey != 0xffffffffu && bval != INFINITY && (bkey & 0x80000000u) == 0u) {
                if (tid == 0) {
                    u[cur] = ui0 + bval;
                    row4col[jstar] = cur;
                    col4row[cur] = jstar;
                }
#pragma unroll
                for (int c = 0; c < CPL; ++c) r4c[c] = (mine && best_c == c) ? cur : r4c[c];
                continue;
            }
        }
        unsigned sc = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            spc[c] = INFINITY;
            const unsigned p0 = (unsigned)(nc - 1 - (tid + NT * c));
            key[c] = (r4c[c] == -1) ? (0x7fffffffu - p0) : (0x80000000u + p0);
        }
        for (int it = tid; it < nc; it += NT) remaining[it] = nc - 1 - it;      // (first read: behind the first step's barrier)
        int i = cur, num_remaining = nc, sink = -1;
        double min_val = 0.0;
        while (true) {
            const int last = num_remaining - 1;
            const float* crow = cost + (long)i * nc;
            const double ui = u[i];
            double bval = INFINITY;
            unsigned bkey = 0xffffffffu;
            int best_c = 0;
            const unsigned act = valid & ~sc;
            float cv[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) cv[c] = crow[((valid >> c) & 1u) ? tid + NT * c : 0];
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const bool a = (act >> c) & 1u;
                const double r = ((min_val + (double)cv[c]) - ui) - v[c];
                const bool upd = a & (r < spc[c]);
                spc[c] = upd ? r : spc[c];
                path[c] = upd ? i : path[c];
                const bool better = a & ((spc[c] < bval) | ((spc[c] == bval) & (key[c] < bkey)));
                bval = better ? spc[c] : bval;
                bkey = better ? key[c] : bkey;
                best_c = better ? c : best_c;
            }
            int jstar, inext;
            bool mine;
            wg_lexmin(bval, bkey, best_c, jstar, inext, mine);
            if (bkey == 0xffffffffu || bval == INFINITY) { sink = -2; break; }
            min_val = bval;
            const bool j_unassigned = (bkey & 0x80000000u) == 0u;
            const int best_it = j_unassigned ? (int)(0x7fffffffu - bkey) : (int)(bkey - 0x80000000u);
            sc |= mine ? (1u << best_c) : 0u;
            // swap-removal from `remaining` (position best_it): the column that sat last moves there.  Every write of earlier steps is ordered
            // before this read by the barrier inside wg_lexmin; this step's write (thread 0) hits position `last` only with the value it holds
            const int jlast = remaining[last];
            if (tid == 0) remaining[best_it] = jlast;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const unsigned nk = (r4c[c] == -1) ? (0x7fffffffu - (unsigned)best_it) : (0x80000000u + (unsigned)best_it);
                key[c] = (tid + NT * c == jlast) ? nk : key[c];
            }
            num_remaining = last;
            if (j_unassigned) { sink = jstar; break; }
            i = inext;
        }
        if (sink == -2) { if (tid == 0) status[b] = 1; return; }
        // ---- dual update (visited rows other than cur = row4col[j] of the visited non-sink columns: distinct rows, no conflicts)
        if (tid == 0) u[cur] += min_val;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = tid + NT * c;
            if ((sc >> c) & 1u) {
                const double delta = min_val - spc[c];
                v[c] -= delta;
                if (j != sink) u[r4c[c]] += delta;
            }
            if (j < nc) pathl[j] = path[c];
        }
        __syncthreads();
        if (tid == 0) {      // augment (short sequential walk)
            int j = sink;
            while (true) {
                const int ii = pathl[j];
                row4col[j] = ii;
                const int t = col4row[ii];
                col4row[ii] = j;
                j = t;
                if (ii == cur) break;
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CPL; ++c)
            if ((valid >> c) & 1u) r4c[c] = row4col[tid + NT * c];
    }
    __syncthreads();
    if (!transpose) {
        for (int i = tid; i < nr; i += NT) { oi[i] = i; oj[i] = col4row[i]; }
    } else {
        for (int j = tid; j < nc; j += NT) remaining[j] = (row4col[j] != -1) ? 1 : 0;
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int j = 0; j < nc; ++j) { const int f = remaining[j]; remaining[j] = run; run += f; }
        }
        __syncthreads();
        for (int j = tid; j < nc; j += NT)
            if (row4col[j] != -1) { oi[remaining[j]] = j; oj[remaining[j]] = row4col[j]; }
    }
    if (tid == 0) status[b] = 0;
}

inline size_t lds_bytes(int nc_cap) { return (size_t)nc_cap * (3 * 8 + 4 * 4 + 2) + 16; }
inline int round_cap(int nc) { return (nc + 15) & ~15; }

}  // namespace

extern "C" int cdetr_match_cost(const float* logits, int32_t ncls, const float* boxes, const float* tgt,
                                const int32_t* tgt_off, const int64_t* cost_off, int32_t B, int32_t Q, float w_class,
                                float w_bbox, float w_giou, float* cost, void* stream) {
    CDETR_CHECK_ARG(logits && boxes && tgt_off && cost_off && cost && B > 0 && Q > 0 && ncls > 0, "cdetr_match_cost: bad args");
    dim3 grid(64, B);
    hipLaunchKernelGGL(match_cost_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logits, ncls, boxes, tgt,
                       tgt_off, cost_off, Q, w_class, w_bbox, w_giou, cost);
    return cdetr_launch_status("cdetr_match_cost");
}

extern "C" int cdetr_lsap(const float* cost, const int64_t* cost_off, const int32_t* tgt_off, int32_t B, int32_t Q,
                          int32_t nc_max, int32_t Mmax, int64_t* idx_i, int64_t* idx_j, int32_t* status, void* stream) {
    CDETR_CHECK_ARG(cost && cost_off && tgt_off && idx_i && idx_j && status && B > 0 && Q > 0 && Mmax > 0 && nc_max >= Q,
                    "cdetr_lsap: bad args");
    const int cap = round_cap(nc_max);
    const size_t bytes = lds_bytes(cap);
    if (bytes > 158 * 1024) {
        cdetr_set_error("cdetr_lsap: max(Q, T) = %d exceeds the LDS-resident solver's capacity (3800)", nc_max);
        return CDETR_ERR_UNSUPPORTED;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (nc_max <= 1024 && cdetr_tune_env("CDETR_LSAP_GENERIC") == nullptr) {
        // single-wave register-resident solver; nr <= Mmax rows, nc <= nc_max columns
        const size_t state = (size_t)Mmax * 12 + (size_t)nc_max * 12 + 64;
        const size_t with_cost = state + (size_t)Mmax * nc_max * 4;
        static const int lds_env = getenv("CDETR_LSAP_COST_LDS") ? atoi(getenv("CDETR_LSAP_COST_LDS")) : 1;      // A/B: 0 = cost matrix read from L2 (small LDS footprint)
        const bool cost_lds = lds_env && with_cost <= 156 * 1024;
        const size_t wb = cost_lds ? with_cost : state;
        auto go = [&](auto kern) {
            if (wb > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wb);
            static const int prio = getenv("CDETR_LSAP_PRIO") ? atoi(getenv("CDETR_LSAP_PRIO")) : 1;      // A/B: s_setprio 3 in the solve
            hipLaunchKernelGGL(kern, dim3(B), dim3(64), wb, st, cost, cost_off, tgt_off, Q, Mmax, idx_i, idx_j, status, prio);
        };
        // columns per lane = ceil(nc_max / 64): the per-iteration column scan is unrolled exactly that far
        if (nc_max <= 128) { if (cost_lds) go(lsap_wave_kernel<2, true>); else go(lsap_wave_kernel<2, false>); }
        else if (nc_max <= 256) { if (cost_lds) go(lsap_wave_kernel<4, true>); else go(lsap_wave_kernel<4, false>); }
        else if (nc_max <= 320) {
            static const bool shfl = getenv("CDETR_LSAP_SHFL") != nullptr;      // A/B knob: arg-min over ds_bpermute instead of DPP
            if (cost_lds && shfl) go(lsap_wave_kernel<5, true, false>);
            else if (cost_lds) go(lsap_wave_kernel<5, true>);
            else go(lsap_wave_kernel<5, false>);
        }
        else if (nc_max <= 512) { if (cost_lds) go(lsap_wave_kernel<8, true>); else go(lsap_wave_kernel<8, false>); }
        else { if (cost_lds) go(lsap_wave_kernel<16, true>); else go(lsap_wave_kernel<16, false>); }
        return cdetr_launch_status("cdetr_lsap");
    }
    if (nc_max <= 1024) {
        if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lsap_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        hipLaunchKernelGGL(lsap_kernel<64>, dim3(B), dim3(64), bytes, st, cost, cost_off, tgt_off, Q, Mmax, idx_i, idx_j, status, cap);
    } else if (nc_max <= 4096 && cdetr_tune_env("CDETR_LSAP_GENERIC") == nullptr) {
        // crowded images: the register-resident solver over the 16 waves of one workgroup (rows <= Mmax, columns <= nc_max)
        const size_t wb = 2 * 16 * sizeof(WgSlot) + (size_t)Mmax * 12 + (size_t)nc_max * 12 + 64;
        if (wb > 64 * 1024) {        // (Q ~ 1500 queries at 3800 targets; lsap_kernel<256> used to take such shapes)
            const void* fn = nc_max <= 2048 ? reinterpret_cast<const void*>(lsap_wg_kernel<2>) : reinterpret_cast<const void*>(lsap_wg_kernel<4>);
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wb) != hipSuccess) {
                (void)hipGetLastError();
                cdetr_set_error("cdetr_lsap: %zu bytes of LDS for %d x %d problems exceed the device limit", wb, Mmax, nc_max);
                return CDETR_ERR_UNSUPPORTED;
            }
        }
        if (nc_max <= 2048) hipLaunchKernelGGL(lsap_wg_kernel<2>, dim3(B), dim3(1024), wb, st, cost, cost_off, tgt_off, Q, Mmax, idx_i, idx_j, status);
        else hipLaunchKernelGGL(lsap_wg_kernel<4>, dim3(B), dim3(1024), wb, st, cost, cost_off, tgt_off, Q, Mmax, idx_i, idx_j, status);
    } else {
        if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lsap_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        hipLaunchKernelGGL(lsap_kernel<256>, dim3(B), dim3(256), bytes, st, cost, cost_off, tgt_off, Q, Mmax, idx_i, idx_j, status, cap);
    }
    return cdetr_launch_status("cdetr_lsap");
}
