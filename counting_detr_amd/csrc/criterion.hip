// criterion.hip -- SetCriterion (A2/models/anchor_detr.py:143-367: labels / boxes / cardinality / vars) as two launches.
//
// cdetr_criterion_fwd: ONE workgroup walks the whole batch (B*Q query rows, sum_b min(Q, T_b) matched pairs -- a few
// thousand scalars): it evaluates the six reported scalars AND the gradient of each differentiable loss w.r.t. its
// inputs, so the backward is a single scaled sum (cdetr_criterion_bwd).  The reference runs ~130 tiny tensor kernels
// here forward and ~100 backward.  Expressions follow the reference's autograd graph term by term:
//   loss_ce        sigmoid_focal_loss (A2/models/segmentation.py:198-223) .mean(1).sum() / num_boxes * Q
//   class_error    100 - accuracy of the matched queries (A2/util/misc.py:436-452, first-maximum argmax)
//   cardinality    mean_b | #{q: argmax != last class} - T_b |                      (:199-211, no gradient)
//   loss_bbox      sum |src - tgt| / num_boxes                                       (:213-225)
//   loss_giou      sum (1 - GIoU(xyxy(src), xyxy(tgt))) / num_boxes                  (:226-232; max/min ties split their
//                  gradient evenly and clamp(min=0) passes it at 0, like torch)
//   loss_variance  sum_k (mean_j|dw_j| / |v0_k| + |log v0_k| + same for h) / num_boxes (:264-289; log of a negative variance
//                  gives NaN in the loss and, as in torch, a zero contribution in the gradient of the |log| term)
// All reductions are two-level and ordered (per-wave shuffle tree, then a serial sum over the waves): bit-reproducible.
#include "../../include/cdetr_hip.h"
#include "common.h"

namespace {

constexpr int NRED = 12;
enum { R_CE = 0, R_L1, R_GIOU, R_CORRECT, R_DW, R_DH, R_IW, R_IH, R_LOGS, R_CARD, R_SPARE0, R_SPARE1 };

__device__ __forceinline__ float sgnf(float x) { return (float)((x > 0.f) - (x < 0.f)); }   // 0 for NaN, like torch.sign
__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

// block-wide ordered sum of `v` into red[slot] (called by all threads; ends with a barrier)
__device__ __forceinline__ void block_sum(float v, float* wred, float* red, int slot, int tid) {
    v = wave_sum(v);
    const int nw = blockDim.x >> 6;
    if ((tid & 63) == 0) wred[(tid >> 6) * NRED + slot] = v;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += wred[w * NRED + slot];
        red[slot] = s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void criterion_fwd_kernel(const cdetr_criterion_desc d) {
    __builtin_amdgcn_s_setprio(3);                   // one workgroup on the step's critical path (see lsap_wave_kernel)
    extern __shared__ int tcls[];                    // [B*Q] target class of every query (num_classes = no object)
    __shared__ float wred[4 * NRED], red[NRED];
    __shared__ int card[64];                         // per-image count of "object" queries (B <= 64)
    const int tid = threadIdx.x;
    const int B = d.B, Q = d.Q, C = d.C, BQ = B * Q;
    const float nb = d.num_boxes[0];
    const float inv_nb = 1.f / nb;

    for (int i = tid; i < BQ; i += 256) tcls[i] = d.num_classes;
    for (int i = tid; i < B; i += 256) card[i] = 0;
    for (int i = tid; i < BQ * C; i += 256) d.g_logits[i] = 0.f;
    for (int i = tid; i < BQ * 4; i += 256) { d.g_l1[i] = 0.f; d.g_giou[i] = 0.f; d.g_var_box[i] = 0.f; }
    for (int i = tid; i < BQ * 2; i += 256) d.g_vars[i] = 0.f;
    __syncthreads();
    // ---- matched pairs -> target classes
    int K = 0;                                       // total number of pairs (uniform)
    for (int b = 0; b < B; ++b) K += min(Q, d.tgt_off[b + 1] - d.tgt_off[b]);
    for (int s = tid; s < B * d.Mmax; s += 256) {
        const int b = s / d.Mmax, m = s - b * d.Mmax;
        const int Mb = min(Q, d.tgt_off[b + 1] - d.tgt_off[b]);
        if (m < Mb) {
            const int q = (int)d.idx_i[s], t = d.tgt_off[b] + (int)d.idx_j[s];
            tcls[b * Q + q] = (int)d.tgt_labels[t];
        }
    }
    __syncthreads();
    // ---- focal loss over every (b, q, c) + cardinality
    float ce_sum = 0.f;
    for (int r = tid; r < BQ; r += 256) {
        const int tc = tcls[r];
        float best = -INFINITY;
        int arg = 0;
        for (int c = 0; c < C; ++c) {
            const float x = d.logits[(long)r * C + c];
            if (x > best) { best = x; arg = c; }
            const float t = (c == tc) ? 1.f : 0.f;
            const float p = 1.f / (1.f + expf(-x));
            const float ce = softplusf(x) - x * t;                     // BCE with logits
            const float pt = p * t + (1.f - p) * (1.f - t);
            const float m = 1.f - pt;
            const float at = d.alpha * t + (1.f - d.alpha) * (1.f - t);
            ce_sum += at * ce * m * m;
            const float dpt = (2.f * t - 1.f) * p * (1.f - p);
            d.g_logits[(long)r * C + c] = at * ((p - t) * m * m - ce * 2.f * m * dpt) * inv_nb;
        }
        if (arg != C - 1) atomicAdd(&card[r / Q], 1);
    }
    block_sum(ce_sum, wred, red, R_CE, tid);
    // ---- matched pairs: L1, GIoU (+ gradients), accuracy, statistics of the variance loss
    float l1 = 0.f, gl = 0.f, correct = 0.f, sdw = 0.f, sdh = 0.f, siw = 0.f, sih = 0.f, slog = 0.f;
    for (int s = tid; s < B * d.Mmax; s += 256) {
        const int b = s / d.Mmax, m = s - b * d.Mmax;
        const int Mb = min(Q, d.tgt_off[b + 1] - d.tgt_off[b]);
        if (m >= Mb) continue;
        const int q = (int)d.idx_i[s], t = d.tgt_off[b] + (int)d.idx_j[s];
        const long r = (long)b * Q + q;
        const float4 sb = *reinterpret_cast<const float4*>(d.boxes + r * 4);
        const float4 tb = *reinterpret_cast<const float4*>(d.tgt_boxes + (long)t * 4);
        // L1
        const float e0 = sb.x - tb.x, e1 = sb.y - tb.y, e2 = sb.z - tb.z, e3 = sb.w - tb.w;
        l1 += fabsf(e0) + fabsf(e1) + fabsf(e2) + fabsf(e3);
        *reinterpret_cast<float4*>(d.g_l1 + r * 4) = make_float4(sgnf(e0) * inv_nb, sgnf(e1) * inv_nb, sgnf(e2) * inv_nb, sgnf(e3) * inv_nb);
        // GIoU on xyxy
        const float x1 = sb.x - 0.5f * sb.z, y1 = sb.y - 0.5f * sb.w, x2 = sb.x + 0.5f * sb.z, y2 = sb.y + 0.5f * sb.w;
        const float u1 = tb.x - 0.5f * tb.z, v1 = tb.y - 0.5f * tb.w, u2 = tb.x + 0.5f * tb.z, v2 = tb.y + 0.5f * tb.w;
        const float a1 = (x2 - x1) * (y2 - y1), a2 = (u2 - u1) * (v2 - v1);
        const float ltx = fmaxf(x1, u1), lty = fmaxf(y1, v1), rbx = fminf(x2, u2), rby = fminf(y2, v2);
        const float iw0 = rbx - ltx, ih0 = rby - lty;
        const float iw = fmaxf(iw0, 0.f), ih = fmaxf(ih0, 0.f);
        const float inter = iw * ih;
        const float uni = a1 + a2 - inter;
        const float iou = inter / uni;
        const float ex1 = fminf(x1, u1), ey1 = fminf(y1, v1), ex2 = fmaxf(x2, u2), ey2 = fmaxf(y2, v2);
        const float cw0 = ex2 - ex1, ch0 = ey2 - ey1;
        const float cw = fmaxf(cw0, 0.f), ch = fmaxf(ch0, 0.f);
        const float area = cw * ch;
        const float giou = iou - (area - uni) / area;
        gl += 1.f - giou;
        // selection weights of max / min w.r.t. the source coordinate (ties split evenly), clamp passes at >= 0
        auto wgt = [](float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); };      // d max(a,b) / da
        const float ciw = iw0 >= 0.f ? 1.f : 0.f, cih = ih0 >= 0.f ? 1.f : 0.f;
        const float ccw = cw0 >= 0.f ? 1.f : 0.f, cch = ch0 >= 0.f ? 1.f : 0.f;
        // d inter / d(x1, y1, x2, y2)
        const float di_x1 = -ciw * wgt(x1, u1) * ih, di_x2 = ciw * wgt(u2, x2) * ih;           // d min(x2,u2)/dx2 = [x2 < u2] (+ tie)
        const float di_y1 = -cih * wgt(y1, v1) * iw, di_y2 = cih * wgt(v2, y2) * iw;
        // d a1
        const float da_x1 = -(y2 - y1), da_x2 = (y2 - y1), da_y1 = -(x2 - x1), da_y2 = (x2 - x1);
        // d area (enclosing box)
        const float dA_x1 = -ccw * wgt(u1, x1) * ch, dA_x2 = ccw * wgt(x2, u2) * ch;           // d min(x1,u1)/dx1 = [x1 < u1] (+ tie)
        const float dA_y1 = -cch * wgt(v1, y1) * cw, dA_y2 = cch * wgt(y2, v2) * cw;
        auto dg = [&](float di, float da, float dA) {
            const float du = da - di;
            const float diou = (di * uni - inter * du) / (uni * uni);
            const float dra = (du * area - uni * dA) / (area * area);      // d (union / area)
            return diou + dra;                                              // giou = iou - 1 + union / area
        };
        const float g1 = dg(di_x1, da_x1, dA_x1), g2 = dg(di_y1, da_y1, dA_y1), g3 = dg(di_x2, da_x2, dA_x2), g4 = dg(di_y2, da_y2, dA_y2);
        // loss = (1 - giou) / nb ; chain to (cx, cy, w, h)
        *reinterpret_cast<float4*>(d.g_giou + r * 4) = make_float4(-(g1 + g3) * inv_nb, -(g2 + g4) * inv_nb, -0.5f * (g3 - g1) * inv_nb,
                                                                   -0.5f * (g4 - g2) * inv_nb);
        // accuracy of the matched query
        {
            float best = -INFINITY;
            int arg = 0;
            for (int c = 0; c < C; ++c) {
                const float x = d.logits[r * C + c];
                if (x > best) { best = x; arg = c; }
            }
            correct += (arg == (int)d.tgt_labels[t]) ? 1.f : 0.f;
        }
        // variance statistics
        const float v0 = d.vars[r * 2], v1v = d.vars[r * 2 + 1];
        sdw += fabsf(e2);
        sdh += fabsf(e3);
        siw += 1.f / fabsf(v0);
        sih += 1.f / fabsf(v1v);
        slog += fabsf(logf(v0)) + fabsf(logf(v1v));
    }
    block_sum(l1, wred, red, R_L1, tid);
    block_sum(gl, wred, red, R_GIOU, tid);
    block_sum(correct, wred, red, R_CORRECT, tid);
    block_sum(sdw, wred, red, R_DW, tid);
    block_sum(sdh, wred, red, R_DH, tid);
    block_sum(siw, wred, red, R_IW, tid);
    block_sum(sih, wred, red, R_IH, tid);
    block_sum(slog, wred, red, R_LOGS, tid);
    const float Kf = (float)K;
    const float mw = red[R_DW] / Kf, mh = red[R_DH] / Kf;            // NaN when K == 0 (never used then)
    // ---- variance gradients (need the batch-wide means)
    for (int s = tid; s < B * d.Mmax; s += 256) {
        const int b = s / d.Mmax, m = s - b * d.Mmax;
        const int Mb = min(Q, d.tgt_off[b + 1] - d.tgt_off[b]);
        if (m >= Mb) continue;
        const int q = (int)d.idx_i[s], t = d.tgt_off[b] + (int)d.idx_j[s];
        const long r = (long)b * Q + q;
        const float e2 = d.boxes[r * 4 + 2] - d.tgt_boxes[(long)t * 4 + 2], e3 = d.boxes[r * 4 + 3] - d.tgt_boxes[(long)t * 4 + 3];
        d.g_var_box[r * 4 + 2] = sgnf(e2) / Kf * red[R_IW] * inv_nb;
        d.g_var_box[r * 4 + 3] = sgnf(e3) / Kf * red[R_IH] * inv_nb;
        const float v0 = d.vars[r * 2], v1v = d.vars[r * 2 + 1];
        d.g_vars[r * 2] = (-mw * sgnf(v0) / (v0 * v0) + sgnf(logf(v0)) / v0) * inv_nb;
        d.g_vars[r * 2 + 1] = (-mh * sgnf(v1v) / (v1v * v1v) + sgnf(logf(v1v)) / v1v) * inv_nb;
    }
    if (tid == 0) {
        float cerr = 0.f;
        for (int b = 0; b < B; ++b) cerr += fabsf((float)card[b] - (float)(d.tgt_off[b + 1] - d.tgt_off[b]));
        d.losses[0] = red[R_CE] / (float)Q * inv_nb * (float)Q;                                  // .mean(1).sum() / nb * Q
        d.losses[1] = K > 0 ? 100.f - red[R_CORRECT] * (100.f / Kf) : 100.f;
        d.losses[2] = cerr / (float)B;
        d.losses[3] = red[R_L1] * inv_nb;
        d.losses[4] = red[R_GIOU] * inv_nb;
        d.losses[5] = K > 0 ? (mw * red[R_IW] + mh * red[R_IH] + red[R_LOGS]) * inv_nb : 0.f;
        if (d.loss_weights) {                          // the weighted total (A2/engine.py:37): one more element, no extra launches
            float t = 0.f;
            for (int k = 0; k < 6; ++k) t += d.loss_weights[k] != 0.f ? d.loss_weights[k] * d.losses[k] : 0.f;
            d.losses[6] = t;
        }
    }
}

// d_logits = g[0] * G_ce ; d_boxes = g[3] * G_l1 + g[4] * G_giou + g[5] * G_varbox ; d_vars = g[5] * G_vars
__global__ __launch_bounds__(256) void criterion_bwd_kernel(const float* __restrict__ g6, const float* __restrict__ gt,
                                                            const float* __restrict__ lw, const float* __restrict__ g_logits,
                                                            const float* __restrict__ g_l1, const float* __restrict__ g_giou,
                                                            const float* __restrict__ g_vb, const float* __restrict__ g_vars,
                                                            float* __restrict__ d_logits, float* __restrict__ d_boxes,
                                                            float* __restrict__ d_vars, const int BQ, const int C) {
    const float t = (gt && lw) ? gt[0] : 0.f;
    auto eff = [&](int k) { return (g6 ? g6[k] : 0.f) + ((gt && lw) ? t * lw[k] : 0.f); };
    const float gce = eff(0), gl1 = eff(3), ggi = eff(4), gva = eff(5);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < BQ * C; i += gridDim.x * 256) d_logits[i] = gce * g_logits[i];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < BQ * 4; i += gridDim.x * 256)
        d_boxes[i] = gl1 * g_l1[i] + ggi * g_giou[i] + gva * g_vb[i];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < BQ * 2; i += gridDim.x * 256) d_vars[i] = gva * g_vars[i];
}

}  // namespace

extern "C" int cdetr_criterion_fwd(const cdetr_criterion_desc* dp, void* stream) {
    CDETR_CHECK_ARG(dp != nullptr, "cdetr_criterion_fwd: null descriptor");
    const cdetr_criterion_desc d = *dp;
    CDETR_CHECK_ARG(d.B > 0 && d.B <= 64 && d.Q > 0 && d.C > 0 && d.Mmax > 0, "cdetr_criterion_fwd: bad sizes (B <= 64)");
    CDETR_CHECK_ARG((long)d.B * d.Q * 4 <= 160 * 1024 - 4096, "cdetr_criterion_fwd: B*Q too large for one workgroup's LDS");
    CDETR_CHECK_ARG(d.logits && d.boxes && d.vars && d.tgt_boxes && d.tgt_labels && d.tgt_off && d.idx_i && d.idx_j && d.num_boxes &&
                    d.losses && d.g_logits && d.g_l1 && d.g_giou && d.g_var_box && d.g_vars, "cdetr_criterion_fwd: null pointer");
    const int bytes = d.B * d.Q * 4;
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(criterion_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) {
            cdetr_set_error("cdetr_criterion_fwd: hipFuncSetAttribute(%d): %s", bytes, hipGetErrorString(e));
            return CDETR_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(criterion_fwd_kernel, dim3(1), dim3(256), bytes, reinterpret_cast<hipStream_t>(stream), d);
    return cdetr_launch_status("cdetr_criterion_fwd");
}

extern "C" int cdetr_criterion_bwd(const float* g6, const float* g_total, const float* loss_weights, const float* g_logits, const float* g_l1,
                                   const float* g_giou, const float* g_var_box, const float* g_vars, float* d_logits, float* d_boxes,
                                   float* d_vars, int32_t BQ, int32_t C, void* stream) {
    CDETR_CHECK_ARG((g6 || (g_total && loss_weights)) && g_logits && g_l1 && g_giou && g_var_box && g_vars && d_logits && d_boxes && d_vars && BQ > 0 && C > 0,
                    "cdetr_criterion_bwd: bad args");
    int blocks = (BQ * 4 + 255) / 256;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(criterion_bwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g6, g_total, loss_weights,
                       g_logits, g_l1, g_giou, g_var_box, g_vars, d_logits, d_boxes, d_vars, BQ, C);
    return cdetr_launch_status("cdetr_criterion_bwd");
}
