// glue.hip -- the small "between the GEMMs" pieces of the step as single launches (each replaces 5-40 tensor-library launches of
// ~4 us: at two images per GPU the step is a chain of ~700 launches, so every one that disappears is time).
//
// cdetr_mask_prep        padding mask -> feature-level mask (nearest), its first row / column (the RCDA key masks), the normalised
//                        key positions of mask2pos and each image's un-padded extent   (A2/models/backbone.py:143,
//                        A2/models/transformer.py:497-503, row_column_decoupled_attention.py:238-249)
// cdetr_stem_pack        NCHW image -> zero-padded 4-channel NHWC buffer of the row-packed 7x7 stem (backbone.py ResNetBody.stem_rows)
// cdetr_exemplar_fwd/bwd exemplar feature = mean over the exemplar boxes of the layer4 feature at the box centre
//                        (A2/models/backbone.py:116-136; per image, or the reference's rects[0]-for-all rule)
// cdetr_aggr_weight_fwd/bwd  per-image effective 1x1 projection weight W1 + W2 * pf (ops.AggrProjFn: the concat-free form of
//                        cat([x, x * pf]) -> conv1x1, A2/models/backbone.py:132-136 + anchor_detr.py:119) and its backward
// cdetr_box_head_fwd/bwd boxes = sigmoid(tmp + [inverse_sigmoid(ref), 0, 0])   (A2/models/transformer.py:193-203,
//                        A2/util/misc.py:475-479; torch's clamp-backward conventions)
#include "../../include/cdetr_hip.h"
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------- mask prep
// one workgroup per image.  Nearest-neighbour source index as torch's upsample_nearest: min(int(dst * (in / out)), in - 1), float scale.
__global__ __launch_bounds__(256) void mask_prep_kernel(const uint8_t* __restrict__ mask, int H, int W, int h, int w,
                                                        uint8_t* __restrict__ m, uint8_t* __restrict__ mask_row,
                                                        uint8_t* __restrict__ mask_col, float* __restrict__ pos_row,
                                                        float* __restrict__ pos_col, float* __restrict__ extent) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint8_t* src = mask + (long)b * H * W;
    uint8_t* dst = m + (long)b * h * w;
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;
    for (int i = tid; i < h * w; i += 256) {
        const int y = i / w, x = i - y * w;
        const int yy = min((int)floorf(y * sy), H - 1), xx = min((int)floorf(x * sx), W - 1);
        dst[i] = src[(long)yy * W + xx] ? 1 : 0;
    }
    // first row / first column and the positions are re-derived from the SOURCE mask (no read-after-write of `dst` through L1)
    auto cell = [&](int y, int x) -> bool {
        const int yy = min((int)floorf(y * sy), H - 1), xx = min((int)floorf(x * sx), W - 1);
        return src[(long)yy * W + xx] != 0;
    };
    // (the cells of the first row / column are fetched by all threads at once and parked in LDS: the two scans below used to be chains of
    // ~50 dependent global loads each -- 20 us for a handful of bytes)
    __shared__ float free_row[1024], free_col[1024];
    for (int x = tid; x < w; x += 256) { const bool c = cell(0, x); mask_row[(long)b * w + x] = c ? 1 : 0; if (x < 1024) free_row[x] = c ? 0.f : 1.f; }     // mask[:, 0, :]
    for (int y = tid; y < h; y += 256) { const bool c = cell(y, 0); mask_col[(long)b * h + y] = c ? 1 : 0; if (y < 1024) free_col[y] = c ? 0.f : 1.f; }     // mask[:, :, 0]
    __syncthreads();
    const bool lds_ok = h <= 1024 && w <= 1024;
    if (tid == 64) {     // mask2pos: (cumsum(~mask) - 0.5) / total along each axis; tens of elements -- a serial scan is the cheapest form
        float ty = 0.f;
        for (int y = 0; y < h; ++y) ty += lds_ok ? free_col[y] : (cell(y, 0) ? 0.f : 1.f);
        float run = 0.f;
        for (int y = 0; y < h; ++y) { run += lds_ok ? free_col[y] : (cell(y, 0) ? 0.f : 1.f); pos_col[(long)b * h + y] = (run - 0.5f) / ty; }
        extent[2 * b] = ty;          // un-padded rows of this image, in feature cells
    }
    if (tid == 128) {
        float tx = 0.f;
        for (int x = 0; x < w; ++x) tx += lds_ok ? free_row[x] : (cell(0, x) ? 0.f : 1.f);
        float run = 0.f;
        for (int x = 0; x < w; ++x) { run += lds_ok ? free_row[x] : (cell(0, x) ? 0.f : 1.f); pos_row[(long)b * w + x] = (run - 0.5f) / tx; }
        extent[2 * b + 1] = tx;      // un-padded columns
    }
}

// ---------------------------------------------------------------------------------------------------- stem pack
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ img, float* __restrict__ xp, int B, int H, int W,
                                                        int Ha, int Wa, int py, int px) {
    const long total = (long)B * Ha * Wa;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int xa = (int)(i % Wa);
        long t = i / Wa;
        const int ya = (int)(t % Ha);
        const int b = (int)(t / Ha);
        const int y = ya - py, x = xa - px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const float* p = img + ((long)b * 3 * H + y) * W + x;
            v.x = p[0]; v.y = p[(long)H * W]; v.z = p[2L * H * W];
        }
        reinterpret_cast<float4*>(xp)[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------- exemplar feature
// grid (C / 256 rounded up, B).  rects [B][K][4] normalised xyxy; a row with x2 < 0 is an absent exemplar.
// per_image != 0: image b uses rects[b] scaled by extent[b] = (rows, cols); else every image uses rects[0] scaled by (h, w).
// Centre = truncation of ((x1 * w + x2 * w) / 2) in fp32, as the reference's int() on a float tensor.
__global__ __launch_bounds__(256) void exemplar_fwd_kernel(const float* __restrict__ x, const float* __restrict__ rects,
                                                           const float* __restrict__ extent, int per_image, int h, int w, int C, int K,
                                                           int* __restrict__ idx, float* __restrict__ inv_cnt, float* __restrict__ pf) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int rb = per_image ? b : 0;
    const float hv = per_image ? extent[2 * b] : (float)h, wv = per_image ? extent[2 * b + 1] : (float)w;
    float acc = 0.f;
    int cnt = 0;
    for (int k = 0; k < K; ++k) {
        const float* r = rects + ((long)rb * K + k) * 4;
        const float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
        int cell = -1;
        if (x2 >= 0.f) {
            const int xc = min(max((int)((x1 * wv + x2 * wv) / 2.f), 0), w - 1);
            const int yc = min(max((int)((y1 * hv + y2 * hv) / 2.f), 0), h - 1);
            cell = yc * w + xc;
            ++cnt;
            if (c < C) acc += x[((long)b * h * w + cell) * C + c];
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) idx[b * K + k] = cell;
    }
    const float ic = 1.f / (float)max(cnt, 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) inv_cnt[b] = ic;
    if (c < C) pf[(long)b * C + c] = acc * ic;       // K = 3: differs from torch's mean (sum * (1/3) on the GPU, sum / 3 on the CPU) by <= 1 ulp
}

__global__ __launch_bounds__(256) void exemplar_bwd_kernel(const float* __restrict__ dpf, const int* __restrict__ idx,
                                                           const float* __restrict__ inv_cnt, float* __restrict__ dx, int P, int C, int K) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float g = dpf[(long)b * C + c] * inv_cnt[b];
    for (int k = 0; k < K; ++k) {                    // two exemplars may share a cell: sequential adds by the same thread
        const int cell = idx[b * K + k];
        if (cell >= 0) dx[((long)b * P + cell) * C + c] += g;
    }
}

// ---------------------------------------------------------------------------------------------------- effective projection weight
// Weff[b][o][c] = W[o][c] + W[o][C + c] * pf[b][c];  WeffT[b][c][o] = the same, transposed (k-contiguous operand of the data gradient).
// 32x32 tiles through LDS so that both images are written in 128-byte runs.  grid (C/32, d/32, B).
__global__ __launch_bounds__(256) void aggr_weight_fwd_kernel(const float* __restrict__ W, const float* __restrict__ pf,
                                                              float* __restrict__ Weff, float* __restrict__ WeffT, int d, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, o0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 8 rows per pass
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int o = o0 + r, c = c0 + tx;
        float v = 0.f;
        if (o < d && c < C) {
            v = W[(long)o * 2 * C + c] + W[(long)o * 2 * C + C + c] * pf[(long)b * C + c];
            Weff[((long)b * d + o) * C + c] = v;
        }
        tile[r][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, o = o0 + tx;
        if (o < d && c < C) WeffT[((long)b * C + c) * d + o] = tile[tx][r];
    }
}

// From dWeff [B][d][C]:  gW[o][c] += sum_b dWeff;  gW[o][C + c] += sum_b dWeff * pf[b][c];  dpf[b][c] += sum_o dWeff[b][o][c] * W[o][C + c].
// grid (C / 256 rounded up, d / AGG_ROWS): thread = one column c over a slice of AGG_ROWS output rows, all images (B <= 16) in registers.
// (8 rows, unrolled: the loads of a slice are in flight together and the grid is 256 workgroups at d = 256, C = 2048 -- with 32-row slices
// the kernel was a chain of 32 dependent round trips on 64 workgroups: 56 us.)
constexpr int AGG_MAX_B = 16, AGG_ROWS = 8;
__global__ __launch_bounds__(256) void aggr_weight_bwd_kernel(const float* __restrict__ dWeff, const float* __restrict__ pf,
                                                              const float* __restrict__ W, float* __restrict__ gW, float* __restrict__ dpf,
                                                              int B, int d, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int o0 = blockIdx.y * AGG_ROWS, o1 = min(o0 + AGG_ROWS, d);
    float pfb[AGG_MAX_B], acc[AGG_MAX_B];
    for (int b = 0; b < B; ++b) { pfb[b] = pf[(long)b * C + c]; acc[b] = 0.f; }
#pragma unroll
    for (int oo = 0; oo < AGG_ROWS; ++oo) {
        const int o = o0 + oo;
        if (o >= o1) break;
        const float w2 = W[(long)o * 2 * C + C + c];
        float s1 = 0.f, s2 = 0.f;
        for (int b = 0; b < B; ++b) {
            const float v = dWeff[((long)b * d + o) * C + c];
            s1 += v;
            s2 += v * pfb[b];
            acc[b] += v * w2;
        }
        if (gW) {                                   // this thread is the only writer of these two elements
            gW[(long)o * 2 * C + c] += s1;
            gW[(long)o * 2 * C + C + c] += s2;
        }
    }
    for (int b = 0; b < B; ++b) atomicAdd(dpf + (long)b * C + c, acc[b]);
}

// ---------------------------------------------------------------------------------------------------- box head tail
__device__ __forceinline__ float inv_sigmoid(float x, float eps, float& dinv) {
    // A2/util/misc.py:475-479: x1 = clamp(x, 0, 1); log(clamp(x1, min=eps) / clamp(1 - x1, min=eps)).  torch's clamp backward lets
    // the gradient through where the input lies INSIDE the closed interval (x >= min && x <= max).
    const float x1 = fminf(fmaxf(x, 0.f), 1.f);
    const float a = fmaxf(x1, eps), bq = fmaxf(1.f - x1, eps);
    const bool in01 = (x >= 0.f) && (x <= 1.f);
    dinv = in01 ? ((x1 >= eps ? 1.f / a : 0.f) + ((1.f - x1) >= eps ? 1.f / bq : 0.f)) : 0.f;
    return logf(a / bq);
}

__global__ __launch_bounds__(256) void box_head_fwd_kernel(const float* __restrict__ tmp, const float* __restrict__ ref,
                                                           float* __restrict__ boxes, int M, int R) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float4 t = reinterpret_cast<const float4*>(tmp)[m];
    const float2 r = reinterpret_cast<const float2*>(ref)[m % R];
    float d0, d1;
    const float a0 = t.x + inv_sigmoid(r.x, 1e-5f, d0), a1 = t.y + inv_sigmoid(r.y, 1e-5f, d1);
    float4 o;
    o.x = 1.f / (1.f + expf(-a0)); o.y = 1.f / (1.f + expf(-a1)); o.z = 1.f / (1.f + expf(-t.z)); o.w = 1.f / (1.f + expf(-t.w));
    reinterpret_cast<float4*>(boxes)[m] = o;
}

// d_tmp = d_boxes * s (1 - s);  d_ref[m % R][j] (+)= d_tmp[j] * d inverse_sigmoid(ref_j)   (atomic when rows share a reference point)
__global__ __launch_bounds__(256) void box_head_bwd_kernel(const float* __restrict__ d_boxes, const float* __restrict__ boxes,
                                                           const float* __restrict__ ref, float* __restrict__ d_tmp,
                                                           float* __restrict__ d_ref, int M, int R) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float4 g = reinterpret_cast<const float4*>(d_boxes)[m];
    const float4 s = reinterpret_cast<const float4*>(boxes)[m];
    float4 o;
    o.x = g.x * s.x * (1.f - s.x); o.y = g.y * s.y * (1.f - s.y); o.z = g.z * s.z * (1.f - s.z); o.w = g.w * s.w * (1.f - s.w);
    reinterpret_cast<float4*>(d_tmp)[m] = o;
    if (d_ref) {
        const float2 r = reinterpret_cast<const float2*>(ref)[m % R];
        float d0, d1;
        inv_sigmoid(r.x, 1e-5f, d0);
        inv_sigmoid(r.y, 1e-5f, d1);
        if (M == R) { d_ref[2 * m] = o.x * d0; d_ref[2 * m + 1] = o.y * d1; }
        else { atomicAdd(d_ref + 2 * (m % R), o.x * d0); atomicAdd(d_ref + 2 * (m % R) + 1, o.y * d1); }
    }
}

inline int blocks_for(long n, int per = 256, int cap = 4096) {
    long b = (n + per - 1) / per;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int cdetr_mask_prep(const uint8_t* mask, int32_t B, int32_t H, int32_t W, int32_t h, int32_t w, uint8_t* m, uint8_t* mask_row,
                               uint8_t* mask_col, float* pos_row, float* pos_col, float* extent, void* stream) {
    CDETR_CHECK_ARG(mask && m && mask_row && mask_col && pos_row && pos_col && extent && B > 0 && H > 0 && W > 0 && h > 0 && w > 0,
                    "cdetr_mask_prep: bad args");
    hipLaunchKernelGGL(mask_prep_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), mask, H, W, h, w, m, mask_row,
                       mask_col, pos_row, pos_col, extent);
    return cdetr_launch_status("cdetr_mask_prep");
}

// Cross-stream ordering WITHOUT a host-visible event: a one-thread "signal" kernel in one stream's chain bumps a counter in device memory, a
// one-wave "wait" kernel at the head of another stream's work sleeps until it sees the bump (or a timeout).  engine.Trainer releases the next
// batch's frozen stage with it at the moment the Hungarian solve of the current step is about to start: the solve keeps its whole cost matrix
// in LDS (~144 KB = one compute unit's LDS) and must be resident BEFORE the stem / layer1 workgroups flood every CU, and an event cannot be
// recorded in the middle of a captured graph.  `seen` is the waiter's own count of bumps consumed (self-healing: it adopts the counter's value),
// the timeout makes a missing signal (or a waiter that shares its hardware queue with the signalling chain) a delay, never a hang.
namespace {
__global__ void flag_signal_kernel(int* flag) { __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void flag_wait_kernel(const int* flag, int* seen, long timeout_ticks, long post_ticks) {
    const int want = *seen + 1;
    const long t0 = wall_clock64();
    int cur;
    while ((cur = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - want < 0 && wall_clock64() - t0 < timeout_ticks)
        __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) *seen = (cur - want >= 0) ? cur : want;
    const long t1 = wall_clock64();                    // the signalling chain's NEXT kernel gets a head start before this stream's work begins
    while (wall_clock64() - t1 < post_ticks) __builtin_amdgcn_s_sleep(16);
}
__global__ void delay_kernel(long ticks) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace
extern "C" int cdetr_flag_signal(int32_t* flag, void* stream) {
    CDETR_CHECK_ARG(flag != nullptr, "cdetr_flag_signal: null flag");
    hipLaunchKernelGGL(flag_signal_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), flag);
    return cdetr_launch_status("cdetr_flag_signal");
}
extern "C" int cdetr_flag_wait(const int32_t* flag, int32_t* seen, int32_t timeout_us, int32_t post_us, void* stream) {
    CDETR_CHECK_ARG(flag && seen && timeout_us >= 0 && timeout_us <= 100000 && post_us >= 0 && post_us <= 1000,
                    "cdetr_flag_wait: bad args (timeout 0 .. 100000 us, post 0 .. 1000 us)");
    hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), flag, seen, (long)timeout_us * 100,
                       (long)post_us * 100);            // wall_clock64: 100 MHz
    return cdetr_launch_status("cdetr_flag_wait");
}
// One idle wavefront for `us` microseconds (s_sleep between reads of the 100 MHz wall clock): holds a stream back without occupying the chip
// (the stream-concurrency probe of engine.Trainer).
extern "C" int cdetr_delay(int32_t us, void* stream) {
    CDETR_CHECK_ARG(us >= 0 && us <= 100000, "cdetr_delay: 0 .. 100000 us");
    if (us == 0) return CDETR_OK;
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), (long)us * 100);      // wall_clock64: 100 MHz
    return cdetr_launch_status("cdetr_delay");
}

extern "C" int cdetr_stem_pack(const float* images, float* xp, int32_t B, int32_t H, int32_t W, int32_t Ha, int32_t Wa, int32_t pad_y,
                               int32_t pad_x, void* stream) {
    CDETR_CHECK_ARG(images && xp && B > 0 && H > 0 && W > 0 && Ha >= H + pad_y && Wa >= W + pad_x && pad_y >= 0 && pad_x >= 0 &&
                    (reinterpret_cast<uintptr_t>(xp) & 15) == 0, "cdetr_stem_pack: bad args");
    hipLaunchKernelGGL(stem_pack_kernel, dim3(blocks_for((long)B * Ha * Wa, 256, 8192)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       images, xp, B, H, W, Ha, Wa, pad_y, pad_x);
    return cdetr_launch_status("cdetr_stem_pack");
}

extern "C" int cdetr_exemplar_fwd(const float* x, const float* rects, const float* extent, int32_t per_image, int32_t B, int32_t h,
                                  int32_t w, int32_t C, int32_t K, int32_t* idx, float* inv_cnt, float* pf, void* stream) {
    CDETR_CHECK_ARG(x && rects && idx && inv_cnt && pf && B > 0 && h > 0 && w > 0 && C > 0 && K > 0 && (!per_image || extent),
                    "cdetr_exemplar_fwd: bad args");
    hipLaunchKernelGGL(exemplar_fwd_kernel, dim3((C + 255) / 256, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, rects, extent,
                       per_image, h, w, C, K, idx, inv_cnt, pf);
    return cdetr_launch_status("cdetr_exemplar_fwd");
}

extern "C" int cdetr_exemplar_bwd(const float* dpf, const int32_t* idx, const float* inv_cnt, float* dx, int32_t B, int32_t P, int32_t C,
                                  int32_t K, void* stream) {
    CDETR_CHECK_ARG(dpf && idx && inv_cnt && dx && B > 0 && P > 0 && C > 0 && K > 0, "cdetr_exemplar_bwd: bad args");
    hipLaunchKernelGGL(exemplar_bwd_kernel, dim3((C + 255) / 256, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dpf, idx, inv_cnt,
                       dx, P, C, K);
    return cdetr_launch_status("cdetr_exemplar_bwd");
}

extern "C" int cdetr_aggr_weight_fwd(const float* W, const float* pf, float* Weff, float* WeffT, int32_t B, int32_t d, int32_t C,
                                     void* stream) {
    CDETR_CHECK_ARG(W && pf && Weff && WeffT && B > 0 && d > 0 && C > 0, "cdetr_aggr_weight_fwd: bad args");
    hipLaunchKernelGGL(aggr_weight_fwd_kernel, dim3((C + 31) / 32, (d + 31) / 32, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W, pf,
                       Weff, WeffT, d, C);
    return cdetr_launch_status("cdetr_aggr_weight_fwd");
}

extern "C" int cdetr_aggr_weight_bwd(const float* dWeff, const float* pf, const float* W, float* gW, float* dpf, int32_t B, int32_t d,
                                     int32_t C, void* stream) {
    CDETR_CHECK_ARG(dWeff && pf && W && dpf && B > 0 && B <= AGG_MAX_B && d > 0 && C > 0, "cdetr_aggr_weight_bwd: bad args (B <= %d)", AGG_MAX_B);
    hipLaunchKernelGGL(aggr_weight_bwd_kernel, dim3((C + 255) / 256, (d + AGG_ROWS - 1) / AGG_ROWS), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dWeff,
                       pf, W, gW, dpf, B, d, C);
    return cdetr_launch_status("cdetr_aggr_weight_bwd");
}

extern "C" int cdetr_box_head_fwd(const float* tmp, const float* ref, float* boxes, int32_t M, int32_t R, void* stream) {
    CDETR_CHECK_ARG(tmp && ref && boxes && M > 0 && R > 0 && M % R == 0, "cdetr_box_head_fwd: bad args");
    hipLaunchKernelGGL(box_head_fwd_kernel, dim3((M + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), tmp, ref, boxes, M, R);
    return cdetr_launch_status("cdetr_box_head_fwd");
}

extern "C" int cdetr_box_head_bwd(const float* d_boxes, const float* boxes, const float* ref, float* d_tmp, float* d_ref, int32_t M,
                                  int32_t R, void* stream) {
    CDETR_CHECK_ARG(d_boxes && boxes && ref && d_tmp && M > 0 && R > 0 && M % R == 0, "cdetr_box_head_bwd: bad args");
    hipLaunchKernelGGL(box_head_bwd_kernel, dim3((M + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_boxes, boxes, ref,
                       d_tmp, d_ref, M, R);
    return cdetr_launch_status("cdetr_box_head_bwd");
}
