/* cdetr_hip.h -- C-ABI of libcdetr_hip.so: the MI355X (gfx950) kernels of the Counting-DETR hot path.
 *
 * The reference (VinAIResearch/Counting-DETR) is pure Python/PyTorch and has NO FFI of its own; each entry
 * point below replaces the torch-op sequence at the cited reference lines (A2/ = src/CountDETR_147_2nd_stage/).
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add at each site.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (cdetr_last_error() gives the thread-local message);
 *     nothing throws across the boundary;
 *   - every buffer is DEVICE memory allocated by the caller (torch caching allocator); the library never
 *     allocates, never synchronises, holds no mutable global state (re-entrant: it is called from the Python main
 *     thread in forward and from the autograd engine thread in backward);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); kernels are async;
 *   - all tensors are fp32 unless stated; activations are NHWC / row-major [rows][channels];
 *   - arithmetic: fp32 MFMA (v_mfma_f32_32x32x2_f32) -- exact fp32 products, fp32 accumulation.
 */
#ifndef CDETR_HIP_H
#define CDETR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDETR_ABI_VERSION 2

/* row-gather modes of cdetr_conv_geom */
#define CDETR_ROWS_DENSE 0      /* row(m) = m (linear layers, 1x1 stride-1 convs) */
#define CDETR_ROWS_CONV_FWD 1   /* m = output pixel; tap (ky,kx) -> input pixel (zero outside) */
#define CDETR_ROWS_CONV_DGRAD 2 /* m = input pixel;  tap (ky,kx) -> output pixel feeding it (transposed conv) */

typedef struct {
    int32_t mode;
    int32_t Ha, Wa; /* spatial dims of the GATHERED tensor: its row index is (n*Ha + y)*Wa + x            */
    int32_t Hc, Wc; /* spatial dims of the enumerated row space (m = (n*Hc + y)*Wc + x)                    */
    int32_t kh, kw, stride, pad, dil;
} cdetr_conv_geom;

/* C[m][n] = epilogue( sum_tap sum_k A[row(m,tap)][k] * Wt(n,tap,k) )
 * b_layout 0 : Wt(n,tap,k) = B[n*ldb + tap*K + k]          (weight [N][taps][K], k contiguous; forward)
 * b_layout 1 : Wt(n,tap,k) = B[(k*taps + tap)*ldb + n]     (weight [K][taps][N], n contiguous; data-gradient)
 * epilogue   : v = acc (+ bias[n]); v *= out_scale; v += resid[m][n]; if gate: v = gate[m][n] > 0 ? v : 0;
 *              if relu: v = max(v, 0)
 * w_scale    : optional per-output-channel scale of the WEIGHT (frozen-BN fold): index n for b_layout 0,
 *              index k for b_layout 1.
 * Replaces: F.conv2d + FrozenBatchNorm2d + ReLU (+residual) A2/models/resnet.py:140-160, backbone.py:50-60;
 *           F.linear at A2/models/row_column_decoupled_attention.py:165-208,311, transformer.py:412-439; and
 *           their autograd data-gradients.                                                                    */
typedef struct {
    int32_t M, N, K, taps;
    int32_t batch; /* grid.z; per-batch element strides sA/sB/sC (0 = shared); see batch_inner for two-level batches */
    int32_t b_layout;
    int32_t relu;
    int32_t precision; /* 0 = fp32 MFMA (exact fp32 products), 1 = split-bf16 x3 on the bf16 matrix pipe (~1e-5 rel);
                          2 = "bf16x2": B rounded to bf16, A split hi+lo, 2 MFMAs per product (~2^-9 rel per product, random);
                          3 = plain bf16: both operands rounded, 1 MFMA.  2 / 3 are meant for the BACKWARD passes (data / weight
                          gradients); kernels without a reduced-term form (few-row GEMMs, n-contiguous operands) run them as 1. */
    float out_scale;
    const float* A; int64_t lda, sA; /* A may be NULL under the same conditions as C below (the operand exists as its bf16 twin A16 only) */
    const float* B; int64_t ldb, sB;
    float* C; int64_t ldc, sC; /* C may be NULL when C16 is given and the operands are in the direct-to-LDS kernel's formats (A16, B16 or
                                * B_split, K % 64 == 0, 16-byte aligned): only the bf16 copy is written -- an inner activation gradient of
                                * a bottleneck is read by nothing but the next bf16 contractions */
    const float* w_scale;
    const float* bias;
    const float* resid; int64_t ldr;
    const float* gate; int64_t ldg;
    cdetr_conv_geom g;
    const void* B_split; /* optional (NULL = absent), b_layout 0 + precision 1 + batch 1 only: the SAME operand pre-split for the
                          * bf16x3 kernels, w_scale already folded in: row n = [taps*K/32 groups][hi 32 bf16 | lo 32 bf16], i.e.
                          * the byte offsets of (n, tap, k-group) equal those of the fp32 operand (ldb applies unchanged);
                          * written by cdetr_weight_mirror.  Kernels that cannot use it read B / w_scale instead.          */
    int32_t batch_inner; /* 0: batch item z sits at z*s.  > 0: z = outer*batch_inner + inner sits at inner*s + outer*s2      */
    int32_t flags;       /* (e.g. heads inside images: one launch for the per-head GEMMs of every image)
                          * flags: CDETR_GEMM_A_GROUPS / CDETR_GEMM_C_GROUPS below (0 = none)                                       */
    int64_t sA2, sB2, sC2;
    const void* A16;     /* optional: a bf16 TWIN of A (same lda / batch strides, in elements): with precision 3 the tile kernels read it
                          * instead of A -- half the operand bytes, no conversion at staging.  Needs lda and the strides to be multiples
                          * of 8 and a 16-byte aligned base; other kernel classes read A.                                             */
    void* C16;           /* optional (NULL = absent): a bf16 TWIN of C written by the same epilogue (same ldc / batch strides, in
                          * elements) -- the operand format of the plain-bf16 weight gradients (cdetr_wgrad_desc.dY16 / X16).      */
    void* splitk_ws;     /* optional (NULL = never split): device scratch that lets the tile kernels split the reduction of a problem with
                          * too few output tiles to fill 256 CUs (two images per GPU: 5000 rows x 256 channels = 158 workgroups) across
                          * 2-4 workgroups per tile; the partial tiles meet here and the last workgroup to arrive adds them IN SLICE ORDER
                          * (deterministic) and runs the fused epilogue.  Layout: [4096 int32 arrival counters | partial tiles]; the caller
                          * zeroes the counters ONCE (every call leaves them zero) and must not share one scratch between streams that
                          * may run cdetr_gemm concurrently.  16-byte aligned.
                          * ORDERING IS gfx950-SPECIFIC, not a HIP memory-model guarantee: the partial tiles travel as RELAXED agent-scope
                          * atomic stores / loads (sc1: they bypass the XCD-private L2s), `s_waitcnt vmcnt(0)` + the workgroup barrier put
                          * them before the relaxed agent-scope counter RMW.  An acquire / release pair would be the portable form and costs
                          * a write-back + invalidate of a whole 4 MiB L2 per workgroup (+80 us per launch when measured); the same
                          * protocol serves cdetr_gemm_dl's split form and cdetr_rcda_fwd_desc.ws.  tools/splitk_stress.py (4000 launches,
                          * bit-identical results, counters back at zero) is the regression check to re-run after a compiler / runtime
                          * update.                                                                                                  */
    int64_t splitk_ws_bytes; /* size of splitk_ws in bytes (>= 16 KiB + partial tiles; too small = fewer slices or none)            */
    const void* A16lo;   /* optional: the LO plane of A's split-bf16 form, lo = bf16(A - float(A16)) (same shape / strides as A16).  With
                          * A16 + A16lo + B_split the split-bf16 x3 product (precision 1) needs no conversion at all: the direct-to-LDS
                          * tile kernel (igemm_dl_kernel) copies operand tiles HBM -> LDS with global_load_lds and feeds the matrix
                          * pipe from there.  Same arithmetic as splitting the fp32 operand at staging (bit-identical products).       */
    const void* B16;     /* optional, b_layout 0: bf16(B * w_scale), same [N][taps*K] shape and ldb (in elements): what the plain-bf16
                          * (precision 3) direct-to-LDS kernel streams instead of the hi halves of B_split's interleaved groups.   */
    const void* gate16;  /* optional: bf16 twin of `gate` (same ldg, in elements): the direct-to-LDS kernel tests its sign instead of the fp32
                          * tensor's (bf16 rounding keeps sign and zero: the ReLU mask is identical) -- half the bytes of the epilogue's
                          * largest read in the backbone's data gradients.  `gate` must still be given (other kernel classes read it).  */
    void* C16lo;         /* optional: the LO plane of C written by the same epilogue, C16lo = bf16(C - float(C16)) (needs C16): the
                          * next layer's A16lo.  C itself may then be NULL when no consumer reads the fp32 tensor.                     */
} cdetr_gemm_desc;
/* cdetr_gemm_desc.flags (direct-to-LDS kernel, precision 1):
 * CDETR_GEMM_A_GROUPS: A16 holds A pre-split in INTERLEAVED groups, the format of B_split -- row m = [K/32 groups][hi 32 bf16 | lo 32 bf16],
 *   i.e. the byte offset of (m, k-group) equals that of the fp32 operand (lda applies unchanged, 4 bytes per element; lda % 32 == 0; A16lo
 *   unused).  A k-tile of the kernel is then ONE full 128-byte line per row; with the planes A16 / A16lo it is two half lines.
 * CDETR_GEMM_C_GROUPS: C16lo receives C in the same interleaved form ([N/32 groups][hi 32 | lo 32] per row, ldc applies unchanged,
 *   N % 32 == 0, ldc % 32 == 0) -- the next layer's grouped A operand; C16 (the plain hi twin) stays optional.                       */
#define CDETR_GEMM_A_GROUPS 1
#define CDETR_GEMM_C_GROUPS 2
/* CDETR_GEMM_PRIO: the launch's waves run at raised instruction priority (s_setprio): for launches of a step's MAIN chain that share the
 *   chip with another stream's throughput work (the backbone's data gradients beside the weight gradients); results are unaffected.   */
#define CDETR_GEMM_PRIO 4
/* CDETR_GEMM_RESID_GROUPS: `resid` points to a tensor in the interleaved-group form (what CDETR_GEMM_C_GROUPS wrote: a bottleneck's output kept
 *   ONCE as [hi 32 | lo 32] groups instead of fp32 + twin); ldr applies unchanged (4 bytes per element), ldr % 32 == 0, N % 32 == 0; the epilogue
 *   adds hi + lo.  Direct-to-LDS kernel only, like the other two group flags.                                                          */
#define CDETR_GEMM_RESID_GROUPS 8
/* CDETR_GEMM_GATE16_ONLY: `gate` is NOT an fp32 tensor (the gating activation exists as interleaved groups + its twin): only gate16 may be read,
 *   i.e. the problem must run on the direct-to-LDS kernel (CDETR_ERR_ARG otherwise); `gate` must still be non-NULL (it switches the gate on).  */
#define CDETR_GEMM_GATE16_ONLY 16
int cdetr_gemm(const cdetr_gemm_desc* d, void* stream);
/* The direct-to-LDS tile kernel (csrc/igemm_dl.hip) with an explicit configuration -- what cdetr_gemm picks by itself for problems
 * whose operands are given pre-split (A16 [+ A16lo] and B_split); for tests and tile sweeps.  tile: 0 = 128x128, 1 = 128x64,
 * 2 = 64x128, 3 = 64x64 (rows x channels per workgroup); stages: LDS ring depth 2..4 (3 for tile 0), or 13 = the halo-resident form for
 * stride-1 3x3 rows with pad == dil over a same-size map (forward or data-gradient rows; the pixel rows around a tile are staged once per
 * channel chunk, the nine taps read them at row offsets; <= 448 halo rows, <= 160 KB of LDS), or 200 + 10 * slices + ring depth (2 | 3) =
 * the reduction of every output tile cut into 2..8 slices (grid.y): each slice parks its partial tile in cdetr_gemm_desc.splitk_ws
 * (4096 arrival counters + tiles * slices * tile bytes), the slice that arrives last adds the partials in slice order and runs the
 * fused epilogue (bit-identical whichever slice that is; needs slices <= k-tiles).  CDETR_ERR_UNSUPPORTED when the
 * operand formats / alignment do not allow it (cdetr_last_error says why).                                                        */
int cdetr_gemm_dl(const cdetr_gemm_desc* d, int32_t tile, int32_t stages, void* stream);
/* n INDEPENDENT GEMMs submitted together (same results as n cdetr_gemm calls; no problem may read another's output).  Few-row
 * problems and the 64x128-tile class with a pre-split weight run as grouped launches (one kernel for up to 12 problems).
 * Replaces: sibling F.linear calls on independent inputs, e.g. the five in-projections of
 * A2/models/row_column_decoupled_attention.py:165-208 and the memory-side projections of all decoder layers.               */
int cdetr_gemm_group(const cdetr_gemm_desc* descs, int32_t n, void* stream);

/* dW[i][tap][c] += w_scale[i] * sum_p dY[p][i] * X[row(p,tap)][c]      (weight-gradient, split-K + fp32 atomics)
 * dbias[i]      += sum_p dY[p][i]                                       (optional, fused: NULL = skip)
 * Replaces: autograd of F.conv2d / F.linear w.r.t. weight and bias.  dW / dbias must hold the value to accumulate
 * onto (the caller zeroes the gradient arena once per step).                                                  */
typedef struct {
    int32_t P, Nout, Cin, taps;
    int32_t batch;
    int32_t precision; /* as in cdetr_gemm_desc */
    const float* dY; int64_t ldy, sY;
    const float* X; int64_t ldx, sX;
    float* dW; int64_t ldw, sW;
    const float* w_scale;
    float* dbias;
    cdetr_conv_geom g; /* mode DENSE or CONV_FWD (p = output pixel) */
    int32_t batch_inner; /* two-level batch, as in cdetr_gemm_desc */
    int32_t wg_target;   /* 0 = default.  > 0: the number of workgroups the launch (for cdetr_wgrad_group: the grouped launch this problem   */
                         /* becomes part of -- the largest request of its members) should spread its pixel slices over.  The default (768 = 3  */
                         /* per CU; 384 for a single weight of <= 16 tiles) suits a launch that has the chip to itself; a caller that runs the  */
                         /* launch BESIDE another stream's chain asks for 384 (fewer, longer workgroups disturb the chain less), the step's     */
                         /* last weight gradients for several thousand.  Results do not depend on it beyond fp32 summation order.               */
    int64_t sY2, sX2, sW2;
    const void* dY16;  /* optional bf16 TWINS of dY / X (same shapes, leading dimensions and batch strides, in elements): with        */
    const void* X16;   /* precision 3 (plain bf16) the kernel reads these instead -- half the operand bytes, no conversion at staging. */
                       /* Needs both, Nout / Cin / ldy / ldx multiples of 8, 16-byte aligned bases, no dbias; otherwise dY / X are read. */
                       /* dY and / or X may be NULL when the tensor exists as its twin only (an inner gradient written with cdetr_gemm_desc.C  */
                       /* == NULL; an activation kept as interleaved groups + twin): the twin-fed tile kernel must then be the one that runs   */
                       /* (more than 1024 pixels, the conditions above), CDETR_ERR_ARG otherwise -- never a silent read of a missing tensor.   */
                       /* CDETR_WGRAD_KP (environment, A/B): 32-pixel blocks staged per barrier by the twin-fed kernels, 1 (default) / 2 / 4.  */
} cdetr_wgrad_desc;
int cdetr_wgrad(const cdetr_wgrad_desc* d, void* stream);
/* n INDEPENDENT weight-gradient problems submitted together (same semantics as n cdetr_wgrad calls in any order; problems may
 * accumulate into the same dW / dbias).  Problems of the few-pixel and of the 64x64 transpose-read kernel class run as grouped
 * launches (one kernel for up to 16 problems), the rest one by one.  Replaces: the per-parameter autograd weight-gradient nodes of
 * one layer's backward (torch.autograd of F.linear at A2/models/transformer.py:242-279, 337-409), whose results only the
 * optimizer reads.                                                                                                       */
int cdetr_wgrad_group(const cdetr_wgrad_desc* descs, int32_t n, void* stream);

/* out[n] += sum_m X[m][n]   (bias gradients; atomics) */
int cdetr_colsum(const float* X, int64_t ldx, int32_t M, int32_t N, float* out, void* stream);

/* ---- fused optimizer tail over the flat arenas (A2/engine.py:54-57 clip_grad_norm_(0.1) + torch.optim.AdamW) ------
 * cdetr_sumsq:      out[0] = sum_i g[i]^2, bit-reproducible (fixed summation order: data-parallel ranks holding the same
 *                   reduced gradient must compute the same clip coefficient, or their parameters drift apart).
 *                   workspace: CDETR_SUMSQ_WS_FLOATS device floats of scratch (per-block partial sums; a second one-block
 *                   launch adds them in index order).
 * cdetr_adamw_step: g' = g * grad_div; coef = min(max_norm / (||g'|| + 1e-6), 1) (max_norm <= 0: no clip);
 *                   p *= 1 - lr*wd; m = b1 m + (1-b1) g'coef; v = b2 v + (1-b2)(g'coef)^2;
 *                   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),  lr = lr[i] * state[1], t = state[0] + 1.
 *                   state (device float[4]): [0] step count (incremented), [1] lr scale (StepLR), [2] <- ||g'|| (logging),
 *                   [3] += 1 when ||g'|| is NaN / Inf: that step changes neither p, m, v nor the step count (the reference
 *                   aborts on a non-finite loss BEFORE its optimizer step, A2/engine.py:44-49).                               */
#define CDETR_SUMSQ_MAX_BLOCKS 2048
#define CDETR_SUMSQ_WS_FLOATS CDETR_SUMSQ_MAX_BLOCKS
int cdetr_sumsq(const float* g, int64_t n, float* out, float* workspace, void* stream);
int cdetr_adamw_step(float* p, const float* g, float* m, float* v, const float* lr, int64_t n, const float* sumsq,
                     float* state, float max_norm, float beta1, float beta2, float eps, float weight_decay, float grad_div,
                     void* stream);
/* cdetr_adamw_step2: the same with the learning rates given as TWO values split at element lr_split when lr == NULL (elements
 * [0, lr_split) use lr0, the rest lr1; lr_split a multiple of 4): the reference's parameter groups (A2/main.py:157-183) are two
 * contiguous ranges of the arena, and the per-element table is a 150 MB stream the update does not need.                       */
int cdetr_adamw_step2(float* p, const float* g, float* m, float* v, const float* lr, float lr0, float lr1, int64_t lr_split, int64_t n,
                      const float* sumsq, float* state, float max_norm, float beta1, float beta2, float eps, float weight_decay,
                      float grad_div, void* stream);
/* dz[i] = y[i] > 0 ? dy[i] * scale : 0      (ReLU backward of the fused linear+ReLU layers) */
int cdetr_relu_mask(const float* y, const float* dy, float* dz, int64_t n, float scale, void* stream);
/* the same, also writing a bf16 twin of dz (dz16 may be NULL) */
int cdetr_relu_mask2(const float* y, const float* dy, float* dz, void* dz16, int64_t n, float scale, void* stream);

/* ---- transformer-layer glue (HBM-bound, fused so a layer touches its activations as few times as possible) --------
 * cdetr_layernorm_fwd/bwd: nn.LayerNorm over the last dim C (multiple of 256, <= 1024); fwd saves mean/rstd per row;
 *   bwd: dx (+ add), dgamma += ..., dbeta += ... (A2/models/transformer.py:233,274,334-339,420,425).
 * cdetr_posadd2:   Qr = X + Prow (broadcast over h), Qc = X + Pcol (broadcast over w)   (transformer.py:248-255).
 * cdetr_hw_reduce: Or[n,x,:] = sr * sum_y Xr[n,y,x,:] (+ Ar), Oc[n,y,:] = sc * sum_x Xc[n,y,x,:] (+ Ac)
 *                  (forward: the mean-before-project key inputs; backward: sums of gradient maps over the broadcast axis).
 * cdetr_bcast_add2: out = T + sr * Br[n,x,:] + sc * Bc[n,y,:]                                                   */
/* nn.GroupNorm(G, C) on the NHWC activation x [B][P][C] (P = h*w; C / G a multiple of 4, <= 64): fwd saves mean / rstd [B*G];
 * bwd writes dx and ACCUMULATES dgamma / dbeta [C].  Replaces: GroupNorm(32, 256) of the input projection,
 * A2/models/anchor_detr.py:86-92 (evaluated there on an NCHW tensor).                                                       */
int cdetr_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                        int32_t B, int32_t P, int32_t C, int32_t G, float eps, void* stream);
int cdetr_groupnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx,
                        float* dgamma, float* dbeta, int32_t B, int32_t P, int32_t C, int32_t G, void* stream);
/* The same two calls with the pixels of an image spread over workgroups (C == 256: coalesced 1 KB rows, B * P / 32 workgroups instead of B * G) and the
 * group statistics crossing them through `ws` (>= B * ceil(P / 32) * G * 12 bytes forward, * 8 backward; contents undefined before and after): two
 * launches each -- per-chunk partials (forward: count / mean / centred second moment, merged with Chan's update in chunk order), then merge + apply.
 * Shapes the split form does not cover (C != 256, fewer than 128 pixels, ws too small / NULL) run cdetr_groupnorm_fwd / _bwd.                       */
int cdetr_groupnorm_fwd_ws(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                           int32_t B, int32_t P, int32_t C, int32_t G, float eps, void* ws, int64_t ws_bytes, void* stream);
int cdetr_groupnorm_bwd_ws(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx,
                           float* dgamma, float* dbeta, int32_t B, int32_t P, int32_t C, int32_t G, void* ws, int64_t ws_bytes, void* stream);

int cdetr_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                        int32_t rows, int32_t C, float eps, void* stream);
/* the same forward, plus o1 = y + a1 and (optional pair) o2 = y + a2 in the same pass: the positional adds that consume a decoder
 * LayerNorm's output (tgt + query_pos_x / tgt + query_pos_y, A2/models/transformer.py:385-389; next layer's tgt + query_pos, :369) */
int cdetr_layernorm_fwd_add(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                            const float* a1, const float* a2, float* o1, float* o2, int32_t rows, int32_t C, float eps, void* stream);
int cdetr_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* add, float* dx, float* dgamma, float* dbeta, int32_t rows, int32_t C, void* dx16, void* stream);
/* (dx16: optional bf16 twin of dx, C = 256 -- the A16 operand of the plain-bf16 data-gradient GEMM that consumes dx)
 * the same with the incoming gradient given as a sum, dy + g1 + g2 (g2 optional), and the side effects acc1 += g1, acc2 += g2 (optional, in
 * place): cdetr_grad_merge folded into the LayerNorm backward that consumes its result (decoder backward: the gradient of a sub-layer input is
 * the residual branch plus one or two projection branches whose values also accumulate into the stack-wide query-position gradients).  Br / Bc
 * (optional pair, rows = N*H*W): two more addends broadcast over the map, + sr * Br[n,x,:] + sc * Bc[n,y,:] -- cdetr_bcast_add2_sum folded in the
 * same way (encoder backward: the gradient of a layer's input feeds the LayerNorm backward of the layer below).  C = 256.                       */
int cdetr_layernorm_bwd_merge(const float* dy, const float* g1, const float* g2, float* acc1, float* acc2, const float* Br, const float* Bc,
                              float sr, float sc, int32_t H, int32_t W, const float* x, const float* mean, const float* rstd,
                              const float* gamma, const float* add, float* dx, float* dgamma, float* dbeta, int32_t rows, int32_t C,
                              void* dx16, void* stream);
int cdetr_posadd2(const float* X, const float* Prow, const float* Pcol, float* Qr, float* Qc, int32_t N, int32_t H, int32_t W,
                  int32_t C, void* stream);
int cdetr_hw_reduce(const float* Xr, const float* Xc, const float* Ar, const float* Ac, float* Or, float* Oc, int32_t N,
                    int32_t H, int32_t W, int32_t C, float scale_r, float scale_c, void* stream);
/* cdetr_posadd2 and cdetr_hw_reduce(X, X, Prow, Pcol, ...) of one encoder layer as ONE launch (both read the same source and nothing of each
 * other: A2/models/transformer.py:246-252): Qr / Qc = X + Prow / Pcol (broadcast), Kr / Kc = mean over H / W of X, scaled, + Prow / Pcol. */
int cdetr_posadd2_hw_reduce(const float* X, const float* Prow, const float* Pcol, float* Qr, float* Qc, float* Kr, float* Kc, int32_t N,
                            int32_t H, int32_t W, int32_t C, float scale_r, float scale_c, void* stream);
int cdetr_bcast_add2(const float* T, const float* Br, const float* Bc, float* out, int32_t N, int32_t H, int32_t W, int32_t C,
                     float sr, float sc, void* stream);
/* the same with up to two further addends of T's shape (T2, T3; NULL = absent): out = T + T2 + T3 + sr*Br + sc*Bc -- sibling data gradients
 * of one input (the three projections of A2/models/row_column_decoupled_attention.py:165-208 that read src) run as ONE grouped launch into
 * separate buffers and are summed here instead of being chained through residual epilogues.                                           */
int cdetr_bcast_add2_sum(const float* T, const float* T2, const float* T3, const float* Br, const float* Bc, float* out, int32_t N, int32_t H,
                         int32_t W, int32_t C, float sr, float sc, void* stream);
/* decoder-layer glue (A2/models/transformer.py:366-403): cdetr_add2: O1 = T + A, O2 = T + B (B and O2 NULL together);
 * cdetr_grad_merge (backward of those sites): out = base + g1 (+ g2), acc1 += g1, acc2 += g2 (g2 / acc1 / acc2 may be NULL);
 * n = element count, a multiple of 4, all pointers 16-byte aligned.                                                    */
int cdetr_add2(const float* T, const float* A, const float* B, float* O1, float* O2, int64_t n, void* stream);
/* sine positional embedding (A2/models/transformer.py:474-494): out[r][i] = i even ? sin(x) : cos(x),
 * x = pos[r * pstride] * 2 pi / temperature^(2 floor(i/2) / nfeat), rows of `out` / `dout` ldo floats apart;
 * backward: dpos[r * dstride] (+)= sum_i dout[r][i] * d out[r][i] / d pos (accumulate != 0 adds).                       */
int cdetr_sine_embed(const float* pos, int32_t pstride, float* out, int64_t ldo, int32_t rows, int32_t nfeat, float temperature,
                     void* stream);
int cdetr_sine_embed_bwd(const float* pos, int32_t pstride, const float* dout, int64_t ldo, float* dpos, int32_t dstride, int32_t rows,
                         int32_t nfeat, float temperature, int32_t accumulate, void* stream);
int cdetr_grad_merge(const float* base, const float* g1, const float* g2, float* acc1, float* acc2, float* out, int64_t n,
                     void* stream);

/* ---- k-contiguous mirrors of the weights used as data-gradient operands ------------------------------------------------
 * The data-gradient GEMMs of A2/models/resnet.py:140-160 (conv backward) and of every F.linear site contract over the
 * OUTPUT channels of a weight W[o][tap][c]; the matrix pipe wants that axis contiguous.  One launch rewrites every
 * registered weight as Wt[c][tap][o] = W[o][tap][c] * scale[o] (scale = folded FrozenBN, may be NULL) and / or as the
 * pre-split bf16 operand of the bf16x3 GEMM kernels; items live in device memory, `tile0` = prefix sum of
 * ceil(R/32)*ceil(C/32)*taps.                                                                                          */
typedef struct {
    const float* src;     /* W  [R][taps][C]                                                                    */
    float* dst;           /* transpose = 1: Wt [C][taps][R] fp32 (may be NULL)                                  */
    void* dst_split;      /* pre-split bf16 image (may be NULL): transpose = 1: of Wt (needs R % 32 == 0),      */
                          /* transpose = 0: of W itself (needs C % 32 == 0); layout as cdetr_gemm_desc.B_split  */
    const float* scale;   /* [R] or NULL                                                                        */
    int32_t R, C, taps, tile0, transpose, pad_;
    void* dst_hi;         /* optional, transpose = 1: bf16(Wt) [C][taps][R], the plain-bf16 data-gradient operand        */
                          /* (cdetr_gemm_desc.B16: full cache lines of hi values, no interleaved lo halves)             */
} cdetr_mirror_item;
int cdetr_weight_mirror(const cdetr_mirror_item* items_dev, int32_t n_items, int32_t total_tiles, void* stream);
/* the forward images only (every item transpose = 0, 16-byte aligned src / dst_split, C % 32 == 0): a streaming pass, 4 weights per thread;
 * item.tile0 = first block of the item in units of 4096 weights (ceil(R * taps * C / 4096) blocks per item), total_blocks their sum.           */
int cdetr_weight_images(const cdetr_mirror_item* items_dev, int32_t n_items, int32_t total_blocks, void* stream);

/* 3x3 stride-2 pad-1 max pooling, NHWC (A2/models/resnet.py:206,265) */
int cdetr_maxpool3x3s2(const float* X, float* Y, int32_t Nimg, int32_t H, int32_t W, int32_t C, void* stream);
/* the same, also writing the split-bf16 planes of Y (Y16 = bf16(Y), Y16lo = bf16(Y - float(Y16)); either may be NULL) */
int cdetr_maxpool3x3s2_split(const float* X, float* Y, void* Y16, void* Y16lo, int32_t Nimg, int32_t H, int32_t W, int32_t C, void* stream);

/* ---- Row-Column Decoupled Attention core (A2/models/row_column_decoupled_attention.py:215-309) -------------
 * Inputs are the PROJECTED tensors: q_row,q_col [N][L][E]; k_row [N][W][E] (already averaged over H);
 * k_col [N][H][E] (averaged over W); v [N][H][W][E]; E = nh*32.  mask_row [N][W], mask_col [N][H] uint8 (1 = pad)
 * or NULL.  Computes, per (n, head):  A_row = softmax_W(scale*q_row.k_row^T), A_col = softmax_H(scale*q_col.k_col^T),
 * out[n][q][head*32+c] = sum_h sum_w A_col[q][h] A_row[q][w] v[n][h][w][head*32+c].
 * a_row [N][nh][L][Wp], a_col [N][nh][L][Hp] (Wp = W rounded up to 4, Hp = H rounded up to 8; pad = 0) are written
 * for the backward pass; both NULL = inference: nothing is saved (2 x 8 MB of stores per encoder call at 800 x 800).  Feature maps up to
 * 128 x 1024 keys; the MFMA two-step kernels cover W <= 96 (an 800 x 1333 FSCD-LVIS image at stride 16), wider maps the generic kernel.  */
typedef struct {
    int32_t N, L, H, W, nh; /* head dim fixed at 32 */
    int32_t precision;      /* 0 = fp32 MFMA, 1 = split-bf16 x3 (as in cdetr_gemm_desc) */
    float scale;
    const float* q_row; const float* q_col; const float* k_row; const float* k_col; const float* v;
    const uint8_t* mask_row; const uint8_t* mask_col;
    float* out; float* a_row; float* a_col;
    void* ws;               /* optional scratch (zero-initialised once, >= 16 KB + partial outputs; the kernels leave its counters zero): lets the */
    int64_t ws_bytes;       /* two-step forward cut the key rows into slices when (N * nh * L / 128) workgroups would not fill the chip; the     */
                            /* split-reduction scratch of cdetr_gemm_desc.splitk_ws may be passed (launches of one stream do not overlap)      */
} cdetr_rcda_fwd_desc;
int cdetr_rcda_fwd(const cdetr_rcda_fwd_desc* d, void* stream);

/* Backward of the core: given d_out [N][L][E] and the saved a_row/a_col, writes
 * ds_row [N][nh][L][Wp], ds_col [N][nh][L][Hp] (gradients of the PRE-softmax logits, already multiplied by
 * `scale`) and accumulates d_v [N][H][W][E] (must be zeroed by the caller).
 * Optional (NULL = skip): with k_row [N][W][E], k_col [N][H][E] and dq_row / dq_col [N][L][E] given, the same launch also
 * writes the query gradients dq_row[q] = sum_w ds_row[q][w] k_row[w], dq_col[q] = sum_h ds_col[q][h] k_col[h] (per head) --
 * the logits -> projected-query step of the reference's autograd (A2/models/row_column_decoupled_attention.py:230-262).   */
typedef struct {
    int32_t N, L, H, W, nh;
    int32_t precision;
    float scale;
    const float* d_out; const float* a_row; const float* a_col; const float* v;
    float* ds_row; float* ds_col; float* d_v;
    const float* k_row; const float* k_col;
    float* dq_row; float* dq_col;
    const float* q_row; const float* q_col;   /* optional, with dq_*: also ACCUMULATE the key gradients; ds_row / ds_col may then  */
                                              /* both be NULL (the logit gradients never leave the chip)                        */
    float* dk_row; float* dk_col;             /* dk_row[n][w] += sum_q ds_row[q][w] q_row[q] (per head; caller zeroes them)    */
} cdetr_rcda_bwd_desc;
int cdetr_rcda_bwd(const cdetr_rcda_bwd_desc* d, void* stream);
static inline int32_t cdetr_rcda_wp(int32_t W) { return (W + 3) & ~3; }
static inline int32_t cdetr_rcda_hp(int32_t H) { return (H + 7) & ~7; }

/* ---- decoder self-attention core (nn.MultiheadAttention(256, 8) at A2/models/transformer.py:337,369-370) ---------
 * qk [N][L][2E] = projected queries | keys, v [N][L][E], E = nh*32; o = softmax(scale * q k^T) v per head, lse [N][nh][L]
 * saved for backward.  cdetr_mha_bwd writes d_qk [N][L][2E], d_v [N][L][E]; work: N*nh*L floats of scratch (the row sums D of the
 * fp32 mode's two launches; the split-bf16 mode is ONE launch whose key half forms D itself and leaves `work` untouched).
 * cdetr_mha_bwd precision: 0 = fp32, 1 = split-bf16 x3 throughout, 3 = the scores recomputed in split-bf16 x3 (p = exp(s - lse) against the
 * forward's lse) and the four gradient contractions (dP, dQ, dK, dV) in plain bf16, one MFMA per product.                           */
int cdetr_mha_fwd(const float* qk, const float* v, float* o, float* lse, int32_t N, int32_t L, int32_t nh, float scale,
                  int32_t precision, void* stream);
int cdetr_mha_bwd(const float* qk, const float* v, const float* o, const float* d_o, const float* lse, float* d_qk, float* d_v,
                  float* work, int32_t N, int32_t L, int32_t nh, float scale, int32_t precision, void* stream);

/* ---- Hungarian matcher (A2/models/matcher.py:197-247 + scipy.optimize.linear_sum_assignment) -------------
 * cdetr_match_cost: per image b, cost[b] = 5*L1 + 2*focal-class + 2*(-GIoU) in fp32 with the reference's expression
 * order, written in SOLVER layout: [nr][nc] with nr = min(Q,T_b), nc = max(Q,T_b) (transposed when T_b < Q,
 * exactly like scipy transposes a tall matrix).  cost_off[b] = float offset of image b's block.
 * logits [B][Q][ncls] (column 0 used: all labels are 0), boxes [B][Q][4] cxcywh, tgt [sum T][4], tgt_off[B+1].    */
int cdetr_match_cost(const float* logits, int32_t ncls, const float* boxes, const float* tgt, const int32_t* tgt_off,
                     const int64_t* cost_off, int32_t B, int32_t Q, float w_class, float w_bbox, float w_giou,
                     float* cost, void* stream);
/* cdetr_lsap: exact rectangular assignment on the device (float64 duals, shortest augmenting paths, the tie rule of
 * scipy's solver) -- one workgroup per image.  Writes idx_i/idx_j (int64, [B][Mmax]) = (query, target) pairs with
 * idx_i ascending -- the (row_ind, col_ind) scipy returns; status[b] = 0 ok / 1 infeasible / 2 invalid cost.
 * nc_max = max(Q, max_b T_b) (host-known; sizes the LDS-resident solver state, limit 3800);
 * Mmax = row stride of idx_i / idx_j (>= max_b min(Q, T_b)).                                                    */
int cdetr_lsap(const float* cost, const int64_t* cost_off, const int32_t* tgt_off, int32_t B, int32_t Q,
               int32_t nc_max, int32_t Mmax, int64_t* idx_i, int64_t* idx_j, int32_t* status, void* stream);

/* ---- SetCriterion (A2/models/anchor_detr.py:143-367, losses [labels, boxes, cardinality, vars], no aux) ----------------
 * cdetr_criterion_fwd: from the raw predictions, the concatenated targets and the matcher's device indices
 * (idx_i / idx_j [B][Mmax], first min(Q, T_b) entries of row b valid) compute
 *   losses[6] = { loss_ce, class_error, cardinality_error, loss_bbox, loss_giou, loss_variance }
 * and the gradient of every differentiable loss w.r.t. its inputs:
 *   g_logits [B,Q,C] = d loss_ce / d logits;  g_l1 / g_giou / g_var_box [B,Q,4] = d {loss_bbox, loss_giou, loss_variance} / d boxes;
 *   g_vars [B,Q,2] = d loss_variance / d vars.   num_boxes is a DEVICE scalar (the data-parallel normaliser).
 * cdetr_criterion_bwd: d_logits = g[0] g_logits; d_boxes = g[3] g_l1 + g[4] g_giou + g[5] g_var_box; d_vars = g[5] g_vars
 * (g = upstream gradient of the six scalars, device; see the prototype for how it is formed).                              */
typedef struct {
    int32_t B, Q, C, num_classes, Mmax;
    float alpha;
    const float* logits;        /* [B,Q,C] */
    const float* boxes;         /* [B,Q,4] cxcywh */
    const float* vars;          /* [B,Q,2] */
    const float* tgt_boxes;     /* [sum T,4] */
    const int64_t* tgt_labels;  /* [sum T] */
    const int32_t* tgt_off;     /* [B+1] */
    const int64_t* idx_i;       /* [B,Mmax] */
    const int64_t* idx_j;       /* [B,Mmax] */
    const float* num_boxes;     /* device scalar */
    float* losses;              /* [6] */
    float* g_logits;
    float* g_l1;
    float* g_giou;
    float* g_var_box;
    float* g_vars;
    const float* loss_weights;  /* optional [6] (NULL = absent): losses[6] <- sum_k loss_weights[k] * losses[k], the weighted total of
                                   A2/engine.py:37 (losses then has 7 elements) */
} cdetr_criterion_desc;
int cdetr_criterion_fwd(const cdetr_criterion_desc* d, void* stream);
/* g6 (upstream gradient of the six scalars) and g_total (upstream gradient of the weighted total, with loss_weights) may each be
 * NULL = zero: the effective gradient of loss k is g6[k] + g_total[0] * loss_weights[k]. */
int cdetr_criterion_bwd(const float* g6, const float* g_total, const float* loss_weights, const float* g_logits, const float* g_l1,
                        const float* g_giou, const float* g_var_box, const float* g_vars, float* d_logits, float* d_boxes, float* d_vars,
                        int32_t BQ, int32_t C, void* stream);

const char* cdetr_last_error(void);
int cdetr_abi_version(void);
/* Stream plumbing of the trainer (no reference counterpart: the reference runs one stream and drains it every step, A2/engine.py:33-57).
 * cdetr_delay: one idle wavefront for `us` microseconds -- holds `stream` back without occupying the chip (stream-concurrency probe).
 * cdetr_flag_signal / cdetr_flag_wait: ordering between two streams from INSIDE captured graphs, where no event can be recorded: the signal
 *   (one thread) adds 1 to *flag; the wait (one thread, at the head of the other stream's work) sleeps until *flag has passed *seen -- its own
 *   count of consumed signals, updated by the kernel -- or `timeout_us` has gone by (a missing signal costs a delay, never a hang).
 *   Every signal needs exactly one wait: a wait advances *seen by one (to the flag's value when it was satisfied, by one when it timed out
 *   and the signal arrives later), so a signal nobody waited for leaves the flag one ahead and every later wait passes at once.  A replay
 *   whose signal has no waiter CONSUMES it with timeout_us = 0 (advances *seen by one, whichever side of the signal it lands on).
 *   The flags are a SCHEDULING hint only: whatever must hold for correctness is ordered by events / stream order as well.
 *   engine.Trainer releases the next batch's frozen stage (A2/models/backbone.py:93-95) the moment the Hungarian solve of the current step
 *   (A2/models/matcher.py:243-247) is about to start: the solve needs one whole compute unit's LDS for its cost matrix.                   */
int cdetr_delay(int32_t us, void* stream);
int cdetr_flag_signal(int32_t* flag, void* stream);
int cdetr_flag_wait(const int32_t* flag, int32_t* seen, int32_t timeout_us, int32_t post_us, void* stream);   /* post_us: idle on after the signal */

/* ---- glue (csrc/glue.hip): the small steps between the GEMMs, one launch each -------------------------------------------------
 * cdetr_mask_prep: padding mask [B][H][W] (bytes, non-zero = padding) -> m [B][h][w] (nearest-neighbour down-sampling,
 *   A2/models/backbone.py:143), mask_row [B][w] = m[:, 0, :], mask_col [B][h] = m[:, :, 0] (the RCDA key masks,
 *   row_column_decoupled_attention.py:238-249), pos_row [B][w] / pos_col [B][h] = mask2pos (A2/models/transformer.py:497-503),
 *   extent [B][2] = un-padded (rows, columns) of each image in feature cells.
 * cdetr_stem_pack: images NCHW [B][3][H][W] -> xp [B][Ha][Wa][4] NHWC, image at (pad_y, pad_x), zeros elsewhere, 4th channel 0
 *   (input of the row-packed 7x7 stem, A2/models/resnet.py:178-180).
 * cdetr_exemplar_fwd: pf[b][c] = mean over the present exemplars k of x[b][cell(b,k)][c] with x [B][h*w][C] NHWC, rects [B][K][4]
 *   normalised xyxy (x2 < 0: absent), cell = centre of the box in feature cells, truncated (A2/models/backbone.py:122-131);
 *   per_image: image b uses rects[b] and extent[b]; else rects[0] scaled by (h, w) for every image (the reference's rule).
 *   Writes idx [B][K] (cell or -1) and inv_cnt [B] for the backward.   cdetr_exemplar_bwd: dx[b][cell][c] += dpf[b][c] * inv_cnt[b].
 * cdetr_aggr_weight_fwd: Weff[b] = W[:, :C] + W[:, C:] * pf[b] ([B][d][C]) and its transpose WeffT [B][C][d]; W is [d][2C]
 *   (A2/models/backbone.py:132-136 + anchor_detr.py:119 with the concatenation folded into the weight).
 * cdetr_aggr_weight_bwd: from dWeff [B][d][C]: gW[:, :C] += sum_b dWeff[b]; gW[:, C:] += sum_b dWeff[b] * pf[b] (gW may be NULL);
 *   dpf[b][c] += sum_o dWeff[b][o][c] * W[o][C + c]  (dpf must be zeroed by the caller).  B <= 16.
 * cdetr_box_head_fwd: boxes[m] = sigmoid(tmp[m] + [inverse_sigmoid(ref[m % R]), 0, 0]); tmp / boxes [M][4], ref [R][2], M % R == 0
 *   (A2/models/transformer.py:193-203, A2/util/misc.py:475-479).  cdetr_box_head_bwd: d_tmp = d_boxes * s (1 - s) and, when d_ref is
 *   not NULL, d_ref[m % R] (+)= d_tmp[:2] * d inverse_sigmoid (torch's clamp-backward rule; plain store when M == R, else atomic
 *   accumulation into a caller-zeroed buffer).                                                                                  */
int cdetr_mask_prep(const uint8_t* mask, int32_t B, int32_t H, int32_t W, int32_t h, int32_t w, uint8_t* m, uint8_t* mask_row,
                    uint8_t* mask_col, float* pos_row, float* pos_col, float* extent, void* stream);
int cdetr_stem_pack(const float* images, float* xp, int32_t B, int32_t H, int32_t W, int32_t Ha, int32_t Wa, int32_t pad_y, int32_t pad_x,
                    void* stream);
int cdetr_exemplar_fwd(const float* x, const float* rects, const float* extent, int32_t per_image, int32_t B, int32_t h, int32_t w,
                       int32_t C, int32_t K, int32_t* idx, float* inv_cnt, float* pf, void* stream);
int cdetr_exemplar_bwd(const float* dpf, const int32_t* idx, const float* inv_cnt, float* dx, int32_t B, int32_t P, int32_t C, int32_t K,
                       void* stream);
int cdetr_aggr_weight_fwd(const float* W, const float* pf, float* Weff, float* WeffT, int32_t B, int32_t d, int32_t C, void* stream);
int cdetr_aggr_weight_bwd(const float* dWeff, const float* pf, const float* W, float* gW, float* dpf, int32_t B, int32_t d, int32_t C,
                          void* stream);
int cdetr_box_head_fwd(const float* tmp, const float* ref, float* boxes, int32_t M, int32_t R, void* stream);
int cdetr_box_head_bwd(const float* d_boxes, const float* boxes, const float* ref, float* d_tmp, float* d_ref, int32_t M, int32_t R,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif
