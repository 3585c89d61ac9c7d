// wgrad_dl.hip -- weight gradients dW[i][tap][c] += sum_p dY[p][i] * X[row(p, tap)][c] from the bf16 TWINS of dY and X
// (cdetr_wgrad_desc.dY16 / X16) with DIRECT-TO-LDS operand loads.  Math: autograd of F.conv2d / F.linear w.r.t. the weight
// (A2/models/resnet.py:140-160 backward, transformer.py:242-279 backward); same contract as wgrad_tr16_kernel (igemm.hip).
//
// The reduction runs over pixels while both operands are channel-contiguous, so an operand tile in LDS is the natural
// [64-channel group][pixel][64 channels] image (128-byte rows = one cache line of a pixel row) and an MFMA fragment is a transposed
// view of it, read with gfx950's ds_read_b64_tr_b16.  The image is filled by `global_load_lds_dwordx4`: a lane's 16 bytes are 8
// channels of one pixel, 8 lanes fetch one full 128-byte line, a wave instruction fills 8 pixel rows, no VGPR round trip, no ds_write.
// Bank swizzle on the source side (the DMA writes lane-linearly): the two 64-byte halves of a row are swapped on pixel rows with
// bit 1 set, so the 4 rows x 64 bytes a half-wave's transpose read touches cover all 64 banks once.  A ring of STAGES pixel tiles keeps STAGES - 1 tiles of loads in flight across the barriers (counted vmcnt), so latency
// is hidden by the ring instead of by occupancy -- which is what lets the OUTPUT tile grow to 128 x 128: the register-staged kernel
// at 64 x 64 re-reads every pixel row of dY Cin / 64 times and of X Cout / 64 times from L2 (7.7 GB per step at two 800x800 images,
// ~8.4 TB/s: the L2 -> CU wall), a 128 x 128 tile halves that.
#include "../../include/cdetr_hip.h"
#include "common.h"
#include "rows.h"
#include "dl_common.h"
#include <stdlib.h>
#include <algorithm>
#include <vector>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 lds_tr8(const unsigned char* p0) {     // 8 consecutive pixels (k) of this lane's channel: two transpose reads
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 512));     // 4 pixel rows of 128 B further
    const s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, c);
}

template <int FI, int FJ, int KP, int STAGES>
struct WdCfg {
    static constexpr int BI = 64 * FI, BJ = 64 * FJ;            // output tile: BI output channels x BJ input channels (of one tap)
    static constexpr int GRPB = KP * 128;                        // bytes of one 64-channel group of one pixel tile
    static constexpr int A_STAGE = FI * GRPB, B_STAGE = FJ * GRPB;
    static constexpr int NPIX = KP / 32;                         // pixel rows per thread and tile (256 threads = 32 rows x 8 chunks per pass)
    static constexpr int NA = FI * NPIX, NB = FJ * NPIX;         // 16-byte pieces per thread per pixel tile
    static constexpr int LDS = STAGES * (A_STAGE + B_STAGE);
    static_assert(KP == 32 || KP == 64, "pixels per tile");
};

template <int FI, int FJ, int KP, int STAGES>
__device__ __forceinline__ void wgrad_dl_body(const cdetr_wgrad_desc& d, const int tilesI, const int tilesJ, const int kt_per_slice,
                                              const int bx, const int by, const bool single, unsigned char* smem) {
    using Cf = WdCfg<FI, FJ, KP, STAGES>;
    constexpr int BI = Cf::BI, BJ = Cf::BJ, NA = Cf::NA, NB = Cf::NB, NPIX = Cf::NPIX, GRPB = Cf::GRPB, A_STAGE = Cf::A_STAGE, B_STAGE = Cf::B_STAGE;
    constexpr int NI = NA + NB;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int i32 = lane & 31, g = lane >> 5;
    const int ti = bx % tilesI;
    const int tj = bx / tilesI;
    const int tap = tj / tilesJ;
    const int c0 = (tj - tap * tilesJ) * BJ;
    const int i0 = ti * BI;
    const __bf16* __restrict__ dY = reinterpret_cast<const __bf16*>(d.dY16);
    const __bf16* __restrict__ X = reinterpret_cast<const __bf16*>(d.X16);
    const int nkt_all = (d.P + KP - 1) / KP;
    const int kt_begin = by * kt_per_slice;
    const int nkt = min(nkt_all, kt_begin + kt_per_slice) - kt_begin;
    if (nkt <= 0) return;

    // ---------------------------------------------------------------- loader: NPIX pixel rows per thread and tile, one 16-byte chunk of
    // each 64-channel group of them (8 consecutive lanes = one 128-byte line)
    const int prow = tid >> 3;                                  // pixel row inside a 32-row pass
    const int chunk = (tid & 7) ^ (((tid >> 4) & 1) << 2);      // logical chunk behind LDS slot (tid & 7): halves swapped on rows with bit 1 set
    const bool dense = d.g.mode == CDETR_ROWS_DENSE;
    const int ky = dense ? 0 : tap / d.g.kw, kx = dense ? 0 : tap - (tap / d.g.kw) * d.g.kw;
    int pp[NPIX], pn[NPIX], py[NPIX], px[NPIX];
#pragma unroll
    for (int h = 0; h < NPIX; ++h) {
        pp[h] = kt_begin * KP + h * 32 + prow;
        pn[h] = py[h] = px[h] = 0;
        if (!dense) {
            const int hw = d.g.Hc * d.g.Wc;
            pn[h] = pp[h] / hw;
            const int rem = pp[h] - pn[h] * hw;
            py[h] = rem / d.g.Wc;
            px[h] = rem - py[h] * d.g.Wc;
        }
    }
    int aoff[FI], boff[FJ];                                     // byte offset of this thread's 8 channels inside a pixel row, per group
#pragma unroll
    for (int q = 0; q < FI; ++q) aoff[q] = min(i0 + q * 64 + chunk * 8, d.Nout - 8) * 2;
#pragma unroll
    for (int q = 0; q < FJ; ++q) boff[q] = min(c0 + q * 64 + chunk * 8, d.Cin - 8) * 2;
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(dl_zero_page) + (tid & 7) * 16;
    auto issue_next = [&](int stage) __attribute__((always_inline)) {
        unsigned char* la = smem + stage * A_STAGE + w * 1024;
        unsigned char* lb = smem + STAGES * A_STAGE + stage * B_STAGE + w * 1024;
#pragma unroll
        for (int h = 0; h < NPIX; ++h) {
            const int p = pp[h];
            const bool pv = p < d.P;
            long row = -1;
            if (pv) {
                if (dense) row = p;
                else {
                    const int iy = py[h] * d.g.stride - d.g.pad + ky * d.g.dil;
                    const int ix = px[h] * d.g.stride - d.g.pad + kx * d.g.dil;
                    if (iy >= 0 && iy < d.g.Ha && ix >= 0 && ix < d.g.Wa) row = ((long)pn[h] * d.g.Ha + iy) * d.g.Wa + ix;
                }
            }
            const unsigned char* yrow = reinterpret_cast<const unsigned char*>(dY + (long)(pv ? p : 0) * d.ldy);
            const unsigned char* xrow = reinterpret_cast<const unsigned char*>(X + (row >= 0 ? row : 0) * d.ldx);
#pragma unroll
            for (int q = 0; q < FI; ++q)        // LDS: group q, pixel rows h * 32 .. + 31 = pass q * NPIX + h of 4 KB
                __builtin_amdgcn_global_load_lds((gbl_vp)(pv ? yrow + aoff[q] : zero), (lds_vp)(la + (q * NPIX + h) * 4096), 16, 0, 0);
#pragma unroll
            for (int q = 0; q < FJ; ++q)
                __builtin_amdgcn_global_load_lds((gbl_vp)(row >= 0 ? xrow + boff[q] : zero), (lds_vp)(lb + (q * NPIX + h) * 4096), 16, 0, 0);
            pp[h] += KP;
            if (!dense) {
                px[h] += KP;
                while (px[h] >= d.g.Wc) { px[h] -= d.g.Wc; ++py[h]; }
                while (py[h] >= d.g.Hc) { py[h] -= d.g.Hc; ++pn[h]; }
            }
        }
    };

    // ---------------------------------------------------------------- fragments: transposed reads of the [pixel][64 channels] images
    // lane -> pixel g * 8 + (l16 >> 2) (+ 4 for the second read), channel ((lane >> 4) & 1) * 16 + (l16 & 3) * 4 of its 32-channel fragment
    const int l16 = lane & 15;
    const int swz = ((l16 >> 3) & 1) << 2;                      // the row's half swap: pixel bit 1 = bit 3 of l16
    const int lpart = (g * 8 + (l16 >> 2)) * 128 + (((lane >> 4) & 1) * 2 + ((l16 & 3) >> 1)) * 16 + (l16 & 1) * 8;
    int aofs[FI], bofs[FJ];
#pragma unroll
    for (int a = 0; a < FI; ++a) {
        const int ch = wm * 32 * FI + a * 32;                   // first channel of the fragment inside the tile
        aofs[a] = (ch >> 6) * GRPB + ((((ch & 63) >> 3) ^ swz) * 16) + lpart;
    }
#pragma unroll
    for (int b = 0; b < FJ; ++b) {
        const int ch = wn * 32 * FJ + b * 32;
        bofs[b] = STAGES * A_STAGE + (ch >> 6) * GRPB + ((((ch & 63) >> 3) ^ swz) * 16) + lpart;
    }
    f32x16 acc[FI][FJ];
#pragma unroll
    for (int a = 0; a < FI; ++a)
#pragma unroll
        for (int b = 0; b < FJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    auto compute = [&](int stage) __attribute__((always_inline)) {
        const unsigned char* as = smem + stage * A_STAGE;
        const unsigned char* bs = smem + stage * B_STAGE;
#pragma unroll
        for (int hp = 0; hp < KP / 16; ++hp) {
            bf16x8 ah[FI], bh[FJ];
#pragma unroll
            for (int a = 0; a < FI; ++a) ah[a] = lds_tr8(as + aofs[a] + hp * 2048);
#pragma unroll
            for (int b = 0; b < FJ; ++b) bh[b] = lds_tr8(bs + bofs[b] + hp * 2048);
#pragma unroll
            for (int a = 0; a < FI; ++a)
#pragma unroll
                for (int b = 0; b < FJ; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
        }
    };

    // ---------------------------------------------------------------- the ring (as igemm_dl_kernel)
    int f_idx = 0;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nkt) { issue_next(s); ++f_idx; }
    int st = 0, sn = STAGES - 1;
    for (int kt = 0; kt < nkt; ++kt) {
        const int fly = min(STAGES - 2, nkt - 1 - kt);
        if (fly >= 2) wait_vm(2 * NI);
        else if (fly == 1) wait_vm(NI);
        else wait_vm(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (f_idx < nkt) { issue_next(sn); ++f_idx; }
        compute(st);
        __builtin_amdgcn_sched_barrier(0);
        st = (st + 1 == STAGES) ? 0 : st + 1;
        sn = (sn + 1 == STAGES) ? 0 : sn + 1;
    }
    mfma_drain(acc);

    // ---------------------------------------------------------------- accumulate into the gradient: lane = input channel (contiguous)
    float* __restrict__ dW = d.dW;
#pragma unroll
    for (int a = 0; a < FI; ++a) {
#pragma unroll
        for (int b = 0; b < FJ; ++b) {
            const int c = c0 + (wn * FJ + b) * 32 + i32;
            if (c >= d.Cin) continue;
            float ws[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) ws[r] = 1.f;
            if (d.w_scale) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ws[r] = d.w_scale[min(i0 + (wm * FI + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, d.Nout - 1)];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + (wm * FI + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (i >= d.Nout) continue;
                const float v = acc[a][b][r] * ws[r];
                float* dst = dW + (long)i * d.ldw + (long)tap * d.Cin + c;
                if (single) *dst += v;
                else atomicAdd(dst, v);
            }
        }
    }
}

template <int FI, int FJ, int KP, int STAGES>
__global__ __launch_bounds__(256) void wgrad_dl_kernel(const cdetr_wgrad_desc d, const int tilesI, const int tilesJ, const int kt_per_slice) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    wgrad_dl_body<FI, FJ, KP, STAGES>(d, tilesI, tilesJ, kt_per_slice, blockIdx.x, blockIdx.y, gridDim.y == 1, smem);
}

// grouped launch: up to WD_MAX independent problems of one tile class, workgroups concatenated along grid.x (as WgradGroupArgs, igemm.hip)
constexpr int WD_MAX = 16;
struct WdItem { cdetr_wgrad_desc d; int tilesI, tilesJ, per, nx, ny, pad_; };
struct WdArgs { int n; int blk0[WD_MAX + 1]; WdItem it[WD_MAX]; };
static_assert(sizeof(WdArgs) <= 4000, "grouped launch arguments must fit the kernel-argument segment");

template <int FI, int FJ, int KP, int STAGES>
__global__ __launch_bounds__(256) void wgrad_dl_group_kernel(const WdArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int q = 0;
    while (q + 1 < g.n && (int)blockIdx.x >= g.blk0[q + 1]) ++q;
    const WdItem& it = g.it[q];
    const int l = blockIdx.x - g.blk0[q];
    if (l >= it.nx * it.ny) return;
    wgrad_dl_body<FI, FJ, KP, STAGES>(it.d, it.tilesI, it.tilesJ, it.per, l % it.nx, l / it.nx, false, smem);
}

template <typename K>
int raise_lds_once(K kern, int bytes, bool& done) {
    if (bytes > 64 * 1024 && !done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) {
            cdetr_set_error("cdetr_wgrad (direct-to-LDS): hipFuncSetAttribute(%d): %s", bytes, hipGetErrorString(e));
            return CDETR_ERR_LAUNCH;
        }
        done = true;
    }
    return CDETR_OK;
}

// slices of the pixel range so that the launch has ~target workgroups of >= min_kt pixel tiles each
inline void plan_slices(const cdetr_wgrad_desc& d, int BI, int BJ, int KP, long target, int& tilesI, int& tilesJ, int& per, int& slices) {
    tilesI = (d.Nout + BI - 1) / BI;
    tilesJ = (d.Cin + BJ - 1) / BJ;
    const int nkt = (d.P + KP - 1) / KP;
    const long base = (long)tilesI * tilesJ * d.taps;
    long s = (target + base - 1) / base;
    const long max_s = std::max(1, nkt / (KP == 32 ? 4 : 2));       // >= 128 pixels per slice
    s = std::max(1L, std::min(s, max_s));
    per = (int)((nkt + s - 1) / s);
    slices = (nkt + per - 1) / per;
}

template <int FI, int FJ, int KP, int STAGES>
int launch_wd(const cdetr_wgrad_desc& d, long target, hipStream_t st) {
    using Cf = WdCfg<FI, FJ, KP, STAGES>;
    static bool raised = false;
    if (int rc = raise_lds_once(wgrad_dl_kernel<FI, FJ, KP, STAGES>, Cf::LDS, raised)) return rc;
    int tilesI, tilesJ, per, slices;
    plan_slices(d, Cf::BI, Cf::BJ, KP, target, tilesI, tilesJ, per, slices);
    hipLaunchKernelGGL((wgrad_dl_kernel<FI, FJ, KP, STAGES>), dim3(tilesI * tilesJ * d.taps, slices), dim3(256), Cf::LDS, st, d, tilesI, tilesJ, per);
    return cdetr_launch_status("cdetr_wgrad");
}

template <int FI, int FJ, int KP, int STAGES>
int launch_wd_group(const cdetr_wgrad_desc* descs, const int* idx, int m, long target, hipStream_t st) {
    using Cf = WdCfg<FI, FJ, KP, STAGES>;
    static bool raised = false;
    if (int rc = raise_lds_once(wgrad_dl_group_kernel<FI, FJ, KP, STAGES>, Cf::LDS, raised)) return rc;
    for (int c0 = 0; c0 < m; c0 += WD_MAX) {
        const int n = std::min(WD_MAX, m - c0);
        WdArgs g;
        g.n = n;
        g.blk0[0] = 0;
        long work = 0;                                          // one common slice length: equal work per workgroup across the problems
        for (int k = 0; k < n; ++k) {
            const cdetr_wgrad_desc& d = descs[idx[c0 + k]];
            work += (long)((d.Nout + Cf::BI - 1) / Cf::BI) * ((d.Cin + Cf::BJ - 1) / Cf::BJ) * d.taps * ((d.P + KP - 1) / KP);
        }
        const long per_all = std::max<long>(KP == 32 ? 4 : 2, (work + target - 1) / target);
        for (int k = 0; k < n; ++k) {
            WdItem& it = g.it[k];
            it.d = descs[idx[c0 + k]];
            const int nkt = (it.d.P + KP - 1) / KP;
            it.tilesI = (it.d.Nout + Cf::BI - 1) / Cf::BI;
            it.tilesJ = (it.d.Cin + Cf::BJ - 1) / Cf::BJ;
            it.per = (int)std::min<long>(per_all, nkt);
            it.nx = it.tilesI * it.tilesJ * it.d.taps;
            it.ny = (nkt + it.per - 1) / it.per;
            it.pad_ = 0;
            g.blk0[k + 1] = g.blk0[k] + it.nx * it.ny;
        }
        hipLaunchKernelGGL((wgrad_dl_group_kernel<FI, FJ, KP, STAGES>), dim3(g.blk0[n]), dim3(256), Cf::LDS, st, g);
        if (int rc = cdetr_launch_status("cdetr_wgrad_group")) return rc;
    }
    return CDETR_OK;
}

}  // namespace

// operand formats / alignment the direct-to-LDS weight-gradient kernel needs (the dispatcher in igemm.hip decides whether it should run)
bool cdetr_wgrad_dl_eligible(const cdetr_wgrad_desc& d) {
    if (d.precision != 3 || !d.dY16 || !d.X16 || d.batch != 1 || d.dbias) return false;
    if ((d.Nout & 7) || (d.Cin & 7) || (d.ldy & 7) || (d.ldx & 7) || d.Nout < 32 || d.Cin < 32 || d.P < 64) return false;
    if ((reinterpret_cast<uintptr_t>(d.dY16) & 15) || (reinterpret_cast<uintptr_t>(d.X16) & 15)) return false;
    return true;
}

// cfg = tile * 100 + (KP / 32) * 10 + stages; tile: 0 = 128x128, 1 = 128x64 (Cout x Cin), 2 = 64x128, 3 = 64x64
#define WD_CASES(X)                                                                                                                  \
    X(0, 1, 3, 2, 2, 32, 3) X(0, 1, 4, 2, 2, 32, 4) X(0, 2, 3, 2, 2, 64, 3) X(0, 2, 2, 2, 2, 64, 2)                                   \
    X(1, 1, 3, 2, 1, 32, 3) X(1, 1, 4, 2, 1, 32, 4) X(1, 2, 3, 2, 1, 64, 3) X(2, 1, 3, 1, 2, 32, 3) X(2, 1, 4, 1, 2, 32, 4)           \
    X(2, 2, 3, 1, 2, 64, 3) X(3, 1, 3, 1, 1, 32, 3) X(3, 1, 4, 1, 1, 32, 4) X(3, 2, 3, 1, 1, 64, 3) X(3, 2, 4, 1, 1, 64, 4)

int cdetr_wgrad_dl_launch(const cdetr_wgrad_desc& d, int cfg, long target, hipStream_t st) {
#define X(T, KQ, S, FI, FJ, KP, ST) if (cfg == T * 100 + KQ * 10 + S) return launch_wd<FI, FJ, KP, ST>(d, target, st);
    WD_CASES(X)
#undef X
    cdetr_set_error("cdetr_wgrad (direct-to-LDS): no kernel for configuration %d", cfg);
    return CDETR_ERR_UNSUPPORTED;
}

int cdetr_wgrad_dl_group(const cdetr_wgrad_desc* descs, const int* idx, int m, int cfg, long target, hipStream_t st) {
#define X(T, KQ, S, FI, FJ, KP, ST) if (cfg == T * 100 + KQ * 10 + S) return launch_wd_group<FI, FJ, KP, ST>(descs, idx, m, target, st);
    WD_CASES(X)
#undef X
    cdetr_set_error("cdetr_wgrad_group (direct-to-LDS): no kernel for configuration %d", cfg);
    return CDETR_ERR_UNSUPPORTED;
}
