// igemm_dl.hip -- the tile GEMM of the backbone with DIRECT-TO-LDS operand loads (gfx950 `global_load_lds_dwordx4`).
//
// Same contraction and fused epilogue as igemm_fast_kernel (igemm.hip; math: A2/models/resnet.py:140-160 conv + FrozenBN + ReLU
// + residual, and their data gradients), for operands that ALREADY sit in HBM in the matrix pipe's input format:
//   A  = bf16 planes of the activation / activation gradient, written by the producing epilogue (cdetr_gemm_desc.A16 = hi,
//        A16lo = lo = bf16(x - hi); plain-bf16 products read the hi plane only),
//   B  = the pre-split weight image of cdetr_weight_mirror (cdetr_gemm_desc.B_split: groups of 32 k, [hi 32 | lo 32]).
// A k-tile is then a pure byte copy HBM -> LDS: every lane issues `global_load_lds_dwordx4` (16 bytes = 8 bf16 of one row, DMA'd
// into LDS without touching a VGPR), no conversion, no staging VALU, no ds_write; STAGES tiles ring in LDS, the loads of tile
// t + STAGES - 1 are issued right after the barrier that frees its slot and stay in flight across the next barriers (counted
// `s_waitcnt vmcnt(N)`, raw `s_barrier`): one barrier per k-tile, and between two barriers a wave issues 8-24 MFMAs.
//
// LDS image of one operand tile: [row][8 slots of 16 B] = 128 bytes of k per row (32 k x {hi, lo} for the split product, 64 k of hi
// for plain bf16).  The DMA writes lane l of a wave at base + 16 l, so a wave fills 8 complete rows; WHICH 16-byte chunk of the row a
// lane fetches is free (the global address is per lane), so the bank swizzle lives on the source side: slot s of row r holds logical
// chunk c = s ^ ((r >> 1) & 7).  A fragment read (32 rows x one chunk, ds_read_b128) then touches 16 distinct 16-byte slots in every
// 16-lane service group of MI355X_MICROARCH.md's LDS table: conflict-free.
//
// MFMA operand order: the WEIGHT fragment is the A operand and the activation fragment the B operand of v_mfma_f32_32x32x16_bf16,
// so an accumulator lane holds ONE output row m and, per register quad, four CONSECUTIVE output channels n: the epilogue reads
// residual / gate and writes C (fp32), C16 and C16lo with 16- / 8-byte accesses per lane instead of 48 scalar ones per fragment.
#include "../../include/cdetr_hip.h"
#include "common.h"
#include "rows.h"
#include "dl_common.h"
#include <stdlib.h>

namespace {

template <int FM, int FN, int TERMS, int STAGES>
struct DlCfg {
    static constexpr int BM = 64 * FM, BN = 64 * FN;          // 2 x 2 waves, FM x FN fragments of 32 x 32 each
    static constexpr int NA = BM / 32, NB = BN / 32;          // 16-byte pieces per thread per k-tile (A, B)
    static constexpr int KT = (TERMS == 3) ? 32 : 64;         // k-values per tile: 128 bytes per row either way
    static constexpr int A_STAGE = BM * 128, B_STAGE = BN * 128;
    static constexpr int RING = STAGES * (A_STAGE + B_STAGE);
    static constexpr int STAGING = 4 * 32 * FM * (32 * FN + 4) * 4;       // the epilogue's transposition tiles (one per wave) reuse the ring
    static constexpr int LDS = RING > STAGING ? RING : STAGING;
};

// PROBE: wave 0 of every workgroup stamps s_memtime at its phase boundaries into cdetr_gemm_desc.splitk_ws (8 x uint64 per workgroup:
// start, prologue issued, first tile landed, k-loop done, epilogue operands loaded, stores issued) -- tools/dl_probe.py.
// EARLY: the epilogue's operands (bias, residual rows, gate rows -- nothing the k-loop computes) are requested BEFORE the first operand
// tile, so their HBM round trip (2.9 of a 64 x 64 / K = 256 workgroup's 11.7 us in the phase probe, profiles/r3_dl_probe_v1.txt) runs under
// the k-loop; loads complete in issue order, so the counted vmcnt waits of the ring stay valid (the older loads have landed by then).
//
// NHS > 0: the HALO-RESIDENT form for 3x3 convolutions of stride 1 (forward rows and data-gradient rows; any dilation, pad == dil).  The
// classic form fetches the pixel operand once per filter tap: 9 x BM rows per channel chunk, although the nine taps of a tile of BM
// CONSECUTIVE flattened pixels only touch the BM + 2 dil (W + 1) consecutive rows around it.  Here that row range (the "halo") is DMA'd
// once per channel chunk -- double-buffered, its pieces riding in the per-step load batches of the previous chunk -- and the nine taps are
// nine fragment reads at a row offset (y-tap * W + x-tap) * dil; taps that leave the image (or the row range of the tensor) read a zero row
// of LDS instead (per-lane address select, no masking VALU).  The weights stream through the 3-deep ring as before, one (chunk, tap)
// tile per step.  Operand bytes per step of a 64 x 64 tile: 8 KB + halo / 9 (2.4-3.8 KB) instead of 16 KB; the row swizzle stays
// conflict-free for any row offset (16 consecutive rows x one logical chunk = 16 distinct (row parity, slot) pairs).
// halo_R = dil (W + 1) rows in front of / behind the tile, nah = 32-row passes of the halo (<= 7 NHS: its pieces ride in taps 2..8).
//
// SPLIT: the reduction of one output tile cut into gridDim.y slices (blockIdx.y = slice) -- the problems whose grid leaves the CUs with one
// wave per SIMD (5000 x 256 outputs on 64 x 64 tiles: 316 workgroups; the k-step of a workgroup is a latency chain [wait, barrier, 8 LDS reads,
// dependent MFMAs] that only OTHER resident workgroups can overlap).  Same exchange as igemm_fast_kernel's (igemm.hip): every slice parks
// its partial accumulators in cdetr_gemm_desc.splitk_ws as device-scope write-through stores, counts itself in on the tile's arrival
// counter, and the slice that arrives LAST adds the partials in slice order (a sum that does not depend on who is last), fetches the
// epilogue's operands -- only now: the other slices never touch them -- and runs the fused epilogue.  No slice waits for another.
template <int FM, int FN, int TERMS, int STAGES, bool PROBE = false, bool EARLY = true, int NHS = 0, bool SPLIT = false>
__global__ __launch_bounds__(256) void igemm_dl_kernel(const cdetr_gemm_desc d, const int tilesM, const int tilesN, const int halo_R, const int nah) {
    using Cf = DlCfg<FM, FN, TERMS, STAGES>;
    constexpr bool HALO = NHS > 0;
    static_assert(!HALO || STAGES == 3, "the halo form runs the weight ring three deep");
    static_assert(!SPLIT || (!HALO && !PROBE && !EARLY), "split reductions: classic form, epilogue operands fetched by the finishing slice");
    unsigned long long* probe = PROBE ? reinterpret_cast<unsigned long long*>(d.splitk_ws) + (long)blockIdx.x * 8 : nullptr;
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if constexpr (PROBE) {
            if (threadIdx.x == 0) probe[slot] = __builtin_amdgcn_s_memtime();
        }
    };
    stamp(0);
    if (d.flags & CDETR_GEMM_PRIO) __builtin_amdgcn_s_setprio(2);      // main-chain launch beside another stream's flood: first at the instruction arbiter
    constexpr int BM = Cf::BM, BN = Cf::BN, NA = Cf::NA, NB = Cf::NB, KT = Cf::KT;
    constexpr int A_STAGE = Cf::A_STAGE, B_STAGE = Cf::B_STAGE;
    constexpr int NI = NA + NB;                                // loads per wave per k-tile
    static_assert(TERMS == 3 || TERMS == 1, "split-bf16 x3 or plain bf16");
    static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // the ONLY LDS object (a second one makes hipcc drain vmcnt before every ds_read)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int i32 = lane & 31, g = lane >> 5;
    // XCD-aware tile order (as igemm_fast_kernel): workgroup b runs on XCD b % 8; XCD x owns a band of row-tiles and sweeps the
    // column-tiles of a row-tile back to back, so an activation row-panel is fetched once per XCD and the weights stay L2-resident
    const int bx = blockIdx.x;
    const int band = (tilesM + 7) >> 3;
    const int xcd = bx & 7, jloc = bx >> 3;
    const int tm = xcd * band + jloc / tilesN, tn = jloc % tilesN;
    if (tm >= tilesM) return;
    const int m0 = tm * BM, n0 = tn * BN;
    const int K = d.K, taps = d.taps;
    const int nkt_tap = K / KT;
    const int ksplit = SPLIT ? (int)gridDim.y : 1, kslice = SPLIT ? (int)blockIdx.y : 0;
    // slice kslice owns k-tiles [kt0, kt0 + nkt) of the nkt_tap * taps tiles of the reduction (one slice: all of them)
    const int kt0 = SPLIT ? (int)((long)nkt_tap * taps * kslice / ksplit) : 0;
    const int nkt = SPLIT ? (int)((long)nkt_tap * taps * (kslice + 1) / ksplit) - kt0 : nkt_tap * taps;

    // ---------------------------------------------------------------- epilogue geometry + its operand loads
    // (the accumulators are transposed through LDS at the end so that LR = 4 FN lanes hold 8 consecutive channels each of ONE row)
    constexpr int LDW = 32 * FN + 4;                            // floats per staged row
    constexpr int LR = 4 * FN;                                  // lanes per output row (8 channels each)
    constexpr int RPP = 64 / LR;                                // rows per pass of the wave
    constexpr int NPASS = 32 * FM / RPP;
    static_assert(32 * FM * LDW * 4 * 4 <= Cf::LDS, "the staging tiles fit the ring");
    const int rr = lane / LR, cc = (lane % LR) * 8;
    const int n = n0 + wn * 32 * FN + cc;
    const bool v0 = n < d.N, v1 = n + 4 < d.N;                  // the two 4-channel halves of this lane's 8 channels (N % 4 == 0)
    const int nl0 = min(n, d.N - 4), nl1 = min(n + 4, d.N - 4);
    const int mw = m0 + wm * 32 * FM;
    const __bf16* __restrict__ g16 = d.gate ? reinterpret_cast<const __bf16*>(d.gate16) : nullptr;      // bf16 twin of the gate: same sign, half the bytes
    float4 bia[2], res[NPASS][2];
    uint2 gat16[NPASS][2];
    // CDETR_GEMM_RESID_GROUPS: the residual is stored as interleaved groups [hi 32 | lo 32] (a block output written by CDETR_GEMM_C_GROUPS):
    // 16 bytes of hi + 16 bytes of lo per lane and pass instead of two float4 -- same registers, decoded (hi + lo) in the epilogue
    const bool r_il = d.resid && (d.flags & CDETR_GEMM_RESID_GROUPS);
    const int ng = min(n, d.N - 8);
    // residual / gate-twin rows of every pass: ONE batch of unconditional loads (clamped row / channel, masked use).  An fp32 gate without
    // a twin (not a product configuration: the kernel's operands ARE twins) is fetched in the epilogue: twice the registers for the k-loop
    auto load_epilogue_operands = [&]() __attribute__((always_inline)) {
        bia[0] = d.bias ? *reinterpret_cast<const float4*>(d.bias + nl0) : make_float4(0.f, 0.f, 0.f, 0.f);
        bia[1] = d.bias ? *reinterpret_cast<const float4*>(d.bias + nl1) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const long mrow = min(mw + p * RPP + rr, d.M - 1);
            if (r_il) {
                const unsigned char* rp = reinterpret_cast<const unsigned char*>(d.resid) + (mrow * d.ldr + (ng & ~31)) * 4 + (ng & 31) * 2;
                res[p][0] = *reinterpret_cast<const float4*>(rp);
                res[p][1] = *reinterpret_cast<const float4*>(rp + 64);
            } else {
                res[p][0] = d.resid ? *reinterpret_cast<const float4*>(d.resid + mrow * d.ldr + nl0) : make_float4(0.f, 0.f, 0.f, 0.f);
                res[p][1] = d.resid ? *reinterpret_cast<const float4*>(d.resid + mrow * d.ldr + nl1) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            gat16[p][0] = g16 ? *reinterpret_cast<const uint2*>(g16 + mrow * d.ldg + nl0) : make_uint2(0x3f803f80u, 0x3f803f80u);
            gat16[p][1] = g16 ? *reinterpret_cast<const uint2*>(g16 + mrow * d.ldg + nl1) : make_uint2(0x3f803f80u, 0x3f803f80u);
        }
    };
    if constexpr (EARLY) load_epilogue_operands();

    // ---------------------------------------------------------------- loader state: which 16 bytes this lane fetches
    const int prow = tid >> 3;                                  // row inside a 32-row pass (one pass = one load per thread)
    const int chunk = (tid & 7) ^ ((tid >> 4) & 7);             // logical chunk behind LDS slot (tid & 7) of row prow (+ 32 j)
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(dl_zero_page) + (tid & 7) * 16;
    // A: interleaved groups [hi 32 | lo 32] (A_split: one full line per row and k-tile), or the hi / lo planes (two half lines)
    const bool a_il = TERMS == 3 && (d.flags & CDETR_GEMM_A_GROUPS);
    const __bf16* Apl = reinterpret_cast<const __bf16*>((TERMS == 3 && (chunk >> 2) && !a_il) ? d.A16lo : d.A16);
    const int a_kk = a_il ? chunk * 8 : (TERMS == 3) ? (chunk & 3) * 8 : chunk * 8;
    const long a_ld = a_il ? 2 * d.lda : d.lda;                  // row stride in bf16 elements
    const int a_kb = a_il ? 4 : 2;                              // bytes along k per k-value
    RowCoord arow[NA];
    const unsigned char* ap[NA];
    unsigned amask = 0;
    if constexpr (!HALO) {
#pragma unroll
        for (int j = 0; j < NA; ++j) arow[j] = decode_row(d.g, m0 + j * 32 + prow, d.M);
    }
    auto set_tap = [&](int tap) {
        amask = 0;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const long row = gather_row(d.g, arow[j], tap);
            ap[j] = row >= 0 ? reinterpret_cast<const unsigned char*>(Apl + row * a_ld + a_kk) : zero;
            amask |= (row >= 0 ? 1u : 0u) << j;
        }
    };
    // halo form: piece q of the halo = rows 32 q + prow of [m0 - halo_R, m0 + BM + halo_R); rows outside the tensor come from the zero page
    const int HB = nah * 4096;                                  // bytes of one halo buffer
    const int offB = HALO ? 2 * HB : STAGES * A_STAGE;          // the weight ring behind the pixel operand's buffers
    const int offZ = offB + STAGES * B_STAGE;                   // halo form: 128 zero bytes (the row of every out-of-image tap), then a 4 KB dump
    const unsigned char* hsrc = nullptr;
    unsigned hmask = 0;
    if constexpr (HALO) {
        const long r0 = (long)m0 - halo_R + prow;
        hsrc = reinterpret_cast<const unsigned char*>(Apl) + (r0 * a_ld + a_kk) * 2;
        const int HR = BM + 2 * halo_R;
        for (int q = 0; q < nah; ++q) {
            const long r = r0 + 32 * q;
            hmask |= ((32 * q + prow < HR && r >= 0 && r < d.M) ? 1u : 0u) << q;
        }
        if (tid < 32) *reinterpret_cast<unsigned*>(smem + offZ + tid * 4) = 0u;      // (ordered before its first use by the first barrier)
    }
    auto issue_halo = [&](int q, int buf, int kc) __attribute__((always_inline)) {      // q < nah
        const unsigned char* src = ((hmask >> q) & 1u) ? hsrc + ((long)q * 32 * a_ld) * 2 + kc * a_kb : zero;
        __builtin_amdgcn_global_load_lds((gbl_vp)src, (lds_vp)(smem + buf * HB + q * 4096 + w * 1024), 16, 0, 0);
    };
    const unsigned char* bp[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = min(n0 + j * 32 + prow, d.N - 1);
        if (TERMS == 1 && d.B16)    // plain bf16 image: 64 k = one full 128-byte line per row and k-tile
            bp[j] = reinterpret_cast<const unsigned char*>(d.B16) + (long)n * d.ldb * 2 + chunk * 16;
        else                        // pre-split groups [hi 32 | lo 32]: both halves (x3) or the hi halves of two groups (plain bf16)
            bp[j] = reinterpret_cast<const unsigned char*>(d.B_split) + (long)n * d.ldb * 4 +
                    ((TERMS == 3) ? chunk * 16 : (chunk >> 2) * 128 + (chunk & 3) * 16);
    }
    const int bshift = (TERMS == 1 && d.B16) ? 1 : 2;           // log2(bytes per k of the weight image)
    auto issue = [&](int stage, int tap, int kc) __attribute__((always_inline)) {
        const int koa = kc * a_kb;                              // bytes along k of the A planes / groups
        const long kob = ((long)tap * K + kc) << bshift;        // bytes along (tap, k) of the weight image
        unsigned char* la = smem + stage * A_STAGE + w * 1024;
        unsigned char* lb = smem + offB + stage * B_STAGE + w * 1024;
        if constexpr (!HALO) {
#pragma unroll
            for (int j = 0; j < NA; ++j)
                __builtin_amdgcn_global_load_lds((gbl_vp)(ap[j] + (((amask >> j) & 1u) ? koa : 0)), (lds_vp)(la + j * 4096), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
            __builtin_amdgcn_global_load_lds((gbl_vp)(bp[j] + kob), (lds_vp)(lb + j * 4096), 16, 0, 0);
    };

    // ---------------------------------------------------------------- fragment addresses
    const int fx = (i32 >> 1) & 7;
    int xo[4], wo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int s16 = ((2 * j + g) ^ fx) << 4;
        xo[j] = (wm * 32 * FM + i32) * 128 + s16;
        wo[j] = offB + (wn * 32 * FN + i32) * 128 + s16;
    }
    // halo form: per fragment row its pixel coordinates -> 9 validity bits; the row of tap (ty, tx) sits (ty W + tx) dil rows further
    unsigned vmask[FM];
    const int hsgn = d.g.mode == CDETR_ROWS_CONV_FWD ? d.g.dil : -d.g.dil;      // forward rows read pixel + (k - 1) dil, data-gradient rows pixel - (k - 1) dil
    if constexpr (HALO) {
#pragma unroll
        for (int a = 0; a < FM; ++a) {
            const int m = m0 + wm * 32 * FM + a * 32 + i32;
            const int hw = d.g.Hc * d.g.Wc;
            const int rem = m % hw;
            const int y = rem / d.g.Wc, x = rem - y * d.g.Wc;
            vmask[a] = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + (t / 3 - 1) * hsgn, xx = x + (t % 3 - 1) * hsgn;
                vmask[a] |= ((m < d.M && yy >= 0 && yy < d.g.Hc && xx >= 0 && xx < d.g.Wc) ? 1u : 0u) << t;
            }
        }
    }
    f32x16 acc[FN][FM];
#pragma unroll
    for (int b = 0; b < FN; ++b)
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][a][r] = 0.f;

    auto compute = [&](int stage, int hbuf = 0, int tap = 0) __attribute__((always_inline)) {
        const unsigned char* xs = smem + (HALO ? hbuf * HB : stage * A_STAGE);
        const unsigned char* ws = smem + stage * B_STAGE;
        int xh_o[FM][4];
        if constexpr (HALO) {
            const int ty = tap / 3, tx = tap - 3 * ty;
            const int delta = ((ty - 1) * d.g.Wc + (tx - 1)) * hsgn;
#pragma unroll
            for (int a = 0; a < FM; ++a) {
                const int h = wm * 32 * FM + a * 32 + i32 + halo_R + delta;
                const bool ok = (vmask[a] >> tap) & 1u;
                const int rb = ok ? h * 128 : offZ - hbuf * HB;
                const int hx = ok ? (h >> 1) & 7 : 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) xh_o[a][j] = rb + (((2 * j + g) ^ hx) << 4);
            }
        }
#pragma unroll
        for (int hp = 0; hp < KT / 16; ++hp) {
            bf16x8 xh[FM], xl[FM], wh[FN], wl[FN];
#pragma unroll
            for (int a = 0; a < FM; ++a) {
                if constexpr (HALO) {
                    xh[a] = *reinterpret_cast<const bf16x8*>(xs + xh_o[a][hp]);
                    if constexpr (TERMS == 3) xl[a] = *reinterpret_cast<const bf16x8*>(xs + xh_o[a][2 + hp]);
                } else {
                    xh[a] = *reinterpret_cast<const bf16x8*>(xs + xo[hp] + a * 4096);
                    if constexpr (TERMS == 3) xl[a] = *reinterpret_cast<const bf16x8*>(xs + xo[2 + hp] + a * 4096);
                }
            }
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                wh[b] = *reinterpret_cast<const bf16x8*>(ws + wo[hp] + b * 4096);
                if constexpr (TERMS == 3) wl[b] = *reinterpret_cast<const bf16x8*>(ws + wo[2 + hp] + b * 4096);
            }
#pragma unroll
            for (int b = 0; b < FN; ++b)
#pragma unroll
                for (int a = 0; a < FM; ++a) acc[b][a] = mfma_bf16_terms<TERMS>(wh[b], wl[b], xh[a], xl[a], acc[b][a]);
        }
    };

    // ---------------------------------------------------------------- the pipeline
    int f_tap = kt0 / nkt_tap, f_kc = (kt0 - f_tap * nkt_tap) * KT, f_idx = 0;      // coordinates of the next tile to issue
    auto issue_next = [&](int stage) __attribute__((always_inline)) {
        issue(stage, f_tap, f_kc);
        ++f_idx;
        f_kc += KT;
        if (f_kc == K) {
            f_kc = 0;
            ++f_tap;
            if (f_tap < taps) set_tap(f_tap);
        }
    };
    if constexpr (HALO) {
        // step s = (channel chunk c = s / 9, tap t = s % 9).  Batch b = the weight tile of step b (NB loads per thread) + for t >= 2, while a
        // next chunk exists, NHS pieces of ITS halo (the other halo buffer was last read in step 9 c - 1; batch b is issued behind the
        // barrier of step b - 2, so batches t = 0, 1 must not touch it).  Loads complete in order: at the top of step s everything up to
        // batch s has landed once at most size(batch s + 1) loads are outstanding.
        const int nchunks = nkt_tap, nsteps = nkt;
        auto batch_size = [&](int t, int c) { return NB + ((t >= 2 && c + 1 < nchunks) ? NHS : 0); };
        int bt = 0, bc = 0, bidx = 0;                            // (tap, chunk, index) of the next batch to issue
        auto issue_batch = [&](int stage) __attribute__((always_inline)) {
            issue(stage, bt, bc * KT);
            if (bt >= 2 && bc + 1 < nchunks) {
#pragma unroll
                for (int i = 0; i < NHS; ++i) {
                    const int q = (bt - 2) * NHS + i;
                    if (q < nah) issue_halo(q, (bc + 1) & 1, (bc + 1) * KT);
                    else __builtin_amdgcn_global_load_lds((gbl_vp)zero, (lds_vp)(smem + offZ + 256 + w * 1024), 16, 0, 0);
                }
            }
            ++bidx;
            if (++bt == 9) { bt = 0; ++bc; }
        };
        for (int q = 0; q < nah; ++q) issue_halo(q, 0, 0);
        issue_batch(0);
        if (nsteps > 1) issue_batch(1);
        stamp(1);
        int st = 0, sn = 2, t = 0, c = 0;
        for (int s = 0; s < nsteps; ++s) {
            int tn = t + 1, cn = c;
            if (tn == 9) { tn = 0; ++cn; }
            wait_vm(s + 1 < nsteps ? batch_size(tn, cn) : 0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PROBE) { if (s == 0) stamp(2); }
            if (bidx < nsteps) issue_batch(sn);
            compute(st, c & 1, t);
            __builtin_amdgcn_sched_barrier(0);
            st = (st + 1 == 3) ? 0 : st + 1;
            sn = (sn + 1 == 3) ? 0 : sn + 1;
            t = tn;
            c = cn;
        }
    }
    if constexpr (!HALO) set_tap(f_tap);
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (!HALO && s < nkt) issue_next(s);
    if constexpr (!HALO) stamp(1);
    int st = 0, sn = STAGES - 1;                                 // stage of the tile being computed / of the tile being issued
    for (int kt = 0; !HALO && kt < nkt; ++kt) {
        // tiles kt .. min(kt + STAGES - 2, nkt - 1) are in flight; tile kt must have landed (this wave's pieces), the others may fly on
        const int fly = min(STAGES - 2, nkt - 1 - kt);
        if (fly >= 2) wait_vm(2 * NI);
        else if (fly == 1) wait_vm(NI);
        else wait_vm(0);
        __builtin_amdgcn_s_barrier();                           // every wave's pieces of tile kt are in LDS; everyone is done reading slot sn
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PROBE) { if (kt == 0) stamp(2); }
        if (f_idx < nkt) issue_next(sn);
        compute(st);
        __builtin_amdgcn_sched_barrier(0);
        st = (st + 1 == STAGES) ? 0 : st + 1;
        sn = (sn + 1 == STAGES) ? 0 : sn + 1;
    }
    mfma_drain(acc);
    stamp(3);

    // ---------------------------------------------------------------- epilogue
    // The accumulators (lane = output row m, register quad = 4 consecutive channels) are transposed through LDS -- the ring is free
    // once every wave is past its last fragment read -- so that 4 FN lanes hold 8 consecutive channels each of ONE row: residual /
    // gate loads and the C / C16 / C16lo stores are full 128- / 256-byte row segments (8 rows per instruction) instead of 16 bytes
    // in each of 32 rows.  Staging rows are padded by 16 bytes: the float4 writes of 8 consecutive lanes (8 rows, same column) and
    // the row-major float4 reads both spread over the banks.
    __builtin_amdgcn_s_barrier();                               // every wave has finished reading the ring
    if constexpr (SPLIT) {
        // Coherence across the 8 XCDs without fences (a device-scope fence writes back / invalidates a whole L2): the partials travel as
        // relaxed device-scope atomic stores / loads (sc1), s_waitcnt orders them before the arrival count, a device-scope RMW.
        // (gfx950-specific ordering, not a HIP memory-model guarantee: include/cdetr_hip.h, cdetr_gemm_desc.splitk_ws; tools/splitk_stress.py)
        constexpr int FR = FM * FN * 16;
        int* cnt = reinterpret_cast<int*>(d.splitk_ws);
        float* wsp = reinterpret_cast<float*>(cnt + SPLITK_COUNTERS);
        const int tile = tm * tilesN + tn;
        float* mine = wsp + ((long)tile * ksplit + kslice) * FR * 256 + tid;
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __hip_atomic_store(mine + ((b * FM + a) * 16 + r) * 256, acc[b][a][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);       // the stores above have been acknowledged
        __syncthreads();                     // ... by every wave
        int* flag = reinterpret_cast<int*>(smem + Cf::LDS);      // 16 bytes behind the ring / staging tiles (launch_dl)
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (old == ksplit - 1) ? 1 : 0;
            if (last) __hip_atomic_store(cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // leave the counters zero for the next launch
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0) return;
        load_epilogue_operands();
        if (ksplit == 2) {                   // two slices: a + b == b + a, in place
            const float* other = wsp + ((long)tile * ksplit + (1 - kslice)) * FR * 256 + tid;
#pragma unroll
            for (int b = 0; b < FN; ++b)
#pragma unroll
                for (int a = 0; a < FM; ++a) {
                    float t[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[r] = __hip_atomic_load(other + ((b * FM + a) * 16 + r) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[b][a][r] += t[r];
                }
        } else {                             // slice order: ((p0 + p1) + p2) + p3 whichever slice finishes
            f32x16 sum[FN][FM];
            for (int q = 0; q < ksplit; ++q) {
                const float* other = wsp + ((long)tile * ksplit + q) * FR * 256 + tid;
#pragma unroll
                for (int b = 0; b < FN; ++b)
#pragma unroll
                    for (int a = 0; a < FM; ++a) {
                        float t[16];
                        if (q == kslice) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) t[r] = acc[b][a][r];
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) t[r] = __hip_atomic_load(other + ((b * FM + a) * 16 + r) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) sum[b][a][r] = (q == 0) ? t[r] : sum[b][a][r] + t[r];
                    }
            }
#pragma unroll
            for (int b = 0; b < FN; ++b)
#pragma unroll
                for (int a = 0; a < FM; ++a) acc[b][a] = sum[b][a];
        }
    }
    float* stg = reinterpret_cast<float*>(smem) + w * (32 * FM * LDW);
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(stg + (a * 32 + i32) * LDW + b * 32 + 8 * q + 4 * g) =
                    make_float4(acc[b][a][4 * q], acc[b][a][4 * q + 1], acc[b][a][4 * q + 2], acc[b][a][4 * q + 3]);
    float* __restrict__ C = d.C;
    __bf16* __restrict__ C16 = reinterpret_cast<__bf16*>(d.C16);
    const bool c_il = d.flags & CDETR_GEMM_C_GROUPS;
    __bf16* __restrict__ C16lo = c_il ? nullptr : reinterpret_cast<__bf16*>(d.C16lo);
    unsigned char* __restrict__ Cil = c_il ? reinterpret_cast<unsigned char*>(d.C16lo) : nullptr;       // (N % 32 == 0: a lane's 8 channels never straddle a group)
    if constexpr (!EARLY && !SPLIT) load_epilogue_operands();
    float4 gat32[NPASS][2];
    if (d.gate && !g16) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const long mrow = min(mw + p * RPP + rr, d.M - 1);
            gat32[p][0] = *reinterpret_cast<const float4*>(d.gate + mrow * d.ldg + nl0);
            gat32[p][1] = *reinterpret_cast<const float4*>(d.gate + mrow * d.ldg + nl1);
        }
    }
    const float bias8[8] = {bia[0].x, bia[0].y, bia[0].z, bia[0].w, bia[1].x, bia[1].y, bia[1].z, bia[1].w};
    if constexpr (PROBE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(4); }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int row = p * RPP + rr;
        const int m = mw + row;
        const float4 s0 = *reinterpret_cast<const float4*>(stg + row * LDW + cc);
        const float4 s1 = *reinterpret_cast<const float4*>(stg + row * LDW + cc + 4);
        const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        float re[8] = {res[p][0].x, res[p][0].y, res[p][0].z, res[p][0].w, res[p][1].x, res[p][1].y, res[p][1].z, res[p][1].w};
        if (r_il) {
            const unsigned hq[4] = {__float_as_uint(res[p][0].x), __float_as_uint(res[p][0].y), __float_as_uint(res[p][0].z), __float_as_uint(res[p][0].w)};
            const unsigned lq[4] = {__float_as_uint(res[p][1].x), __float_as_uint(res[p][1].y), __float_as_uint(res[p][1].z), __float_as_uint(res[p][1].w)};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                re[2 * q] = __uint_as_float(hq[q] << 16) + __uint_as_float(lq[q] << 16);
                re[2 * q + 1] = __uint_as_float(hq[q] & 0xffff0000u) + __uint_as_float(lq[q] & 0xffff0000u);
            }
        }
        float ga[8] = {__uint_as_float(gat16[p][0].x << 16), __uint_as_float(gat16[p][0].x & 0xffff0000u), __uint_as_float(gat16[p][0].y << 16),
                       __uint_as_float(gat16[p][0].y & 0xffff0000u), __uint_as_float(gat16[p][1].x << 16), __uint_as_float(gat16[p][1].x & 0xffff0000u),
                       __uint_as_float(gat16[p][1].y << 16), __uint_as_float(gat16[p][1].y & 0xffff0000u)};
        if (d.gate && !g16) {
            const float4 t0 = gat32[p][0], t1 = gat32[p][1];
            ga[0] = t0.x; ga[1] = t0.y; ga[2] = t0.z; ga[3] = t0.w; ga[4] = t1.x; ga[5] = t1.y; ga[6] = t1.z; ga[7] = t1.w;
        }
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = (sv[e] + bias8[e]) * d.out_scale + re[e];
            t = ga[e] > 0.f ? t : 0.f;
            if (d.relu) t = fmaxf(t, 0.f);
            v[e] = t;
        }
        if (m < d.M && v0) {
            const long o = (long)m * d.ldc + n;
            unsigned hh[4], ll[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split_bf16_pk(v[2 * e], v[2 * e + 1], hh[e], ll[e]);
            const u32x4 h = {hh[0], hh[1], hh[2], hh[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
            if (C) {
                *reinterpret_cast<float4*>(C + o) = make_float4(v[0], v[1], v[2], v[3]);
                if (v1) *reinterpret_cast<float4*>(C + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (Cil) {          // interleaved groups: 8 consecutive channels = 16 bytes of the hi half + 16 bytes of the lo half of one group
                unsigned char* gp = Cil + ((long)m * d.ldc + (n & ~31)) * 4 + (n & 31) * 2;
                *reinterpret_cast<u32x4*>(gp) = h;
                *reinterpret_cast<u32x4*>(gp + 64) = l;
            }
            if (C16) {
                if (v1 && ((o & 7) == 0)) {
                    *reinterpret_cast<u32x4*>(C16 + o) = h;
                    if (C16lo) *reinterpret_cast<u32x4*>(C16lo + o) = l;
                } else {
                    *reinterpret_cast<uint2*>(C16 + o) = make_uint2(h[0], h[1]);
                    if (C16lo) *reinterpret_cast<uint2*>(C16lo + o) = make_uint2(l[0], l[1]);
                    if (v1) {
                        *reinterpret_cast<uint2*>(C16 + o + 4) = make_uint2(h[2], h[3]);
                        if (C16lo) *reinterpret_cast<uint2*>(C16lo + o + 4) = make_uint2(l[2], l[3]);
                    }
                }
            }
        }
    }
    stamp(5);
    if constexpr (PROBE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(6); }
}

template <int FM, int FN, int TERMS, int STAGES, bool PROBE = false, bool EARLY = true, bool SPLIT = false>
int launch_dl(const cdetr_gemm_desc& d, hipStream_t st, int ksplit = 1) {
    using Cf = DlCfg<FM, FN, TERMS, STAGES>;
    const int tilesM = (d.M + Cf::BM - 1) / Cf::BM, tilesN = (d.N + Cf::BN - 1) / Cf::BN;
    auto kern = igemm_dl_kernel<FM, FN, TERMS, STAGES, PROBE, EARLY, 0, SPLIT>;
    if (SPLIT) {
        const long need = (long)SPLITK_COUNTERS * 4 + (long)tilesM * tilesN * ksplit * Cf::BM * Cf::BN * 4;
        const int nkt_all = d.K / Cf::KT * d.taps;
        if (!d.splitk_ws || d.splitk_ws_bytes < need || (long)tilesM * tilesN > SPLITK_COUNTERS || ksplit < 2 || ksplit > 8 || ksplit > nkt_all) {
            cdetr_set_error("cdetr_gemm (direct-to-LDS, split reduction): %d slices of %d k-tiles need splitk_ws >= %ld bytes (have %ld) and <= %d output tiles",
                            ksplit, nkt_all, need, (long)d.splitk_ws_bytes, SPLITK_COUNTERS);
            return CDETR_ERR_ARG;
        }
    }
    constexpr int lds = Cf::LDS + (SPLIT ? 16 : 0);           // split reductions: the arrival flag
    if (lds > 64 * 1024) {
        static bool raised = false;                             // per instantiation
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) {
                cdetr_set_error("cdetr_gemm (direct-to-LDS): hipFuncSetAttribute(%d): %s", lds, hipGetErrorString(e));
                return CDETR_ERR_LAUNCH;
            }
            raised = true;
        }
    }
    dim3 grid(8 * ((tilesM + 7) / 8) * tilesN, SPLIT ? ksplit : 1), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, st, d, tilesM, tilesN, 0, 0);
    return cdetr_launch_status("cdetr_gemm");
}

// halo-resident 3x3 form: LDS = two halo buffers of nah 32-row passes + the 3-deep weight ring + zero row + dump
template <int FM, int FN, int TERMS, int NHS>
int launch_dl_halo(const cdetr_gemm_desc& d, int halo_R, int nah, hipStream_t st) {
    using Cf = DlCfg<FM, FN, TERMS, 3>;
    const int tilesM = (d.M + Cf::BM - 1) / Cf::BM, tilesN = (d.N + Cf::BN - 1) / Cf::BN;
    auto kern = igemm_dl_kernel<FM, FN, TERMS, 3, false, true, NHS>;
    int lds = 2 * nah * 4096 + 3 * Cf::B_STAGE + 256 + 4096;
    if (lds < Cf::STAGING) lds = Cf::STAGING;
    static int raised = 64 * 1024;                              // per instantiation
    if (lds > raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            cdetr_set_error("cdetr_gemm (direct-to-LDS, halo): hipFuncSetAttribute(%d): %s", lds, hipGetErrorString(e));
            return CDETR_ERR_LAUNCH;
        }
        raised = lds;
    }
    dim3 grid(8 * ((tilesM + 7) / 8) * tilesN), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, st, d, tilesM, tilesN, halo_R, nah);
    return cdetr_launch_status("cdetr_gemm");
}

}  // namespace

// Whether the direct-to-LDS tile kernel can run this problem (operand formats / alignment); the dispatcher in igemm.hip decides
// whether it SHOULD (problem size) and which tile.
bool cdetr_gemm_dl_eligible(const cdetr_gemm_desc& d) {
    if (d.b_layout != 0 || d.batch != 1 || !d.A16) return false;
    if (!d.B_split && !(d.precision == 3 && d.B16)) return false;
    if (d.B16 && ((reinterpret_cast<uintptr_t>(d.B16) & 15) || (d.ldb & 7))) return false;
    if (d.precision != 1 && d.precision != 3) return false;
    const bool a_groups = d.flags & CDETR_GEMM_A_GROUPS, c_groups = d.flags & CDETR_GEMM_C_GROUPS;      // (CDETR_GEMM_PRIO: any problem)
    if ((a_groups || c_groups) && d.precision != 1) return false;
    if (d.precision == 1 && !d.A16lo && !a_groups) return false;
    if (a_groups && (d.lda & 31)) return false;
    if (c_groups && (!d.C16lo || (reinterpret_cast<uintptr_t>(d.C16lo) & 15) || (d.ldc & 31) || (d.N & 31))) return false;
    const int KT = d.precision == 1 ? 32 : 64;
    if (d.K % KT != 0 || (d.lda & 7) != 0 || (d.ldb & 31) != 0 || (d.N & 3) != 0 || (d.ldc & 3) != 0) return false;
    if ((reinterpret_cast<uintptr_t>(d.A16) & 15) || (reinterpret_cast<uintptr_t>(d.A16lo) & 15) || (reinterpret_cast<uintptr_t>(d.B_split) & 15)) return false;
    if ((reinterpret_cast<uintptr_t>(d.C) & 15) || (reinterpret_cast<uintptr_t>(d.C16) & 7) || (reinterpret_cast<uintptr_t>(d.C16lo) & 7)) return false;
    if (d.resid && ((reinterpret_cast<uintptr_t>(d.resid) & 15) || (d.ldr & 3))) return false;
    if (d.resid && (d.flags & CDETR_GEMM_RESID_GROUPS) && ((d.ldr & 31) || (d.N & 31))) return false;
    if (d.gate && ((reinterpret_cast<uintptr_t>(d.gate) & 15) || (d.ldg & 3))) return false;
    if (d.gate16 && (reinterpret_cast<uintptr_t>(d.gate16) & 7)) return false;
    if ((d.flags & CDETR_GEMM_GATE16_ONLY) && !(d.gate && d.gate16)) return false;
    if (d.bias && (reinterpret_cast<uintptr_t>(d.bias) & 15)) return false;
    if (!d.C && !d.C16 && !c_groups) return false;
    if (d.C16lo && !d.C16 && !c_groups) return false;
    return true;
}

// Whether the halo-resident 3x3 form can run this (direct-to-LDS-eligible) problem on `tile`: stride-1 3x3 rows over a same-size map with
// pad == dil, the halo's pieces fit the seven batches per chunk that may carry them, everything fits the LDS.
bool cdetr_gemm_dl_halo_plan(const cdetr_gemm_desc& d, int tile, int& halo_R, int& nah, int& nhs) {
    const cdetr_conv_geom& g = d.g;
    if (g.mode != CDETR_ROWS_CONV_FWD && g.mode != CDETR_ROWS_CONV_DGRAD) return false;
    if (g.kh != 3 || g.kw != 3 || d.taps != 9 || g.stride != 1 || g.pad != g.dil || g.Ha != g.Hc || g.Wa != g.Wc) return false;
    if (d.flags & CDETR_GEMM_A_GROUPS) return false;
    const int BM = (tile == 0 || tile == 1) ? 128 : 64, BN = (tile == 0 || tile == 2) ? 128 : 64;
    halo_R = g.dil * (g.Wc + 1);
    nah = (BM + 2 * halo_R + 31) / 32;
    nhs = (nah + 6) / 7;
    if (nhs > 2) return false;
    return 2 * nah * 4096 + 3 * BN * 128 + 256 + 4096 <= 160 * 1024;
}

// tile: 0 = 128x128, 1 = 128x64 (rows x channels), 2 = 64x128, 3 = 64x64; stages 2..4; stages 13 = the halo-resident 3x3 form (3-deep weight ring);
// stages 200 + 10 * slices + ring depth (2 | 3) = the reduction cut into 2..8 slices per output tile (needs cdetr_gemm_desc.splitk_ws)
int cdetr_gemm_dl_launch(const cdetr_gemm_desc& d, int tile, int stages, hipStream_t st) {
    const bool x3 = d.precision == 1;
    if (stages == 13) {
        int R = 0, nah = 0, nhs = 0;
        if (!cdetr_gemm_dl_halo_plan(d, tile, R, nah, nhs)) {
            cdetr_set_error("cdetr_gemm (direct-to-LDS, halo): needs 3x3 stride-1 rows with pad == dil over a same-size map, halo <= 448 rows, <= 160 KB of LDS");
            return CDETR_ERR_UNSUPPORTED;
        }
#define DL_HALO(FM, FN)                                                                                                              \
    return x3 ? (nhs == 1 ? launch_dl_halo<FM, FN, 3, 1>(d, R, nah, st) : launch_dl_halo<FM, FN, 3, 2>(d, R, nah, st))               \
              : (nhs == 1 ? launch_dl_halo<FM, FN, 1, 1>(d, R, nah, st) : launch_dl_halo<FM, FN, 1, 2>(d, R, nah, st))
        switch (tile) {
            case 0: DL_HALO(2, 2);
            case 1: DL_HALO(2, 1);
            case 2: DL_HALO(1, 2);
            default: DL_HALO(1, 1);
        }
#undef DL_HALO
    }
    if (stages >= 200) {                                        // split reduction: stages = 200 + 10 * slices + ring depth (2 or 3)
        const int ks = (stages - 200) / 10, S = (stages - 200) % 10;
#define DL_SPLIT(FM, FN)                                                                                                                     \
    return S == 3 ? (x3 ? launch_dl<FM, FN, 3, 3, false, false, true>(d, st, ks) : launch_dl<FM, FN, 1, 3, false, false, true>(d, st, ks))    \
                  : (x3 ? launch_dl<FM, FN, 3, 2, false, false, true>(d, st, ks) : launch_dl<FM, FN, 1, 2, false, false, true>(d, st, ks))
        if (S == 2 || S == 3) {
            switch (tile) {
                case 0: DL_SPLIT(2, 2);
                case 1: DL_SPLIT(2, 1);
                case 2: DL_SPLIT(1, 2);
                case 3: DL_SPLIT(1, 1);
                default: break;
            }
        }
#undef DL_SPLIT
        cdetr_set_error("cdetr_gemm (direct-to-LDS, split reduction): no kernel for tile %d with stages code %d", tile, stages);
        return CDETR_ERR_UNSUPPORTED;
    }
    static const bool late = getenv("CDETR_DL_LATE_EPILOGUE") != nullptr;       // A/B: epilogue operands fetched after the k-loop (round 3)
    if (stages >= 100) {                                        // phase probe (tools/dl_probe.py): splitk_ws receives the time stamps
        if (!d.splitk_ws) { cdetr_set_error("cdetr_gemm_dl: the phase probe writes into splitk_ws"); return CDETR_ERR_ARG; }
#define DL_PROBE(FM, FN)                                                                                                  \
    return late ? (x3 ? launch_dl<FM, FN, 3, 3, true, false>(d, st) : launch_dl<FM, FN, 1, 3, true, false>(d, st))         \
                : (x3 ? launch_dl<FM, FN, 3, 3, true>(d, st) : launch_dl<FM, FN, 1, 3, true>(d, st))
        if (tile == 0 && stages == 103) { DL_PROBE(2, 2); }
        if (tile == 3 && stages == 103) { DL_PROBE(1, 1); }
        if (tile == 1 && stages == 103) { DL_PROBE(2, 1); }
#undef DL_PROBE
    }
#define DL_GO(FM, FN, S)                                                                                            \
    return late ? (x3 ? launch_dl<FM, FN, 3, S, false, false>(d, st) : launch_dl<FM, FN, 1, S, false, false>(d, st)) \
                : (x3 ? launch_dl<FM, FN, 3, S>(d, st) : launch_dl<FM, FN, 1, S>(d, st))
    switch (tile * 8 + stages) {
        case 0 * 8 + 2: DL_GO(2, 2, 2);
        case 0 * 8 + 3: DL_GO(2, 2, 3);
        case 1 * 8 + 2: DL_GO(2, 1, 2);
        case 1 * 8 + 3: DL_GO(2, 1, 3);
        case 1 * 8 + 4: DL_GO(2, 1, 4);
        case 2 * 8 + 2: DL_GO(1, 2, 2);
        case 2 * 8 + 3: DL_GO(1, 2, 3);
        case 2 * 8 + 4: DL_GO(1, 2, 4);
        case 3 * 8 + 2: DL_GO(1, 1, 2);
        case 3 * 8 + 3: DL_GO(1, 1, 3);
        case 3 * 8 + 4: DL_GO(1, 1, 4);
        default: break;
    }
#undef DL_GO
    cdetr_set_error("cdetr_gemm (direct-to-LDS): no kernel for tile %d with %d stages", tile, stages);
    return CDETR_ERR_UNSUPPORTED;
}
