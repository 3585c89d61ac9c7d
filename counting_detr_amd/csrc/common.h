// Shared device/host helpers for the CDNA4 (gfx950) kernels of libcdetr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CDETR_OK 0
#define CDETR_ERR_ARG (-1)
#define CDETR_ERR_LAUNCH (-2)
#define CDETR_ERR_UNSUPPORTED (-3)

// A/B and test knobs that must be re-read on EVERY call (tests switch kernel variants inside one process) cost a getenv per launch;
// the product path does not pay it: they are consulted only when CDETR_TUNING was set when the library was loaded (tests/conftest.py,
// the sweep tools).  Knobs that select a fixed configuration for a whole process are plain statics read once.
#include <stdlib.h>
static inline const char* cdetr_tune_env(const char* name) {
    static const bool tuning = getenv("CDETR_TUNING") != nullptr;
    return tuning ? getenv(name) : nullptr;
}

// thread-local last-error string (cdetr_last_error)
void cdetr_set_error(const char* fmt, ...);

#define CDETR_CHECK_ARG(cond, ...)          \
    do {                                    \
        if (!(cond)) {                      \
            cdetr_set_error(__VA_ARGS__);   \
            return CDETR_ERR_ARG;           \
        }                                   \
    } while (0)

static inline int cdetr_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        cdetr_set_error("%s: %s", what, hipGetErrorString(e));
        return CDETR_ERR_LAUNCH;
    }
    return CDETR_OK;
}

constexpr int SPLITK_COUNTERS = 4096;      // int32 arrival counters at the head of cdetr_gemm_desc.splitk_ws (one per output tile; igemm.hip, igemm_dl.hip)

// hipcc (ROCm 7.2) does not pad the MFMA -> accumulator-read hazard across a loop-exit edge: the register copies
// (v_accvgpr_read) that follow a k-loop can issue before the last 16-pass MFMA has written its final rows (seen on
// wgrad_fast_kernel<64,64>: accumulator row r = 15 stale).  Scheduling barriers / inline nops do not help (the copies are
// placed by the register allocator ahead of them).  Fix: issue one more MFMA with zero operands on every accumulator in
// the epilogue's own basic block -- an exact +0 -- so the in-block hazard recognizer pads the following reads correctly.
__device__ __forceinline__ void mfma_drain(f32x16& acc) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(0.f, 0.f, acc, 0, 0, 0); }
__device__ __forceinline__ void mfma_drain(f32x4& acc) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(0.f, 0.f, acc, 0, 0, 0); }
template <int FM, int FN>
__device__ __forceinline__ void mfma_drain(f32x16 (&acc)[FM][FN]) {
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b) mfma_drain(acc[a][b]);
}
template <int NF>
__device__ __forceinline__ void mfma_drain(f32x16 (&acc)[NF]) {
#pragma unroll
    for (int a = 0; a < NF; ++a) mfma_drain(acc[a]);
}

// split-bf16 ("bf16x3") operand preparation: x = hi + lo + O(2^-18 |x|) with hi = bf16(x), lo = bf16(x - hi); a product
// a*b is then evaluated as hi*hi + hi*lo + lo*hi on the bf16 matrix pipe (fp32 accumulate), relative error <= ~1e-5.
__device__ __forceinline__ void split_bf16x8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        hi[i] = h;
        lo[i] = (__bf16)(x[i] - (float)h);
    }
}
// Packed form used when a tile is split ONCE while it is staged into LDS: two consecutive-k values -> one dword of hi
// and one of lo (v_cvt_pk_bf16_f32 + shift/and + v_pk_add_f32 + v_cvt_pk_bf16_f32 = 2.5 VALU ops per element).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_bf16_pk(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 v = {x0, x1};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const f32x2 hf = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
    const f32x2 r = v - hf;
    const bf16x2 l = __builtin_convertvector(r, bf16x2);
    lo = __builtin_bit_cast(unsigned, l);
}
// four consecutive-k fp32 values -> hi (8 bytes) at dst, lo (8 bytes) at dst + lo_off (both in bf16 elements)
// Sum / maximum of a value over the two halves of a wave (lane l and lane l ^ 32), the same result in both lanes.  gfx950's
// v_permlane32_swap exchanges the upper half of one register with the lower half of another in ONE VALU instruction; __shfl_xor(x, 32)
// is a ds_bpermute -- an LDS-pipe round trip of ~100 cycles, which sat once per key column in the RCDA dS loop and once or twice per key
// tile in the flash-attention loops (round 6).  a + b is commutative, so the results are bit-identical to the shuffle forms.
__device__ __forceinline__ float xhalf_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);     // r[0] = [lo, lo], r[1] = [hi, hi]
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xhalf_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ void stash_split4(__bf16* dst, int lo_off, float x0, float x1, float x2, float x3) {
    uint2 h, l;
    split_bf16_pk(x0, x1, h.x, l.x);
    split_bf16_pk(x2, x3, h.y, l.y);
    *reinterpret_cast<uint2*>(dst) = h;
    *reinterpret_cast<uint2*>(dst + lo_off) = l;
}
// four consecutive-k fp32 values -> their bf16 roundings (8 bytes) at dst: the plain-bf16 products never read a lo plane
__device__ __forceinline__ void stash_hi4(__bf16* dst, float x0, float x1, float x2, float x3) {
    const f32x2 v0 = {x0, x1}, v1 = {x2, x3};
    uint2 h;
    h.x = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, bf16x2));
    h.y = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, bf16x2));
    *reinterpret_cast<uint2*>(dst) = h;
}
__device__ __forceinline__ f32x16 mfma_bf16x3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);    // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    return acc;
}

// TERMS 3 = the split product above; 1 = plain bf16 (hi * hi only: the backward's arithmetic, ops.PRECISION_BWD 3)
template <int TERMS>
__device__ __forceinline__ f32x16 mfma_bf16_terms(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x16 acc) {
    if constexpr (TERMS == 1) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    else return mfma_bf16x3(ah, al, bh, bl, acc);
}

// Wave-wide reductions on DPP lane permutes (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror inside the rows of 16, row_bcast 15 /
// 31 across them; lane 63 ends up with the total, v_readlane broadcasts it): ~6 VALU-latency steps instead of 6 ds_bpermute round
// trips through the LDS pipe (a LayerNorm row does two reductions: they were most of its time).  The result is wave-uniform.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_pull(float x, float fill) {     // lanes the permute does not write read `fill`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_pull<0xB1, 0xf>(v, 0.f);      // quad_perm [1,0,3,2]
    v += dpp_pull<0x4E, 0xf>(v, 0.f);      // quad_perm [2,3,0,1]
    v += dpp_pull<0x141, 0xf>(v, 0.f);     // row_half_mirror
    v += dpp_pull<0x140, 0xf>(v, 0.f);     // row_mirror: every lane of a row holds the row's sum
    v += dpp_pull<0x142, 0xa>(v, 0.f);     // row_bcast:15 into rows 1 and 3
    v += dpp_pull<0x143, 0xc>(v, 0.f);     // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_pull<0xB1, 0xf>(v, -INFINITY));
    v = fmaxf(v, dpp_pull<0x4E, 0xf>(v, -INFINITY));
    v = fmaxf(v, dpp_pull<0x141, 0xf>(v, -INFINITY));
    v = fmaxf(v, dpp_pull<0x140, 0xf>(v, -INFINITY));
    v = fmaxf(v, dpp_pull<0x142, 0xa>(v, -INFINITY));
    v = fmaxf(v, dpp_pull<0x143, 0xc>(v, -INFINITY));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
