// Row enumeration of the implicit-GEMM kernels (igemm.hip, igemm_dl.hip): a GEMM row m is an output pixel (conv forward) or an input
// pixel (conv data-gradient); (row, filter tap) -> row of the gathered NHWC tensor, or -1 where the tap falls into the padding.
#pragma once
#include "../../include/cdetr_hip.h"
#include <hip/hip_runtime.h>

namespace {

// two-level batch index: z = outer * batch_inner + inner -> inner * s + outer * s2 (batch_inner == 0: one level, z * s)
__host__ __device__ __forceinline__ long batch_off(int z, int inner, long s, long s2) {
    if (inner <= 0) return (long)z * s;
    const int zo = z / inner;
    return (long)(z - zo * inner) * s + (long)zo * s2;
}

struct RowCoord {
    int ybase, xbase, nbase, m;
    bool valid;
};

__device__ __forceinline__ RowCoord decode_row(const cdetr_conv_geom& g, int m, int M) {
    RowCoord r;
    r.m = m;
    r.valid = m < M;
    r.ybase = r.xbase = r.nbase = 0;
    if (g.mode != CDETR_ROWS_DENSE && r.valid) {
        const int hw = g.Hc * g.Wc;
        const int n = m / hw;
        const int rem = m - n * hw;
        const int y = rem / g.Wc;
        const int x = rem - y * g.Wc;
        r.nbase = n * g.Ha * g.Wa;
        if (g.mode == CDETR_ROWS_CONV_FWD) {
            r.ybase = y * g.stride - g.pad;
            r.xbase = x * g.stride - g.pad;
        } else {
            r.ybase = y + g.pad;
            r.xbase = x + g.pad;
        }
    }
    return r;
}

// row index into the gathered tensor for (row coord, tap); -1 when the tap falls outside (zero contribution)
__device__ __forceinline__ long gather_row(const cdetr_conv_geom& g, const RowCoord& r, int tap) {
    if (!r.valid) return -1;
    if (g.mode == CDETR_ROWS_DENSE) return r.m;
    const int ky = tap / g.kw;
    const int kx = tap - ky * g.kw;
    if (g.mode == CDETR_ROWS_CONV_FWD) {
        const int iy = r.ybase + ky * g.dil;
        const int ix = r.xbase + kx * g.dil;
        if (iy < 0 || iy >= g.Ha || ix < 0 || ix >= g.Wa) return -1;
        return (long)r.nbase + (long)iy * g.Wa + ix;
    }
    int ty = r.ybase - ky * g.dil;
    int tx = r.xbase - kx * g.dil;
    if (ty < 0 || tx < 0) return -1;
    if (g.stride > 1) {
        if ((ty % g.stride) != 0 || (tx % g.stride) != 0) return -1;
        ty /= g.stride;
        tx /= g.stride;
    }
    if (ty >= g.Ha || tx >= g.Wa) return -1;
    return (long)r.nbase + (long)ty * g.Wa + tx;
}

}  // namespace
