"""Does the ROW STRIDE of the operands bound the tile GEMM's k-loop?  Every workgroup of a launch walks k in step, so at one k-step all
workgroups of a column tile fetch the SAME 64 weight rows x 128 bytes; with a dense image the rows are K * taps * 2 bytes apart -- a multiple
of 512 bytes for every backbone layer -- and land on a quarter (or fewer) of the L2 channels.  Times the plain-bf16 direct-to-LDS kernel
(64x64 and 128x128 tiles) with the weight image / the activation twin at dense strides and at strides padded by 64 / 32 / 16 elements.
usage: python tools/stride_pad.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops, _ffi

DEV = "cuda"
SHAPES = [(5000, 256, 256, 9, (50, 50, 1, 1, 1)), (5000, 512, 512, 9, (50, 50, 1, 2, 2)), (5000, 256, 1024, 1, None), (5000, 512, 2048, 1, None),
          (5000, 1024, 256, 1, None), (5000, 2048, 512, 1, None), (20000, 128, 128, 9, (100, 100, 1, 1, 1)), (20000, 128, 512, 1, None)]


def bench(call, nsets):
    for i in range(nsets):
        call(i)
    torch.cuda.synchronize()
    reps = max(2 * nsets, 12)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for r in range(reps):
            call(r % nsets)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def strided(rows, cols, pad, dtype, fill):
    buf = torch.zeros(rows, cols + pad, device=DEV, dtype=dtype)
    v = buf[:, :cols]
    v.copy_(fill)
    return v


def run(shape, tile, padA, padB):
    M, N, K, taps, geo = shape
    g = None
    if geo is not None:
        H, W, stride, pad, dil = geo
        g = _ffi.ConvGeom(_ffi.ROWS_CONV_FWD, H, W, H, W, 3, 3, stride, pad, dil)
    torch.manual_seed(0)
    w = torch.randn(N, taps * K, device=DEV) / (K * taps) ** 0.5
    w16 = strided(N, taps * K, padB, torch.bfloat16, w.to(torch.bfloat16))
    wf = strided(N, taps * K, padB, torch.float32, w)
    nsets = max(2, min(16, int(1.5e9 // (4 * (M * K * 2 + M * N * 3)))))
    As = [torch.randn(M, K, device=DEV) for _ in range(nsets)]
    A16 = [strided(M, K, padA, torch.bfloat16, a.to(torch.bfloat16)) for a in As]
    Cs = [torch.empty(M, N, device=DEV) for _ in range(nsets)]
    C16 = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(nsets)]

    def call(i):
        ops.gemm_raw(As[i], K + padA, wf, taps * K + padB, Cs[i], N, M, N, K, taps=taps, relu=True, geom=g, B16=w16, precision=3, A16=A16[i], C16=C16[i], dl=tile)
    if padA:       # lda is shared by A (fp32, unused by this kernel) and its twin: give the descriptor a twin-strided fp32 view too
        Af = [strided(M, K, padA, torch.float32, a) for a in As]

        def call(i):   # noqa: F811
            ops.gemm_raw(Af[i], K + padA, wf, taps * K + padB, Cs[i], N, M, N, K, taps=taps, relu=True, geom=g, B16=w16, precision=3, A16=A16[i], C16=C16[i], dl=tile)
    t = bench(call, nsets)
    return t, Cs[0].clone()


if __name__ == "__main__":
    pads = [(0, 0), (0, 64), (0, 32), (64, 0), (64, 64), (32, 32), (96, 96)]
    print("plain bf16, us per launch; columns: (pad of the activation twin's row stride, pad of the weight image's row stride) in elements")
    print("%-30s %-8s " % ("M N K taps", "tile") + " ".join("%9s" % f"{a}/{b}" for a, b in pads))
    for sh in SHAPES:
        for tile in ((3, 3), (0, 3)):
            ref = None
            cells = []
            for pa, pb in pads:
                t, c = run(sh, tile, pa, pb)
                if ref is None:
                    ref = c
                else:
                    assert torch.equal(ref, c), "padding changed the result"
                cells.append(t)
            print("%-30s %-8s " % (str(sh[:4]), "64x64" if tile[0] == 3 else "128x128") + " ".join("%9.1f" % t for t in cells), flush=True)
