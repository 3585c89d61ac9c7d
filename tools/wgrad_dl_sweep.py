"""Direct-to-LDS weight-gradient kernel (csrc/wgrad_dl.hip) vs the register-staged twin kernel (wgrad_tr16_kernel) on the backbone's
weight-gradient shapes at two 800x800 images, cold operands, every tile / pixels-per-tile / ring depth.
usage: CDETR_WGRAD_DL=0 python tools/wgrad_dl_sweep.py   (the env keeps cdetr_wgrad itself on the register-staged kernel)"""
import os
import sys
os.environ.setdefault("CDETR_WGRAD_DL", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops, _ffi

DEV = "cuda"
CFGS = [13, 14, 23, 22, 113, 123, 213, 223, 313, 314, 323, 324]
# (P pixels, Nout, Cin, taps, geometry)
SHAPES = [(20000, 128, 512, 1, None), (20000, 128, 128, 9, (100, 100, 1, 1, 1)), (20000, 512, 128, 1, None),
          (5000, 256, 1024, 1, None), (5000, 256, 256, 9, (50, 50, 1, 1, 1)), (5000, 1024, 256, 1, None),
          (5000, 512, 2048, 1, None), (5000, 512, 512, 9, (50, 50, 1, 2, 2)), (5000, 2048, 512, 1, None), (5000, 2048, 1024, 1, None),
          (5000, 256, 256, 1, None), (5000, 1024, 256, 1, None)]


def bench(call, nsets):
    for i in range(nsets):
        call(i)
    torch.cuda.synchronize()
    reps = max(2 * nsets, 12)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        call(r % nsets)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot_old = tot_best = 0.0
print("%-30s %12s | " % ("P Nout Cin taps", "reg-staged") + " ".join("%7d" % c for c in CFGS) + "   (us @ best workgroup target / 256)")
for (P, Nout, Cin, taps, geo) in SHAPES:
    g = None
    if geo is not None:
        H, W, stride, pad, dil = geo
        g = _ffi.ConvGeom(_ffi.ROWS_CONV_FWD, H, W, H, W, 3, 3, stride, pad, dil)
    nsets = max(2, min(12, int(1.2e9 // (2 * P * (Nout + Cin) * 3))))
    dys = [torch.randn(P, Nout, device=DEV) for _ in range(nsets)]
    xs = [torch.randn(P, Cin, device=DEV) for _ in range(nsets)]
    dy16 = [t.bfloat16() for t in dys]
    x16 = [t.bfloat16() for t in xs]
    dw = torch.zeros(Nout, taps * Cin, device=DEV)

    def call(i, dl=None):
        ops.wgrad_raw(dys[i], Nout, xs[i], Cin, dw, taps * Cin, P, Nout, Cin, taps=taps, geom=g, dY16=dy16[i], X16=x16[i], dl=dl, precision=3)
    fl = 2.0 * P * Nout * Cin * taps
    old = bench(lambda i: call(i), nsets)
    row, tgt = {}, {}
    for c in CFGS:
        if (c // 100 in (0, 1) and Nout < 128) or (c // 100 in (0, 2) and Cin < 128):
            continue
        cand = {t: bench(lambda i: call(i, (c, t)), nsets) for t in (256, 512, 768)}
        bt = min(cand, key=cand.get)
        row[c] = cand[bt]
        tgt[c] = bt
    best = min((v, k) for k, v in row.items())
    tot_old += old
    tot_best += min(old, best[0])
    print("%-30s %6.1f (%4.0f) | %s  best %d@%d x%.2f" % (str((P, Nout, Cin, taps)), old, fl / old / 1e6,
          " ".join(("%5.1f@%d" % (row[c], tgt[c] // 256)) if c in row else "%7s" % "-" for c in CFGS), best[1], tgt[best[1]], old / best[0]), flush=True)
print(f"sum: reg-staged {tot_old:.0f} us, best-of {tot_best:.0f} us")
