"""Time every igemm variant on the GEMM / conv shapes of the 800x800 B=2 step (true GPU time: N back-to-back launches
between two HIP events).  usage: python tools/gemm_sweep.py [shapes.csv]"""
import os, sys
os.environ.setdefault("CDETR_TUNING", "1")      # the per-call A/B knobs are only consulted when this is set at load time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops, _ffi

SHAPES = [  # (M, N, K, taps, layout, conv geometry or None)
    (600, 256, 256, 1, 0, None), (600, 256, 256, 1, 1, None), (600, 1024, 256, 1, 0, None), (600, 256, 1024, 1, 0, None),
    (5000, 256, 256, 1, 0, None), (5000, 256, 256, 1, 1, None), (5000, 1024, 256, 1, 0, None), (5000, 256, 1024, 1, 0, None),
    (5000, 256, 1024, 1, 1, None), (5000, 1024, 256, 1, 1, None), (5000, 256, 4096, 1, 0, None),
    (5000, 512, 1024, 1, 0, None), (5000, 2048, 512, 1, 0, None), (5000, 512, 2048, 1, 0, None), (5000, 512, 2048, 1, 1, None),
    (5000, 2048, 512, 1, 1, None), (5000, 1024, 256, 1, 0, None), (5000, 256, 1024, 1, 0, None),
    (5000, 512, 512, 9, 0, (50, 50, 1, 2, 2)), (5000, 512, 512, 9, 1, (50, 50, 1, 2, 2)),
    (5000, 256, 256, 9, 0, (50, 50, 1, 1, 1)), (5000, 256, 256, 9, 1, (50, 50, 1, 1, 1)),
    (20000, 128, 128, 9, 0, (100, 100, 1, 1, 1)), (20000, 128, 128, 9, 1, (100, 100, 1, 1, 1)),
    (20000, 512, 128, 1, 0, None), (20000, 128, 512, 1, 0, None), (20000, 512, 128, 1, 1, None),
    (80000, 64, 64, 9, 0, (200, 200, 1, 1, 1)), (80000, 256, 64, 1, 0, None), (80000, 64, 256, 1, 0, None),
]
NAMES = {0: "auto", 1: "f128x128", 2: "f128x64", 3: "f64x64k64", 4: "f64x64k32", 5: "direct", 6: "generic", 7: "f32x64k32", 8: "f128x64w8", 9: "f64x128w8", 10: "f128x128w16", 11: "f128x128w16k64", 12: "f64x128w8k64"}


def run(shape, variant, reps=20):
    M, N, K, taps, bl, geo = shape
    dev = "cuda"
    if geo is None:
        A = torch.randn(M, K, device=dev)
        g = None
    else:
        H, W, stride, pad, dil = geo
        nimg = M // (H * W)
        A = torch.randn(nimg * H * W, K if bl == 0 else K, device=dev)
        mode = _ffi.ROWS_CONV_FWD if bl == 0 else _ffi.ROWS_CONV_DGRAD
        g = _ffi.ConvGeom(mode, H, W, H, W, 3, 3, stride, pad, dil)
    B = torch.randn(N, taps * K, device=dev) if bl == 0 else torch.randn(K, taps * N, device=dev)
    C = torch.empty(M, N, device=dev)
    os.environ["CDETR_GEMM_VARIANT"] = str(variant)
    ldb = taps * K if bl == 0 else N
    call = lambda: ops.gemm_raw(A, K, B, ldb, C, N, M, N, K, taps=taps, b_layout=bl, geom=g)  # noqa: E731
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    return us, 2.0 * M * N * K * taps / us / 1e6


WSHAPES = [  # (P, Nout, Cin, taps, geometry)
    (600, 256, 256, 1, None), (5000, 256, 256, 1, None), (5000, 1024, 256, 1, None), (5000, 256, 1024, 1, None),
    (5000, 256, 4096, 1, None), (5000, 2048, 512, 1, None), (5000, 512, 2048, 1, None), (5000, 512, 1024, 1, None),
    (5000, 512, 512, 9, (50, 50, 1, 2, 2)), (5000, 256, 256, 9, (50, 50, 1, 1, 1)), (20000, 128, 128, 9, (100, 100, 1, 1, 1)),
    (20000, 512, 128, 1, None), (20000, 128, 512, 1, None), (20000, 512, 256, 1, None),
]
WNAMES = {0: "auto", 1: "w128x128", 2: "w128x64", 3: "w64x128", 4: "w64x64"}


def runw(shape, variant, reps=20):
    P, Nout, Cin, taps, geo = shape
    dev = "cuda"
    dY = torch.randn(P, Nout, device=dev)
    X = torch.randn(P, Cin, device=dev)
    dW = torch.zeros(Nout, taps * Cin, device=dev)
    g = None
    if geo is not None:
        H, W, stride, pad, dil = geo
        g = _ffi.ConvGeom(_ffi.ROWS_CONV_FWD, H, W, H, W, 3, 3, stride, pad, dil)
    os.environ["CDETR_WGRAD_VARIANT"] = str(variant)
    call = lambda: ops.wgrad_raw(dY, Nout, X, Cin, dW, taps * Cin, P, Nout, Cin, taps=taps, geom=g)  # noqa: E731
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    return us, 2.0 * P * Nout * Cin * taps / us / 1e6


if __name__ == "__main__":
    if "wgrad" in sys.argv or "all" in sys.argv:
        print("%-30s" % "P,Nout,Cin,taps" + "".join("%22s" % WNAMES[v] for v in range(5)))
        for sh in WSHAPES:
            row = "%-30s" % (",".join(str(x) for x in sh[:4]))
            for v in range(5):
                us, tf = runw(sh, v)
                row += "%12.1fus %5.1fTF" % (us, tf)
            print(row, flush=True)
        if "all" not in sys.argv:
            sys.exit(0)
    VS = (0, 1, 2, 3, 4, 7, 5)
    print("%-34s" % "M,N,K,taps,layout" + "".join("%22s" % NAMES[v] for v in VS))
    for sh in SHAPES:
        row = "%-34s" % (",".join(str(x) for x in sh[:5]))
        for v in VS:
            if v == 5 and (sh[5] is not None or sh[0] > 6000):
                row += "%22s" % "-"
                continue
            us, tf = run(sh, v)
            row += "%12.1fus %5.1fTF" % (us, tf)
        print(row, flush=True)
