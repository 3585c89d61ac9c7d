"""Time the tile GEMMs of the B=2 800x800 step with the reduction cut into 1-4 slices (CDETR_GEMM_SPLITK), forward (bf16x3) and
backward (plain bf16, bf16 twin of A) arithmetic.  usage: python tools/splitk_sweep.py"""
import os, sys
os.environ.setdefault("CDETR_TUNING", "1")      # the per-call A/B knobs are only consulted when this is set at load time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops, _ffi


def run(M, N, K, taps, geo, precision, twin, reps=30):
    dev = "cuda"
    A = torch.randn(M, K, device=dev)
    A16 = A.to(torch.bfloat16) if twin else None
    g = None
    if geo is not None:
        H, W, stride, pad, dil = geo
        g = _ffi.ConvGeom(_ffi.ROWS_CONV_FWD, H, W, H, W, 3, 3, stride, pad, dil)
    B = torch.randn(N, taps * K, device=dev)
    C = torch.empty(M, N, device=dev)
    resid = torch.randn(M, N, device=dev)
    call = lambda: ops.gemm_raw(A, K, B, taps * K, C, N, M, N, K, taps=taps, b_layout=0, geom=g, precision=precision, A16=A16,  # noqa: E731
                                resid=resid, ldr=N, relu=True)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if __name__ == "__main__":
    shapes = [(5000, 256, 1024, 1, None), (5000, 256, 256, 9, (50, 50, 1, 1, 1)), (5000, 256, 256, 1, None), (5000, 256, 2048, 1, None),
              (5000, 512, 512, 9, (50, 50, 1, 2, 2)), (5000, 512, 2048, 1, None), (5000, 512, 1024, 1, None), (5000, 1024, 256, 1, None),
              (5000, 2048, 512, 1, None), (5000, 2048, 1024, 1, None), (20000, 128, 128, 9, (100, 100, 1, 1, 1)), (20000, 128, 512, 1, None),
              (20000, 512, 128, 1, None), (80000, 64, 64, 9, (200, 200, 1, 1, 1)), (80000, 64, 256, 1, None), (600, 256, 1024, 1, None)]
    for prec, twin in ((1, False), (3, True)):
        print(f"precision {prec} ({'bf16x3' if prec == 1 else 'bf16 + twin'}): us per launch with 1 (off) / auto / 2 / 3 / 4 slices | two wave groups "
              "per 64x64 tile: unsplit / 2 / 3 slices")
        for M, N, K, taps, geo in shapes:
            row = f"  M={M:6d} N={N:5d} K={K:5d} taps={taps}: "
            combos = (("0", "0"), ("0", None), ("0", "2"), ("0", "3"), ("0", "4"), ("14", "0"), ("14", "2"), ("14", "3"))
            if os.environ.get("SWEEP_BIG"):      # larger per-wave tiles: 15 = 128x128 / 4 waves of 64x64, 16 = 128x128 / 8 waves of 64x32, 17 = 64x128 / 4 waves
                combos = (("0", None), ("15", "0"), ("15", None), ("15", "2"), ("16", "0"), ("16", None), ("17", "0"), ("17", None))
            for v, s in combos:
                os.environ["CDETR_GEMM_VARIANT"] = v
                if s is None:
                    os.environ.pop("CDETR_GEMM_SPLITK", None)
                else:
                    os.environ["CDETR_GEMM_SPLITK"] = s
                row += ("   |" if s == "0" and v != "0" else "") + f"{run(M, N, K, taps, geo, prec, twin):8.1f}"
            print(row, flush=True)
