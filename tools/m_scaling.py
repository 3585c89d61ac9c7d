"""How does the tile kernel's time scale with the row count M at fixed (N, K)?  If 4x the rows costs well under 4x the time the
shape is parallelism / latency bound at B=2 (few workgroups per CU) and splitting the reduction across workgroups would pay.
usage: python tools/m_scaling.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops, _ffi


def run(M, N, K, taps, geo, precision, reps=30):
    dev = "cuda"
    A = torch.randn(M, K, device=dev)
    g = None
    if geo is not None:
        H, W, stride, pad, dil = geo
        g = _ffi.ConvGeom(_ffi.ROWS_CONV_FWD, H, W, H, W, 3, 3, stride, pad, dil)
    B = torch.randn(N, taps * K, device=dev)
    C = torch.empty(M, N, device=dev)
    call = lambda: ops.gemm_raw(A, K, B, taps * K, C, N, M, N, K, taps=taps, b_layout=0, geom=g, precision=precision)  # noqa: E731
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
    shapes = [(256, 1024, 1, None), (1024, 256, 1, None), (256, 256, 9, (50, 50, 1, 1, 1)), (512, 512, 9, (50, 50, 1, 2, 2)),
              (512, 2048, 1, None), (2048, 512, 1, None), (256, 256, 1, None)]
    for prec in (1, 3):
        print(f"precision {prec} ({'bf16x3' if prec == 1 else 'bf16'}): us per launch (TF algorithmic) at M = 5000 x {{1, 2, 4, 8}}")
        for N, K, taps, geo in shapes:
            row = f"  N={N:5d} K={K:5d} taps={taps}: "
            for mult in (1, 2, 4, 8):
                M = 5000 * mult
                us = run(M, N, K, taps, geo, prec)
                row += f"{us:8.1f} us ({2.0 * M * N * K * taps / us / 1e6:5.0f} TF)"
            print(row, flush=True)
