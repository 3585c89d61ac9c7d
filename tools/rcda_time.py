"""HIP-event timing of the RCDA forward / backward at the encoder (L = H*W) and decoder (L = 300) shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops
N, H, W, nh, E = 2, 50, 50, 8, 256
dev = "cuda"


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for L in (H * W, 300):
    q_row, q_col = torch.randn(N, L, E, device=dev), torch.randn(N, L, E, device=dev)
    k_row, k_col = torch.randn(N, W, E, device=dev), torch.randn(N, H, E, device=dev)
    v = torch.randn(N, H, W, E, device=dev)
    dO = torch.randn(N, L, E, device=dev)
    o, a_row, a_col = ops.rcda_fwd_raw(q_row, q_col, k_row, k_col, v, None, None, nh)
    tf = timeit(lambda: ops.rcda_fwd_raw(q_row, q_col, k_row, k_col, v, None, None, nh))
    tb = timeit(lambda: ops.rcda_bwd_raw(dO, q_row, q_col, k_row, k_col, v, a_row, a_col, nh))
    print("L=%d  fwd %.1f us   bwd (dS + dV + dq/dk GEMMs) %.1f us" % (L, tf, tb))
