"""Run the RCDA core fwd+bwd at the encoder (L=H*W) and decoder (L=300) shapes; use under rocprofv3 --stats for kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops
N, H, W, nh, E = 2, 50, 50, 8, 256
dev = "cuda"
for L in (H * W, 300):
    q_row, q_col = torch.randn(N, L, E, device=dev), torch.randn(N, L, E, device=dev)
    k_row, k_col = torch.randn(N, W, E, device=dev), torch.randn(N, H, E, device=dev)
    v = torch.randn(N, H, W, E, device=dev)
    dO = torch.randn(N, L, E, device=dev)
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        o, a_row, a_col = ops.rcda_fwd_raw(q_row, q_col, k_row, k_col, v, None, None, nh)
        ops.rcda_bwd_raw(dO, q_row, q_col, k_row, k_col, v, a_row, a_col, nh)
torch.cuda.synchronize()
