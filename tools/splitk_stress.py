"""Stress the split reductions for visibility races: many back-to-back launches of split problems (alternating shapes so that the scratch and its
counters are reused immediately), every result compared bit-for-bit with the first one of its shape and with the unsplit launch to 1e-5.
usage: python tools/splitk_stress.py [rounds]"""
import os, sys
os.environ.setdefault("CDETR_TUNING", "1")      # the per-call A/B knobs are only consulted when this is set at load time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = "cuda"
g = torch.Generator().manual_seed(0)
shapes = [(5000, 256, 1024), (5000, 256, 2304), (1237, 132, 512), (600, 256, 1024), (5000, 512, 4608)]
ops_ = []
for M, N, K in shapes:
    A = torch.randn(M, K, generator=g).to(dev)
    B = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    R = torch.randn(M, N, generator=g).to(dev)
    ops_.append((M, N, K, A, B, R))
ref, first, bad = {}, {}, 0
for prec in (1, 3):
    os.environ["CDETR_GEMM_SPLITK"] = "0"
    for i, (M, N, K, A, B, R) in enumerate(ops_):
        out = torch.empty(M, N, device=dev)
        ops.gemm_raw(A, K, B, K, out, N, M, N, K, resid=R, ldr=N, relu=True, precision=prec)
        ref[(prec, i)] = out.clone()
    os.environ.pop("CDETR_GEMM_SPLITK")
    for r in range(rounds):
        outs = []
        for i, (M, N, K, A, B, R) in enumerate(ops_):
            out = torch.empty(M, N, device=dev)
            ops.gemm_raw(A, K, B, K, out, N, M, N, K, resid=R, ldr=N, relu=True, precision=prec)
            outs.append(out)
        for i, out in enumerate(outs):
            k = (prec, i)
            if k not in first:
                first[k] = out.clone()
                e = ((out - ref[k]).abs().max() / ref[k].abs().max()).item()
                assert e < (1e-5 if prec == 1 else 1e-5), (k, e)
            elif not torch.equal(out, first[k]):
                bad += 1
                print("MISMATCH", k, r, ((out - first[k]).abs().max()).item(), flush=True)
print(f"{rounds} rounds x {len(shapes)} shapes x 2 precisions: {bad} mismatches; counters zero: {int(ops.splitk_ws()[:4096].abs().sum()) == 0}")
