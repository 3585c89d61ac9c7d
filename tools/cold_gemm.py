"""GEMM timing with COLD operands: cycles through enough (A, resid, C) buffer sets that nothing survives in L2 / Infinity Cache,
like the activations of a training step (the hot-loop numbers of tools/gemm_sweep.py keep a shape's operands cache resident).
usage: python tools/cold_gemm.py [nsets]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops
ops.PRECISION = 1
dev = "cuda"
SH = [(20000, 512, 128), (5000, 1024, 256), (5000, 256, 1024), (80000, 256, 64), (80000, 64, 256), (5000, 256, 256), (5000, 2048, 512), (20000, 128, 512)]
nsets = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for (M, N, K) in SH:
    W = torch.randn(N, K, device=dev) / K ** 0.5
    mirror = ops.WeightMirror([], [(W, None)]) if K % 32 == 0 else None
    if mirror is not None:
        mirror.refresh()
    sp = mirror.lookup_fwd(W) if mirror is not None else None
    b = torch.randn(N, device=dev)
    for mode in ("hot", "cold"):
        ns = 1 if mode == "hot" else max(2, min(nsets, int(3e9 // (4 * (M * K + 2 * M * N)))))
        As = [torch.randn(M, K, device=dev) for _ in range(ns)]
        Rs = [torch.randn(M, N, device=dev) for _ in range(ns)]
        Cs = [torch.empty(M, N, device=dev) for _ in range(ns)]
        for variant, kw in (("plain", {}), ("bias+resid+relu", {"resid": True})):
            def call(i):
                ops.gemm_raw(As[i], K, W, K, Cs[i], N, M, N, K, bias=b, relu=bool(kw), resid=Rs[i] if kw else None, ldr=N, B_split=sp)
            for i in range(ns):
                call(i)
            torch.cuda.synchronize()
            reps = 3 * ns if mode == "cold" else 30
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(reps):
                call(r % ns)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            byts = 4.0 * (M * K + N * K + M * N * (2 if kw else 1))
            print("%-18s %-5s %-16s %7.1f us %6.1f TF  %5.2f TB/s (compulsory bytes)" % ((M, N, K), mode, variant, us, 2.0 * M * N * K / us / 1e6, byts / us / 1e6), flush=True)
        del As, Rs, Cs
