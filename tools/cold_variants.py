"""Cold-operand timing (see tools/cold_gemm.py) of the tile variants on the hot shapes of the step: CDETR_GEMM_VARIANT is read per call."""
import os, sys
os.environ.setdefault("CDETR_TUNING", "1")      # the per-call A/B knobs are only consulted when this is set at load time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops
ops.PRECISION = 1
dev = "cuda"
SH = [(5000, 1024, 256, True), (5000, 256, 1024, True), (5000, 256, 256, True), (20000, 512, 128, True), (20000, 128, 512, False),
      (5000, 2048, 512, True), (5000, 512, 2048, False), (80000, 256, 64, True), (80000, 64, 256, False), (20000, 512, 256, True), (5000, 512, 1024, False)]
VARS = {0: "auto", 4: "64x64", 9: "64x128w8", 10: "128x128w16", 13: "96x128w12"}
print("%-26s" % "M,N,K,resid" + "".join("%14s" % v for v in VARS.values()))
for (M, N, K, res) in SH:
    W = torch.randn(N, K, device=dev) / K ** 0.5
    mirror = ops.WeightMirror([], [(W, None)]); mirror.refresh()
    sp = mirror.lookup_fwd(W)
    b = torch.randn(N, device=dev)
    ns = max(2, min(24, int(3e9 // (4 * (M * K + 2 * M * N)))))
    As = [torch.randn(M, K, device=dev) for _ in range(ns)]
    Rs = [torch.randn(M, N, device=dev) for _ in range(ns)]
    Cs = [torch.empty(M, N, device=dev) for _ in range(ns)]
    row = "%-26s" % ("%d,%d,%d,%s" % (M, N, K, "r" if res else "-"))
    for v in VARS:
        os.environ["CDETR_GEMM_VARIANT"] = str(v)
        def call(i):
            ops.gemm_raw(As[i], K, W, K, Cs[i], N, M, N, K, bias=b, relu=res, resid=Rs[i] if res else None, ldr=N, B_split=sp)
        for i in range(ns):
            call(i)
        torch.cuda.synchronize()
        reps = 3 * ns
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps):
            call(r % ns)
        e1.record(); torch.cuda.synchronize()
        row += "%12.1fus" % (e0.elapsed_time(e1) / reps * 1e3)
    print(row, flush=True)
    del As, Rs, Cs
