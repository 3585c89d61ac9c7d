"""Dispatch delay in front of the Hungarian solve inside a captured graph: [three tiny kernels] -> cdetr_lsap -> [tiny kernel], replayed; run under
rocprofv3 --kernel-trace and read the start-to-start gaps with tools/step_gaps.py-style arithmetic (printed here from HIP events as a cross-check).
CDETR_LSAP_COST_LDS=0: the cost matrix stays in L2 (the kernel asks for ~5 KB of LDS instead of ~150 KB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops
dev = "cuda"
Q, Ts = 300, (37, 120)
plan = ops.MatchPlan.capacity(2, Q, 128, dev)
plan.set_counts(list(Ts)) if hasattr(plan, "set_counts") else None
g = torch.Generator(device=dev).manual_seed(3)
cost = torch.rand(plan.cost_numel, device=dev, generator=g)
a = torch.zeros(1024, device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        a.add_(1.0); ops.lsap(cost, plan); a.add_(1.0)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=s):
    a.add_(1.0); a.add_(1.0); a.add_(1.0)
    ops.lsap(cost, plan)
    a.add_(1.0)
for _ in range(30):
    gr.replay()
torch.cuda.synchronize()
print("done")
