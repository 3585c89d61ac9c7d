"""Time the device LSAP on the cost matrices of bench.py's actual step (random-init model: what `value` contains) next to random costs of
the same shape; prints how many rows of each problem end in the one-step fast path (recomputed on the host from the matrix)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_trainer, synthetic_batch
from counting_detr_amd import ops
from counting_detr_amd.misc import NestedTensor

dev = torch.device("cuda", 0)
tr = build_trainer(dev, 300, "learned", "bf16x3")
images, rects, targets = synthetic_batch(2, 800, 800, (37, 120), seed=0, device=dev)
with torch.no_grad():
    out, _ = tr.model(NestedTensor(images, torch.zeros(2, 800, 800, dtype=torch.bool, device=dev)), rects=rects)
plan = ops.MatchPlan([37, 120], 300, dev)
tb = torch.cat([t["boxes"] for t in targets]).float()
cost = ops.match_cost(out["pred_logits"].float(), out["pred_boxes"].float(), tb, plan, 2.0, 5.0, 2.0)


def timeit(c):
    for _ in range(3):
        ops.lsap(c, plan)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.lsap(c, plan)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


def fast_rows(c, nr, nc):
    """rows whose argmin column (first scan step, v = 0 while only fast rows happened) is still free, simulated greedily"""
    m = c.reshape(nr, nc)
    taken, fast = set(), 0
    for r in range(nr):
        j = int(np.argmin(m[r]))
        if j not in taken:
            fast += 1
        taken.add(j)
    return fast


print("bench step's cost matrices: %.1f us" % timeit(cost))
ch = cost.cpu().numpy()
off = plan.cost_off_host
for b, T in enumerate((37, 120)):
    blk = ch[off[b]:off[b] + 300 * T]
    print(f"  image {b}: T = {T}: ~{fast_rows(blk, T, 300)} of {T} rows have a free nearest column (greedy estimate); cost range {blk.min():.3f} .. {blk.max():.3f}, "
          f"row-wise spread (median of max - min): {np.median(blk.reshape(T, 300).max(1) - blk.reshape(T, 300).min(1)):.4f}")
rnd = torch.randn_like(cost)
print("random costs, same shapes: %.1f us" % timeit(rnd))
