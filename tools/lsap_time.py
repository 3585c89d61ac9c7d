"""Time the device LSAP on the bench's matching problems (Q = 300, T = 37 and 120) and at Q = 900."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from counting_detr_amd import ops
for Q, Ts in [(300, [37, 120]), (300, [120] * 12), (900, [56, 900])]:
    rng = np.random.default_rng(0)
    plan = ops.MatchPlan(Ts, Q, "cuda")
    blocks = []
    for T in Ts:
        c = rng.standard_normal((Q, T)).astype(np.float32)
        blocks.append((c.T.copy() if T < Q else c).reshape(-1))
    cost = torch.from_numpy(np.concatenate(blocks)).cuda()
    for _ in range(3):
        ops.lsap(cost, plan)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.lsap(cost, plan)
    e1.record(); torch.cuda.synchronize()
    print(Q, Ts[:3], "%.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
