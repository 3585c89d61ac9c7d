"""Host-side cost of one captured step: how long the host spends ISSUING a replay (graph launches + the prefetch's stream operations)
against the device's step time -- pipelined (frozen-stage prefetch) and in-line.  usage: python tools/host_cost.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
tr = bench.build_trainer(dev, 300, "learned", "bf16x3")
images, rects, targets = bench.synthetic_batch(2, 800, 800, (37, 120), seed=0, device=dev)
tr.capture(images, rects, targets, warmup=1)
e = tr._entry
for name, fn in (("pipelined", lambda: tr.replay(pipelined=True)), ("in-line", tr.replay)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    per = []
    for _ in range(N):
        a = time.perf_counter()
        fn()
        per.append(time.perf_counter() - a)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    per.sort()
    print(f"{name}: host issue {t_issue / N * 1e3:.3f} ms/step (median call {per[N // 2] * 1e3:.3f}, max {per[-1] * 1e3:.3f}), wall {t_all / N * 1e3:.3f} ms/step")
# the pieces of a pipelined step, host time each (synchronised between pieces: launch cost only)
def t(fn, n=20):
    torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(n):
        fn()
    b = time.perf_counter() - a
    torch.cuda.synchronize()
    return b / n * 1e3
if e["layout"] == "chain":
    parts = [("F", e["F"]), ("Z", e["Z"]), ("B", e["B"]), ("W0", e["W0"])] + [(f"S{i + 1}", g) for i, g in enumerate(e["S"])] + \
            [(f"W{i + 1}", g) for i, g in enumerate(e["W"])] + [("O", e["O"])]
else:
    parts = [("g_f", e["g_f"]), ("g_p", e["g_p"]), ("g_a", e["g_a"])]
if e["fs"] is not None:
    parts.append(("frozen", e["fs"]["graph"]))
print("host ms per launch: " + "  ".join(f"{n} {t(g.replay):.3f}" for n, g in parts if g is not None))
