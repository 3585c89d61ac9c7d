"""Error-vs-speed of the backward arithmetic (ops.PRECISION_BWD): gradient of the bench step under fp32 MFMA (reference), bf16x3,
bf16x2 and plain bf16 backward (forward always bf16x3 except in the fp32 run), and the graph-replay step time of each."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import ops
from bench import build_trainer, synthetic_batch
dev = torch.device("cuda")
images, rects, targets = synthetic_batch(2, 800, 800, (37, 120), seed=0, device=dev)
from counting_detr_amd.misc import nested_tensor_from_tensor_list
im, mk = nested_tensor_from_tensor_list(images).decompose()
nb = float(sum(len(t["boxes"]) for t in targets))
res = {}
for tag, prec, bwd in (("fp32", "fp32", 1), ("bf16x3", "bf16x3", 1), ("bf16x2", "bf16x3", 2), ("bf16x1", "bf16x3", 3)):
    tr = build_trainer(dev, 300, "learned", prec)
    ops.PRECISION_BWD = bwd
    tr.arith = (tr.arith[0], bwd)                 # the trainer owns its arithmetic (ops.arithmetic)
    with ops.arithmetic(*tr.arith):
        out = tr._fwd_bwd(im, mk, rects, targets, nb)
    torch.cuda.synchronize()
    g = tr.flat_g.detach().double().clone()
    tr.capture(images, rects, targets, warmup=1)
    for _ in range(3):
        tr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    res[tag] = (g, ms, float(out["loss"]), list(tr.seg_bounds), tr.names, dict(tr.offsets))
    del tr
    torch.cuda.empty_cache()
ref = res["fp32"][0]
sb = res["fp32"][3]
print("tag       ms/step   loss        |g| rel.err(l2)  max/|g|max   per-segment l2 rel err [transformer+proj, layer4, layer3, layer2]   worst per-parameter norm rel err")
for tag in ("fp32", "bf16x3", "bf16x2", "bf16x1"):
    g, ms, loss, _, names, offs = res[tag]
    e = (g - ref)
    segs = [float(e[sb[i]:sb[i + 1]].norm() / ref[sb[i]:sb[i + 1]].norm()) for i in range(4)]
    worst = 0.0; wn = ""
    for n in names:
        o, sz = offs[n]
        a, b = float(g[o:o + sz].norm()), float(ref[o:o + sz].norm())
        if b > 1e-12 * float(ref.norm()):
            r = abs(a - b) / b
            if r > worst:
                worst, wn = r, n
    print(f"{tag:8s} {ms:8.3f}  {loss:.6f}  {float(e.norm() / ref.norm()):.3e}      {float(e.abs().max() / ref.abs().max()):.3e}   "
          + " ".join(f"{s:.2e}" for s in segs) + f"   {worst:.2e} ({wn})")
