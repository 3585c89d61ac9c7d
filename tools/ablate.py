"""What is each piece of the step worth on the critical path?  The captured, pipelined step of bench.py (B=2 800x800 Q=300 T=(37,120)) with
one piece REMOVED at a time (results are wrong on purpose -- this prices optimisation candidates before anybody writes a kernel):
  wgrad      no weight-gradient launch at all (backbone + transformer)            -> ceiling of any weight-gradient work
  wgrad_bb   no weight-gradient launch of the backbone's tile class only
  images     the forward weight images are not rewritten at the head of the step   -> ceiling of folding them into AdamW (item 6)
  mirror     the data-gradient weight images are not rewritten
  lsap       the assignment solve returns at once (identity assignment)
  adamw      no optimizer pass
  wgrad_dec / wgrad_enc / wgrad_tr   no parameter gradient of the decoder stack / the encoder stack / both (they run beside layer4's data gradients)
usage: python tools/ablate.py [variant ...]      (default: all; each in its own trainer, same process)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from counting_detr_amd import ops, engine

dev = torch.device("cuda", 0)
VARIANTS = sys.argv[1:] or ["none", "wgrad", "wgrad_bb", "images", "mirror", "lsap", "adamw", "none"]


def timed(tr, steps=20):
    rp = (lambda: tr.replay(pipelined=True)) if tr._entry.get("fs") is not None else tr.replay
    for _ in range(4):
        rp()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rp()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def run(variant):
    saved = {}

    def patch(obj, name, fn):
        saved[(obj, name)] = getattr(obj, name)
        setattr(obj, name, fn)
    if variant == "wgrad":
        patch(ops, "wgrad_raw", lambda *a, **k: None)
    elif variant == "wgrad_bb":
        orig = ops.conv_wgrad_
        patch(ops, "conv_wgrad_", lambda *a, **k: None)
    elif variant in ("wgrad_dec", "wgrad_enc", "wgrad_tr"):      # the decoder's / the encoder's / both stacks' parameter gradients only
        def without_wg(fn):
            def run_(ctx, *a, **k):
                w = ops._wg
                ops._wg = lambda *aa, **kk: None
                try:
                    return fn(ctx, *a, **k)
                finally:
                    ops._wg = w
            return staticmethod(run_)
        if variant in ("wgrad_dec", "wgrad_tr"):
            patch(ops.DecoderStackFn, "_backward", without_wg(ops.DecoderStackFn._backward))
        if variant in ("wgrad_enc", "wgrad_tr"):
            patch(ops.EncoderLayerFn, "_backward", without_wg(ops.EncoderLayerFn._backward))
    elif variant == "lsap":
        def fake(cost, plan):
            idx = torch.zeros((2, plan.B, plan.Mmax), dtype=torch.int64, device=cost.device)
            idx[:] = torch.arange(plan.Mmax, device=cost.device)
            return idx[0], idx[1], torch.zeros(plan.B, dtype=torch.int32, device=cost.device)
        patch(ops, "lsap", fake)
    tr = bench.build_trainer(dev, 300, "learned", "bf16x3")
    if variant in ("images", "mirror"):
        orig_refresh = tr.mirror.refresh
        tr.mirror.refresh("all")
        part = "fwd" if variant == "images" else "bwd"
        tr.mirror.refresh = lambda p="all": None if p == part else orig_refresh(p)
    if variant == "adamw":
        tr._optimizer_step = lambda: torch.zeros((), device=dev)
    images, rects, targets = bench.synthetic_batch(2, 800, 800, (37, 120), seed=0, device=dev)
    try:
        tr.capture(images, rects, targets, warmup=1)
        ms = timed(tr)
    finally:
        for (obj, name), fn in saved.items():
            setattr(obj, name, fn)
    del tr
    torch.cuda.empty_cache()
    return ms


base = None
for v in VARIANTS:
    ms = run(v)
    if v == "none" and base is None:
        base = ms
    print("%-10s %.3f ms/step%s" % (v, ms, "" if base is None or v == "none" else "   (%+.3f)" % (ms - base)), flush=True)
