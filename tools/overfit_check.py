"""Overfit one synthetic batch (B=2 800x800, T=(37,120)) for a few hundred captured steps in the default arithmetic (bf16x3 forward, bf16 backward
from bf16 twins, split reductions) and in the fp32-MFMA mode: the loss must fall the same way in both and no step may be skipped as non-finite.
usage: python tools/overfit_check.py [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, counting_detr_amd
from counting_detr_amd import ops
from counting_detr_amd.args import default_args
from counting_detr_amd.engine import Trainer
from counting_detr_amd.init import seeded_init_
from bench import synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda")
for prec in (1, 0):
    ops.PRECISION = prec
    args = default_args(device="cuda", num_query_position=300)
    model, crit, _ = counting_detr_amd.build_model(args)
    seeded_init_(model); model.to(dev).train(); crit.train()
    tr = Trainer(model, crit, args, device=dev)
    images, rects, targets = synthetic_batch(2, 800, 800, (37, 120), seed=0, device=dev)
    tr.capture(images, rects, targets, warmup=1)
    trace = []
    for i in range(steps):
        out = tr.replay(pipelined=True)        # the benchmark's step: chain of linear graphs + the next step's frozen stage beside the solve
        if i % 50 == 0 or i == steps - 1:
            trace.append((i, round(float(out["loss"]), 4), round(float(out["loss_bbox"]), 4), round(float(out["loss_giou"]), 4)))
    print(("bf16x3 fwd / bf16 bwd" if prec == 1 else "fp32 MFMA"), "non-finite steps:", tr.nonfinite_steps(), "prefetch", dict(tr.prefetch_stats), trace, flush=True)
