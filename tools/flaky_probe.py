import copy, sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch
import test_timed_path_gpu as T
from counting_detr_amd.engine import Trainer
DEV = "cuda"
for rep in range(4):
    model, crit, args = T._small()
    ref_model, crit2 = copy.deepcopy(model), copy.deepcopy(crit)
    b0, b1 = T._batch(2, 64, 96, (5, 9), 1), T._batch(2, 64, 96, (3, 11), 2)
    tr = Trainer(model, crit, args, device=DEV)
    tr._concurrent = lambda a, b: False
    o0 = {k: float(v) for k, v in tr.step(b0[0], b0[1], b0[2], next_samples=b1[0]).items()}
    o1 = {k: float(v) for k, v in tr.step(b1[0], b1[1], b1[2]).items()}
    tr2 = Trainer(ref_model, crit2, args, device=DEV)
    r0 = {k: float(v) for k, v in tr2.train_step(b0[0], b0[1], b0[2]).items()}
    r1 = {k: float(v) for k, v in tr2.train_step(b1[0], b1[1], b1[2]).items()}
    print(os.environ.get("TAG", ""), rep, "grad_norm rel", abs(o0["grad_norm"] - r0["grad_norm"]) / r0["grad_norm"], abs(o1["grad_norm"] - r1["grad_norm"]) / r1["grad_norm"],
          "loss rel", abs(o0["loss"] - r0["loss"]) / r0["loss"], abs(o1["loss"] - r1["loss"]) / r1["loss"], flush=True)
