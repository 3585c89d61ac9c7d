"""Loss trajectory of the bench workload: eager steps vs graph replays from the same initial state (they must agree to noise)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, counting_detr_amd
from counting_detr_amd.args import default_args
from counting_detr_amd.engine import Trainer
from counting_detr_amd.init import seeded_init_
from bench import synthetic_batch
dev = torch.device("cuda")
mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 36
args = default_args(device="cuda", num_query_position=300)
model, crit, _ = counting_detr_amd.build_model(args)
seeded_init_(model); model.to(dev).train(); crit.train()
tr = Trainer(model, crit, args, device=dev)
images, rects, targets = synthetic_batch(2, 800, 800, (37, 120), seed=0, device=dev)
if mode == "graph":
    tr.capture(images, rects, targets, warmup=0)
ls = []
for i in range(n):
    out = tr.replay() if mode == "graph" else tr.train_step(images, rects, targets)
    ls.append(float(out["loss"]))
print(mode, os.environ.get("CDETR_WGRAD_SIDE", "0"), " ".join(f"{v:.4f}" for v in ls[::3]), "last", f"{ls[-1]:.4f}", flush=True)
