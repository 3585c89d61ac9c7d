import sys; sys.path.insert(0, '.')
import torch, counting_detr_amd
from counting_detr_amd.args import default_args
from counting_detr_amd.engine import Trainer
from counting_detr_amd.init import seeded_init_
from bench import synthetic_batch
dev = torch.device("cuda")
args = default_args(device="cuda", num_query_position=300)
model, crit, _ = counting_detr_amd.build_model(args)
seeded_init_(model); model.to(dev).train(); crit.train()
tr = Trainer(model, crit, args, device=dev)
images, rects, targets = synthetic_batch(2, 800, 800, (37, 120), seed=0, device=dev)
for i in range(60):
    out = tr.train_step(images, rects, targets)
    if i % 10 == 0 or i == 59:
        print(i, {k: round(float(v), 4) for k, v in out.items() if k in ("loss", "loss_ce", "loss_bbox", "loss_giou", "grad_norm")}, flush=True)
