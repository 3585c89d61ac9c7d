"""Which torch (aten) ops still launch kernels in one eager train step, by phase: python tools/torch_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from counting_detr_amd import build_model, ops
from counting_detr_amd.args import default_args
from counting_detr_amd.engine import Trainer
from counting_detr_amd.init import seeded_init_
from counting_detr_amd.misc import NestedTensor
dev = torch.device("cuda")
args = default_args()
model, criterion, _ = build_model(args)
seeded_init_(model); model.to(dev); criterion.to(dev)
tr = Trainer(model, criterion, args, device=dev)
g = torch.Generator().manual_seed(0)
images = torch.randn(2, 3, 800, 800, generator=g).to(dev)
mask = torch.zeros(2, 800, 800, dtype=torch.bool, device=dev)
rects = (torch.rand(2, 3, 4, generator=g) * 0.2 + 0.1).to(dev)
targets = []
for t in (37, 120):
    targets.append({"boxes": torch.cat([torch.rand(t, 2, generator=g) * 0.8 + 0.1, torch.rand(t, 2, generator=g) * 0.1 + 0.02], 1).to(dev),
                    "labels": torch.zeros(t, dtype=torch.int64, device=dev)})
for _ in range(2):
    tr.train_step(images, rects, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.train_step(images, rects, targets)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.self_device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda r: -r[2])
print("aten ops with device time in one step:")
for k, c, t in rows[:30]:
    print("%-36s calls %4d  device %8.1f us" % (k, c, t))
print("total aten device time %.2f ms over %d launches" % (sum(r[2] for r in rows) / 1e3, sum(r[1] for r in rows)))
