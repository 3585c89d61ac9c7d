"""GPU time of the phases of one eager train step (HIP events), next to the launch counts (torch profiler)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import build_model, ops
from counting_detr_amd.args import default_args
from counting_detr_amd.engine import Trainer
from counting_detr_amd.misc import NestedTensor
from oracle.weights import seeded_state_dict

dev = torch.device("cuda")
args = default_args()
model, criterion, _ = build_model(args)
model.load_state_dict(seeded_state_dict(), strict=True)
model.to(dev); criterion.to(dev)
tr = Trainer(model, criterion, args, device=dev)
g = torch.Generator().manual_seed(0)
B, H, W = 2, 800, 800
images = torch.randn(B, 3, H, W, generator=g).to(dev)
mask = torch.zeros(B, H, W, dtype=torch.bool, device=dev)
rects = (torch.rand(B, 3, 4, generator=g) * 0.2 + 0.1).to(dev)
targets = []
for t in (37, 120):
    cxcy = torch.rand(t, 2, generator=g) * 0.8 + 0.1
    wh = torch.rand(t, 2, generator=g) * 0.1 + 0.02
    targets.append({"boxes": torch.cat([cxcy, wh], 1).to(dev), "labels": torch.zeros(t, dtype=torch.int64, device=dev)})


def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e


def step(record=None):
    tr.flat_g.zero_()
    e = [ev()]
    nt = NestedTensor(images, mask)
    # mirror AnchorDETR.forward in pieces
    outputs, _ = model(nt, rects=rects)
    e.append(ev())
    loss_dict = criterion(outputs, targets, num_boxes=tr._num_boxes(targets))
    wd = criterion.weight_dict
    loss = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
    e.append(ev())
    tr.mirror.refresh(); ops.MIRROR = tr.mirror
    loss.backward()
    ops.MIRROR = None
    e.append(ev())
    tr._optimizer_step()
    e.append(ev())
    torch.cuda.synchronize()
    return [e[i].elapsed_time(e[i + 1]) for i in range(len(e) - 1)]


for _ in range(3):
    step()
ts = [step() for _ in range(5)]
avg = [sum(t[i] for t in ts) / len(ts) for i in range(4)]
print("eager phase ms: forward %.2f  criterion(+matcher) %.2f  backward %.2f  optimizer %.2f   total %.2f" % (*avg, sum(avg)))

from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    step()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
print("kernels in one step:", len(evs), " GPU time ms: %.2f" % (sum(e.device_time for e in evs) / 1e3 if hasattr(evs[0], "device_time") else -1))
# torch-side kernels by name, top 25 by total time
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for e in evs:
    n = e.name
    ours = "anonymous namespace)::" in n and "at::native" not in n
    if ours:
        continue
    a = agg[n[:90]]; a[0] += 1; a[1] += getattr(e, "device_time", getattr(e, "cuda_time", 0))
tot = sum(v[1] for v in agg.values())
print("torch-side kernels: %d launches, %.2f ms" % (sum(v[0] for v in agg.values()), tot / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%6d %8.1f us  %s" % (v[0], v[1], k))
