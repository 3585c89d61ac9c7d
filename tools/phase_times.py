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

from torch.profiler import profile, ProfilerActivity, record_function
from collections import defaultdict


def step_ranges():
    tr.flat_g.zero_()
    nt = NestedTensor(images, mask)
    with record_function("PH_backbone"):
        feat = model.backbone_features(nt, rects) if hasattr(model, "backbone_features") else None
    with record_function("PH_forward"):
        outputs, _ = model(nt, rects=rects)
    with record_function("PH_criterion"):
        loss_dict = criterion(outputs, targets, num_boxes=tr._num_boxes(targets))
        wd = criterion.weight_dict
        loss = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
    tr.mirror.refresh(); ops.MIRROR = tr.mirror
    with record_function("PH_backward"):
        loss.backward()
    ops.MIRROR = None
    torch.cuda.synchronize()


with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    step_ranges()
evs = prof.events()
ranges = [(e.name, e.time_range.start, e.time_range.end) for e in evs if e.name.startswith("PH_")]
# map kernels to phases through their launching CPU op time (correlation): use the kernel's linked cpu parent time
agg = defaultdict(lambda: [0, 0.0, 0, 0.0])
for e in evs:
    if e.device_type != torch.autograd.DeviceType.CUDA:
        continue
    # find launch time: use the event's corresponding cpu op if available
    t = None
    for k in getattr(e, "kernels", []) or []:
        pass
    par = getattr(e, "cpu_parent", None)
    tl = e.time_range.start
    n = e.name
    ours = "anonymous namespace)::" in n and "at::native" not in n
    ph = "?"
    agg_key = None
    agg[(ours,)][0 if ours else 2] += 0
print("phase ranges (CPU time us):", [(n, round(b - a)) for n, a, b in ranges])
# simpler and robust: profile each phase in its own profiler session
def prof_phase(fn):
    with profile(activities=[ProfilerActivity.CUDA]) as pr:
        fn(); torch.cuda.synchronize()
    k = [e for e in pr.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    ours = [e for e in k if "anonymous namespace)::" in e.name and "at::native" not in e.name]
    oth = [e for e in k if e not in ours]
    dt = lambda L: sum(getattr(e, "device_time", 0) for e in L) / 1e3
    return len(ours), dt(ours), len(oth), dt(oth), oth

state = {}
def ph_fwd():
    state["out"] = model(NestedTensor(images, mask), rects=rects)[0]
def ph_crit():
    ld = criterion(state["out"], targets, num_boxes=tr._num_boxes(targets)); wd = criterion.weight_dict
    state["loss"] = sum(ld[k] * wd[k] for k in ld if k in wd)
def ph_bwd():
    tr.mirror.refresh(); ops.MIRROR = tr.mirror
    state["loss"].backward(); ops.MIRROR = None
tr.flat_g.zero_()
for name, fn in (("forward", ph_fwd), ("criterion", ph_crit), ("backward", ph_bwd)):
    no, to, nt_, tt, oth = prof_phase(fn)
    print("%-10s ours %4d launches %7.2f ms | torch %4d launches %6.2f ms" % (name, no, to, nt_, tt))
    a2 = defaultdict(lambda: [0, 0.0])
    for e in oth:
        a = a2[e.name[:100]]; a[0] += 1; a[1] += getattr(e, "device_time", 0)
    for k, v in sorted(a2.items(), key=lambda kv: -kv[1][1])[:8]:
        print("      %4d %7.1f us  %s" % (v[0], v[1], k))
