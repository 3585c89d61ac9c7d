"""Every device kernel of ONE eager train step that is NOT one of ours (torch / ATen / rocclr launches), in issue order, with the
ATen op, input shapes and the innermost frames of this repo that issued it (torch.profiler, with_stack)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, counting_detr_amd
from torch.profiler import profile, ProfilerActivity
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
for _ in range(4):
    tr.train_step(images, rects, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.train_step(images, rects, targets)
    torch.cuda.synchronize()
rows = []
n_ours = 0
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or ev.device_time_total <= 0 or not ev.name.startswith("aten::"):
        continue
    # leaf ops only (an op whose child also has device time is a wrapper)
    if any(c.device_time_total > 0 and c.name.startswith("aten::") for c in (ev.cpu_children or [])):
        continue
    st = [f for f in (ev.stack or []) if "/counting_detr_amd/" in f or "bench.py" in f or "/tools/" in f]
    chain, pa = [], ev.cpu_parent
    while pa is not None and len(chain) < 3:
        chain.append(pa.name); pa = pa.cpu_parent
    rows.append((ev.time_range.start, ev.name, ev.device_time_total, str(ev.input_shapes)[:70], " < ".join(chain), " | ".join(s.split("/")[-1][:60] for s in st[:3])))
rows.sort()
tot = sum(r[2] for r in rows)
print(f"{len(rows)} torch-side device ops, {tot:.0f} us of device time in one eager step")
agg = collections.Counter(); aggt = collections.Counter()
for _, name, dt, shp, chain, st in rows:
    agg[(name, st)] += 1; aggt[(name, st)] += dt
print("---- grouped by (op, origin) ----")
for k, n in sorted(agg.items(), key=lambda kv: -aggt[kv[0]]):
    print(f"{n:4d} {aggt[k]:8.1f}us  {k[0]:22s} {k[1]}")
print("---- in issue order ----")
for t0, name, dt, shp, chain, st in rows:
    print(f"{dt:6.1f}us {name:22s} {shp:70s} {chain[:60]:60s} {st}")
