"""Where the torch-side small kernels of one eager train step come from: fill / add / copy / ... launches grouped by the
enclosing autograd node or ATen op and input shapes (torch.profiler).  Result of the last run: 12 zero-fills are the RCDA
backward atomics buffers, 12 adds are autograd summing d(posemb) over the encoder layers, ~30 launches are the box-head tail
(inverse_sigmoid / cat / select backward); under graph replay each costs well below its stand-alone 4-5 us."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_step(images, rects, targets)
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::copy_", "aten::mul", "aten::sum", "aten::cat", "aten::index",
        "aten::sub", "aten::div", "aten::where", "aten::clone", "aten::contiguous", "aten::index_select", "aten::gather")
groups = collections.Counter()
for ev in prof.events():
    if ev.name not in want or ev.device_time_total <= 0:
        continue
    chain, pa = [], ev.cpu_parent
    while pa is not None:
        chain.append(pa.name); pa = pa.cpu_parent
    shp = str(ev.input_shapes)[:60] if ev.input_shapes else ""
    groups[(ev.name, " < ".join(chain[:3]) + "  " + shp)] += 1
for (name, fr), n in sorted(groups.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{n:4d}  {name:18s} {fr[-110:]}")
