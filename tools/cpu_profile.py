"""cProfile of the host side of one eager train step (which Python functions the ~700 launches cost)."""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for _ in range(5):
    tr.train_step(images, rects, targets)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    tr.train_step(images, rects, targets)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
