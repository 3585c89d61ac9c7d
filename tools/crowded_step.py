"""The captured train step on crowded images (target-capacity classes 2048 / 3800): python tools/crowded_step.py T0 T1 [H W] [steps]
Prints ms/step; run it under `rocprofv3 --kernel-trace --stats` and feed the kernel_stats CSV to tools/kernel_stats.py for the
assignment kernel's own time (lsap_kernel / lsap_wave_kernel).  Reference call it replaces: A2/models/matcher.py:243-247 (scipy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

Ts = (int(sys.argv[1]), int(sys.argv[2]))
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (384, 576)
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda", 0)
tr = bench.build_trainer(dev, 300, "learned", "bf16x3")
images, rects, targets = bench.synthetic_batch(2, H, W, Ts, seed=0, device=dev)
tr.capture(images, rects, targets, warmup=1)
rp = (lambda: tr.replay(pipelined=True)) if tr._entry.get("fs") is not None else tr.replay
for _ in range(3):
    rp()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = rp()
torch.cuda.synchronize()
print(f"T={Ts} {H}x{W} capacity {tr.target_capacity(max(Ts))}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step, loss {float(out['loss']):.4f}")
