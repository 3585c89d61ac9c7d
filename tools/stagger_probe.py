"""VERDICT r4 item 4, the cheap upper bound: do two SINGLE-image steps running concurrently (each a chain of linear graphs on its own probed
streams, one half a phase behind the other) finish two images sooner than the one two-image chain does?  Two independent trainers (two model
copies: no shared gradient arena, no shared optimizer -- the real thing would add atomics on the shared weight gradients and one optimizer
pass, i.e. cost more) -> if THIS is not faster than the B=2 step, the shared-weight version cannot be.
python tools/stagger_probe.py [H W] ; GPU_MAX_HW_QUEUES=8 recommended (six streams)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from counting_detr_amd import _ffi

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 800)
dev = torch.device("cuda", 0)
N = 20


def timed(fn, n=N):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


tr2 = bench.build_trainer(dev, 300, "learned", "bf16x3")
im, re, tg = bench.synthetic_batch(2, H, W, (37, 120), seed=0, device=dev)
tr2.capture(im, re, tg, warmup=1)
t_b2 = timed(lambda: tr2.replay(pipelined=True))
print(f"B=2 chain, pipelined: {t_b2:.3f} ms per step (2 images)  streams {tr2.side_stream_probe}", flush=True)
del tr2
torch.cuda.empty_cache()

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
trs = []
for s, T in ((sA, 37), (sB, 120)):
    with torch.cuda.stream(s):
        tr = bench.build_trainer(dev, 300, "learned", "bf16x3")
        im1, re1, tg1 = bench.synthetic_batch(1, H, W, (T,), seed=T, device=dev)
        tr.capture(im1, re1, tg1, warmup=1)
        trs.append(tr)
    torch.cuda.synchronize()
for s, tr, nm in ((sA, trs[0], "A (T=37)"), (sB, trs[1], "B (T=120)")):
    with torch.cuda.stream(s):
        t = timed(lambda: tr.replay(pipelined=True))
    print(f"B=1 chain {nm} alone: {t:.3f} ms per step  streams {tr.side_stream_probe}", flush=True)

for off_us in (0, 1500, 3000, 4500):
    def both():
        with torch.cuda.stream(sA):
            trs[0].replay(pipelined=True)
        with torch.cuda.stream(sB):
            trs[1].replay(pipelined=True)
    for _ in range(3):
        both()
    torch.cuda.synchronize()
    with torch.cuda.stream(sB):      # B starts `off_us` behind A; both then free-run: the offset persists while their step times agree
        if off_us:
            _ffi.check(_ffi.lib().cdetr_delay(off_us, _ffi.stream_ptr()), "cdetr_delay")
    t0 = time.perf_counter()
    for _ in range(N):
        both()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / N * 1e3
    print(f"two B=1 chains concurrently, B {off_us} us behind A: {t:.3f} ms per 2 images ({t / t_b2:.3f}x the B=2 chain)", flush=True)
