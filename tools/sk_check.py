"""Which pass makes the fp32 gradient move when reductions are split 4 ways?  usage: python tools/sk_check.py"""
import os, sys, torch
os.environ.setdefault("CDETR_TUNING", "1")      # the per-call A/B knobs are only consulted when this is set at load time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_dp_shared_gpu as t
from counting_detr_amd import ops
dev = torch.device("cuda", 0)
ops.PRECISION = 0
lo, hi = 0, 2


def grad(fwd, bwd):
    tr = t._make(dev)
    images, rects, targets = t._batch(dev)
    crit = tr.criterion
    orig = crit.forward

    def hooked(*a, **k):
        ops.SPLITK = bwd
        return orig(*a, **k)
    crit.forward = hooked
    md = crit.matcher.match_device
    rec = {}

    def md_hook(*a, **k):
        r = md(*a, **k)
        rec["i"], rec["j"] = r[0].cpu().clone(), r[1].cpu().clone()
        return r
    crit.matcher.match_device = md_hook
    ops.SPLITK = fwd
    out = tr.train_step(images[lo:hi].contiguous(), rects[lo:hi].contiguous(), targets[lo:hi])
    torch.cuda.synchronize()
    grad.rec = rec
    return tr, tr.flat_g.detach().cpu().clone(), float(out["loss"])


def report(tag, tr, a, b):
    print(f"{tag}: max err / max |g| = {((a - b).abs().max() / a.abs().max()).item():.3e}  l2 {((a - b).norm() / a.norm()).item():.3e}")


os.environ["CDETR_GEMM_SPLITK"] = "4"
tr, off, l0 = grad(0, 0)
r0 = grad.rec
for f, b in ((1, 0), (0, 1), (1, 1)):
    _, g, l = grad(f, b)
    print("   matching identical:", torch.equal(r0["i"], grad.rec["i"]) and torch.equal(r0["j"], grad.rec["j"]),
          " differing entries:", int((r0["i"] != grad.rec["i"]).sum() + (r0["j"] != grad.rec["j"]).sum()))
    report(f"split fwd {f} bwd {b} (loss {l:.9f} vs {l0:.9f})", tr, off, g)
