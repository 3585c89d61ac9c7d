"""Forward intermediates with the reductions split 4 ways vs unsplit (fp32 MFMA mode).  usage: python tools/sk_check3.py"""
import os, sys, torch
os.environ.setdefault("CDETR_TUNING", "1")      # the per-call A/B knobs are only consulted when this is set at load time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_dp_shared_gpu as t
from counting_detr_amd import ops
from counting_detr_amd.misc import nested_tensor_from_tensor_list
dev = torch.device("cuda", 0)
ops.PRECISION = 0
os.environ["CDETR_GEMM_SPLITK"] = "4"


def fwd(sk):
    ops.SPLITK = sk
    tr = t._make(dev)
    images, rects, targets = t._batch(dev)
    m = tr.model
    m.taps = {}
    with torch.no_grad():
        out = m(nested_tensor_from_tensor_list(images[0:2].contiguous()), rects=rects[0:2].contiguous())
    torch.cuda.synchronize()
    d = {k: v.detach().double().cpu() for k, v in m.taps.items()}
    o = out[0] if isinstance(out, tuple) else out
    d["logits"], d["boxes"], d["vars"] = o["pred_logits"].double().cpu(), o["pred_boxes"].double().cpu(), o["pred_vars"].double().cpu()
    return d


a, b = fwd(0), fwd(1)
for k in a:
    e = (a[k] - b[k]).abs().max().item() / (a[k].abs().max().item() + 1e-30)
    print(f"{k:8s} shape {tuple(a[k].shape)}  max err / max = {e:.3e}")
