"""does a "bf16x2" FORWARD (hi*hi + lo*hi: the weight operand rounded to bf16, the activation split -- 2 MFMAs per
product instead of 3) hold the parity contract?  Runs the full-size golden cases of tests/test_full_size_gpu.py (outputs of the real
reference) with every forward GEMM of the tile kernels at cdetr_gemm_desc.precision = 2 and prints, per case, the worst intermediate /
output error against the 1e-3 bar and whether the Hungarian indices are still exact.  usage: python tools/fwd_bf16x2.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from counting_detr_amd import ops
from fullsize import TAP_KEYS, case_inputs, rel_err, NSAMP
import test_full_size_gpu as T

z = np.load(os.path.join(ROOT, "tests", "golden", "g10_full.npz"), allow_pickle=False)
orig = ops.gemm_raw


def forced(*a, **kw):
    if kw.get("precision") is None:              # forward calls leave it to ops.PRECISION; the backward passes its own
        kw["precision"] = MODE
    return orig(*a, **kw)


for MODE, tag in ((1, "bf16x3 (3 MFMAs per product, the default)"), (2, "bf16x2 (weight rounded to bf16, 2 MFMAs)"), (3, "bf16 (1 MFMA)")):
    ops.gemm_raw = forced
    ops.PRECISION = 1
    for name in ("cfg2", "shipped576", "small_b2"):
        c = case_inputs(z, name)
        model, crit, args = T.build(c)
        imgs, rects, tg = T.to_dev(c)
        model.taps = {}
        with torch.no_grad():
            out, ref = model(imgs, rects=rects)
        taps, model.taps = model.taps, None
        worst = {}
        for k in TAP_KEYS:
            key = f"{name}/tap_{k}"
            t = T.nchw(taps[k]).detach().to(torch.float64).reshape(-1).cpu()
            if f"{key}/full" in z:
                worst[k] = rel_err(t.numpy(), z[f"{key}/full"])
            else:
                worst[k] = rel_err(t[::int(z[f"{key}/step"])][:NSAMP].numpy(), z[f"{key}/sample"])
        for k in ("pred_logits", "pred_boxes", "pred_vars"):
            worst[k] = rel_err(out[k].detach().cpu().numpy(), z[f"{name}/{k}"])
        idx = crit.matcher({k: v for k, v in out.items() if k != "aux_outputs"}, tg)
        exact = all(np.array_equal(idx[b][0].numpy(), z[f"{name}/idx_i{b}"]) and np.array_equal(idx[b][1].numpy(), z[f"{name}/idx_j{b}"])
                    for b in range(c["B"]))
        nbad = sum(int((idx[b][1].numpy() != z[f"{name}/idx_j{b}"]).sum()) for b in range(c["B"]))
        w = max(worst.values())
        print(f"{tag:45s} {name:11s} worst {w:.2e} ({max(worst, key=worst.get)}) outputs " +
              " ".join(f"{k[5:]}={worst[k]:.1e}" for k in ("pred_logits", "pred_boxes", "pred_vars")) +
              f" | layer4 {worst['layer4']:.1e} enc5 {worst['enc5']:.1e} hs5 {worst['hs5']:.1e} | indices exact: {exact} ({nbad} differ)", flush=True)
    ops.gemm_raw = orig
