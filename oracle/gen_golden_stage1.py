"""Generate tests/golden/g7_stage1.npz by running the REAL 1st-stage reference (TEST INFRA; separate process because the
1st- and 2nd-stage trees both use the top-level package names `models` / `util`).

    python -m oracle.gen_golden_stage1

Same stubbing as oracle/gen_golden.py (torchvision symbols, is_main_process -> False, Tensor.cuda -> identity).
"""
import argparse
import os
import sys

import numpy as np
import torch

REF = "/root/reference/src/CountDETR_147_1st_stage"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    from oracle import gen_golden as G
    G.REF = REF
    G.install_stubs()
    from models import build_model
    from oracle.weights import seeded_state_dict, stage1_schema
    args = argparse.Namespace(device="cpu", backbone="resnet50", dilation=True, lr_backbone=1e-5, masks=False,
                              num_feature_levels=1, hidden_dim=256, nheads=8, enc_layers=6, dec_layers=6,
                              dim_feedforward=1024, dropout=0.0, num_query_position=300, num_query_pattern=1,
                              spatial_prior="defined", attention_type="RCDA", frozen_weights=None)
    d = {}
    for name, (H, W), npts in (("n3", (64, 96), 3), ("n57", (96, 128), 57)):
        model, crit, _ = build_model(args)
        print(model.load_state_dict(seeded_state_dict(stage1_schema()), strict=True))
        model.train()
        g = torch.Generator().manual_seed(900 + npts)
        img = torch.randn(1, 3, H, W, generator=g)
        pts = torch.rand(1, npts, 2, generator=g) * 0.8 + 0.1
        whs = torch.rand(1, npts, 2, generator=g) * 0.1 + 0.02
        out = model(img, pts)
        losses = crit(out, {"points": pts, "whs": whs})
        total = sum(losses[k] * crit.weight_dict[k] for k in losses)
        total.backward()
        names = [n for n, p in model.named_parameters()]
        gn = np.array([(p.grad.norm().item() if p.grad is not None else -1.0) for n, p in model.named_parameters()])
        G.put(d, f"{name}/img", img); G.put(d, f"{name}/points", pts); G.put(d, f"{name}/whs", whs)
        for k, v in out.items():
            G.put(d, f"{name}/{k}", v)
        for k, v in losses.items():
            G.put(d, f"{name}/L_{k}", v)
        G.put(d, f"{name}/param_names", np.array(names)); G.put(d, f"{name}/grad_norms", gn)
        print(name, {k: float(v) for k, v in losses.items()})
    if "--keep-g7" not in sys.argv:
        np.savez_compressed(os.path.join(OUT, "g7_stage1.npz"), **d)
    # BASELINE config 5 at its stated size: 900 `defined` anchor points on one 800x800 image (A1/models/transformer.py:114-121),
    # forward + BoundingBoxCriterion + backward.  Inputs are regenerated from the seed on both sides (never stored).
    d = {}
    model, crit, _ = build_model(args)
    model.load_state_dict(seeded_state_dict(stage1_schema()), strict=True)
    model.train()
    from oracle.step import stage1_inputs
    img, pts, whs = stage1_inputs(800, 800, 900, seed=1900)
    taps = {}
    hk = [model.transformer.decoder_layers[i].register_forward_hook(lambda m, i_, o, k=i: taps.__setitem__(f"hs{k}", o.detach()))
          for i in range(6)]
    out = model(img, pts)
    for h in hk:
        h.remove()
    losses = crit(out, {"points": pts, "whs": whs})
    total = sum(losses[k] * crit.weight_dict[k] for k in losses)
    total.backward()
    names = [n for n, p in model.named_parameters()]
    gn = np.array([(p.grad.norm().item() if p.grad is not None else -1.0) for n, p in model.named_parameters()])
    G.put(d, "n900/cfg", np.array([800, 800, 900, 1900]))
    for k, v in out.items():
        G.put(d, f"n900/{k}", v)
    for k, v in losses.items():
        G.put(d, f"n900/L_{k}", v)
    for k, v in taps.items():
        G.put(d, f"n900/tap_{k}", G.digest(v, full_max=4096, nsamp=2048))
    G.put(d, "n900/param_names", np.array(names)); G.put(d, "n900/grad_norms", gn)
    print("n900", {k: float(v) for k, v in losses.items()})
    np.savez_compressed(os.path.join(OUT, "g11_stage1_n900.npz"), **d)


if __name__ == "__main__":
    main()
