"""Generate tests/golden/*.npz by RUNNING THE REAL REFERENCE in this container (TEST INFRA).

    python -m oracle.gen_golden            # needs /root/reference; writes tests/golden/

The reference is imported (never copied) from /root/reference/src/CountDETR_147_2nd_stage with
  * stub `torchvision` modules providing the 4 symbols the path touches (torchvision is not installed),
  * dummy `models.anchor_center` / `models.centerness` (A2/models/__init__.py:10,12 import files that do
    not exist in the tree),
  * `models.backbone.is_main_process -> False` (no pretrained .pth in the container),
  * `torch.Tensor.cuda -> identity` (hard-coded .cuda() at A2/models/transformer.py:122,129).
Weights are NOT stored: they come from oracle.weights.seeded_state_dict on both sides.
Fixtures hold inputs + expected outputs only (data, no reference source text).
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src/CountDETR_147_2nd_stage"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def install_stubs(ref=None):
    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.9.0"
    mods = {n: types.ModuleType(n) for n in ("torchvision.models", "torchvision.models._utils",
                                             "torchvision.models.utils", "torchvision.ops", "torchvision.ops.boxes",
                                             "torchvision.ops.misc")}

    class IntermediateLayerGetter(torch.nn.ModuleDict):
        def __init__(self, model, return_layers):
            rl = dict(return_layers)
            layers = {}
            for name, module in model.named_children():
                layers[name] = module
                rl.pop(name, None)
                if not rl:
                    break
            super().__init__(layers)
            self.return_layers = dict(return_layers)

        def forward(self, x):
            out = {}
            for name, module in self.items():
                x = module(x)
                if name in self.return_layers:
                    out[self.return_layers[name]] = x
            return out

    mods["torchvision.models._utils"].IntermediateLayerGetter = IntermediateLayerGetter
    mods["torchvision.models.utils"].load_state_dict_from_url = lambda *a, **k: None
    mods["torchvision.ops.boxes"].box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    mods["torchvision.ops.misc"].interpolate = torch.nn.functional.interpolate
    mods["torchvision.ops"].roi_align = None
    sys.modules["torchvision"] = tv
    for n, m in mods.items():
        sys.modules[n] = m
    for n, fn in (("models.anchor_center", "build_anchor_center"), ("models.centerness", "build_centerness")):
        m = types.ModuleType(n)
        setattr(m, fn, lambda args: None)
        sys.modules[n] = m
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, ref or REF)
    import models.backbone as mb
    mb.is_main_process = lambda: False


def ref_args(**kw):
    d = dict(device="cpu", backbone="resnet50", dilation=True, lr_backbone=1e-5, masks=False, num_feature_levels=1,
             hidden_dim=256, nheads=8, enc_layers=6, dec_layers=6, dim_feedforward=1024, dropout=0.0,
             num_query_position=300, num_query_pattern=1, spatial_prior="learned", attention_type="RCDA",
             aux_loss=False, cost_class=2, cost_bbox=5, cost_giou=2, cls_loss_coef=2, bbox_loss_coef=5,
             giou_loss_coef=2, variance_loss_coef=2, focal_alpha=0.25, frozen_weights=None)
    d.update(kw)
    return argparse.Namespace(**d)


def gen(seed):
    return torch.Generator().manual_seed(seed)


def digest(t, full_max=20000, nsamp=4096):
    """Small stand-in for a big tensor: [l2, sum, abs-sum] + a strided sample."""
    t = t.detach().to(torch.float64).reshape(-1)
    if t.numel() <= full_max:
        return {"full": t.to(torch.float32).numpy()}
    step = max(1, t.numel() // nsamp)
    return {"stats": np.array([t.norm().item(), t.sum().item(), t.abs().sum().item()]),
            "sample": t[::step][:nsamp].to(torch.float32).numpy(), "step": np.array(step)}


def put(d, name, val):
    if isinstance(val, dict):
        for k, v in val.items():
            d[f"{name}/{k}"] = v
    elif isinstance(val, torch.Tensor):
        d[name] = val.detach().cpu().numpy()
    else:
        d[name] = np.asarray(val)


def synth_targets(T, g):
    cxcy = torch.rand(T, 2, generator=g) * 0.8 + 0.1
    wh = torch.rand(T, 2, generator=g) * 0.10 + 0.02
    return {"boxes": torch.cat([cxcy, wh], 1), "labels": torch.zeros(T, dtype=torch.int64)}


def synth_preds(B, Q, g, neg_var=False):
    logits = torch.randn(B, Q, 2, generator=g) * 1.5 - 1.0
    cxcy = torch.rand(B, Q, 2, generator=g) * 0.9 + 0.05
    wh = torch.rand(B, Q, 2, generator=g) * 0.15 + 0.01
    pvars = torch.rand(B, Q, 2, generator=g) * 0.5 + 0.05
    if neg_var:
        pvars[0, ::7, 0] = -0.1
    return {"pred_logits": logits, "pred_boxes": torch.cat([cxcy, wh], -1), "pred_vars": pvars}


# ----------------------------------------------------------------------------- G1 RCDA
def g1_rcda():
    from models.row_column_decoupled_attention import MultiheadRCDA
    from oracle.weights import _make
    d = {}
    cases = [("enc_like", 64, 8, 1, 6, 9, None, False), ("dec_like_masked", 64, 8, 2, 7, 5, 11, True),
             ("square_masked", 64, 4, 2, 6, 6, 36, True), ("e256", 256, 8, 1, 5, 8, 13, True)]
    for name, E, nh, N, H, W, L, masked in cases:
        g = gen(100 + len(name))
        self_attn = L is None
        L = H * W if L is None else L
        m = MultiheadRCDA(E, nh, dropout=0.0)
        sd = {"in_proj_weight": _make(f"g1.{name}.in_w", (5 * E, E), "linear"),
              "in_proj_bias": _make(f"g1.{name}.in_b", (5 * E,), "bias"),
              "out_proj.weight": _make(f"g1.{name}.out_w", (E, E), "linear"),
              "out_proj.bias": _make(f"g1.{name}.out_b", (E,), "bias")}
        m.load_state_dict(sd)
        v = torch.randn(N, H, W, E, generator=g)
        kr = v + torch.randn(N, 1, W, E, generator=g)
        kc = v + torch.randn(N, H, 1, E, generator=g)
        if self_attn:
            qr, qc = kr.reshape(N, L, E).clone(), kc.reshape(N, L, E).clone()
        else:
            qr, qc = torch.randn(N, L, E, generator=g), torch.randn(N, L, E, generator=g)
        mask = None
        if masked:
            mask = torch.zeros(N, H, W, dtype=torch.bool)
            mask[0, H - 2:, :] = True
            mask[0, :, W - 1:] = True
            if N > 1:
                mask[1, :, W - 2:] = True
        ins = [t.requires_grad_(True) for t in (qr, qc, kr, kc, v)]
        out, _ = m(*ins, key_padding_mask=mask)
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout)
        c = {"E": E, "nh": nh, "mask": mask if mask is not None else torch.zeros(0), "gout": gout, "out": out}
        for nm, t in zip(("qr", "qc", "kr", "kc", "v"), ins):
            c[nm] = t
            c["g_" + nm] = digest(t.grad)
        for pn, p in m.named_parameters():
            c["gp_" + pn] = digest(p.grad)
        for k, val in c.items():
            put(d, f"{name}/{k}", val)
    np.savez_compressed(os.path.join(OUT, "g1_rcda.npz"), **d)


# ----------------------------------------------------------------------------- G2 positional
def g2_pos():
    from models.transformer import mask2pos, pos2posemb1d, pos2posemb2d
    from util.misc import inverse_sigmoid
    d = {}
    mask = torch.zeros(2, 5, 7, dtype=torch.bool)
    mask[0, 4:, :] = True
    mask[0, :, 5:] = True
    y, x = mask2pos(mask)
    put(d, "mask", mask); put(d, "pos_col", y); put(d, "pos_row", x)
    p1 = torch.tensor([[0.0, 0.07142857, 0.5, 0.99, 1.0]])
    put(d, "p1", p1); put(d, "emb1d", pos2posemb1d(p1))
    p2 = torch.rand(2, 6, 2, generator=gen(5))
    put(d, "p2", p2); put(d, "emb2d", pos2posemb2d(p2))
    xs = torch.tensor([0.0, 1e-6, 1e-5, 0.25, 0.5, 0.99999, 1.0, 1.5, -0.2])
    put(d, "isig_in", xs); put(d, "isig_out", inverse_sigmoid(xs))
    np.savez_compressed(os.path.join(OUT, "g2_pos.npz"), **d)


# ----------------------------------------------------------------------------- G3 bottleneck block
def g3_block():
    from models.backbone import FrozenBatchNorm2d
    from models.resnet import Bottleneck, conv1x1
    from oracle.weights import _make
    d = {}
    for name, inpl, planes, stride, dil, down in (("s2_down", 16, 8, 2, 1, True), ("dil2", 32, 8, 1, 2, False)):
        ds = None
        if down:
            ds = torch.nn.Sequential(conv1x1(inpl, planes * 4, stride), FrozenBatchNorm2d(planes * 4))
        blk = Bottleneck(inpl, planes, stride, ds, dilation=dil, norm_layer=FrozenBatchNorm2d)
        sd = {}
        for k, v in blk.state_dict().items():
            kind = "conv" if v.ndim == 4 else "bn_" + k.split(".")[-1]
            sd[k] = _make(f"g3.{name}.{k}", tuple(v.shape), kind)
        blk.load_state_dict(sd)
        x = torch.randn(2, inpl, 9, 11, generator=gen(7)).requires_grad_(True)
        y = blk(x)
        gy = torch.randn(y.shape, generator=gen(8))
        y.backward(gy)
        put(d, f"{name}/x", x); put(d, f"{name}/y", y); put(d, f"{name}/gy", gy); put(d, f"{name}/gx", x.grad)
        for pn, p in blk.named_parameters():
            put(d, f"{name}/gp_{pn}", p.grad)
        put(d, f"{name}/cfg", np.array([inpl, planes, stride, dil, int(down)]))
    np.savez_compressed(os.path.join(OUT, "g3_block.npz"), **d)


# ----------------------------------------------------------------------------- G4 matcher + G5 criterion
G45_CASES = [("q300_t37", 300, (37,), False), ("q576_t200", 576, (200,), False), ("q900_t56", 900, (56,), False),
             ("q300_t450", 300, (450,), False), ("q900_t900", 900, (900,), False), ("b2_q40", 40, (7, 13), False),
             ("b2_q300", 300, (37, 120), False), ("negvar", 50, (9,), True), ("t0", 30, (0, 5), False)]
# FSC-147 images hold up to 3731 objects (A2/data/fsc147.py:80-84 feeds every one of them to the matcher, A2/models/matcher.py:229-247):
# the target-capacity classes 1024 / 2048 / 3800 of the device matcher and criterion are pinned by these
G45_LARGE_T = [("q300_t3000", 300, (3000,), False), ("q576_t3731", 576, (3731,), False), ("q900_t3000", 900, (3000,), False),
               ("b2_q300_t2100", 300, (37, 2100), False), ("q300_t1100", 300, (1100,), False)]


def g45_matcher_criterion(cases=G45_CASES, fname="g45_matcher_criterion.npz"):
    from models.anchor_detr import SetCriterion
    from models.matcher import OriginalHungarianMatcher
    matcher = OriginalHungarianMatcher(2, 5, 2)
    wd = {"loss_ce": 2, "loss_bbox": 5, "loss_giou": 2, "loss_variance": 2}
    crit = SetCriterion(1, matcher, wd, ["labels", "boxes", "cardinality", "vars"], focal_alpha=0.25)
    d = {}
    for name, Q, Ts, neg in cases:
        g = gen(1000 + Q + sum(Ts))
        B = len(Ts)
        outs = synth_preds(B, Q, g, neg_var=neg)
        tg = [synth_targets(T, g) for T in Ts]
        for k in outs:
            outs[k].requires_grad_(True)
        idx = matcher(outs, tg)
        # full cost (incl. the cross-image blocks the reference computes) only for the small cases
        if Q * sum(Ts) <= 20000:
            with torch.no_grad():
                from util.box_ops import box_cxcywh_to_xyxy, generalized_box_iou
                p = outs["pred_logits"].flatten(0, 1).sigmoid()
                ob = outs["pred_boxes"].flatten(0, 1)
                tb = torch.cat([t["boxes"] for t in tg])
                ids = torch.cat([t["labels"] for t in tg])
                neg_c = 0.75 * (p ** 2.0) * (-(1 - p + 1e-8).log())
                pos_c = 0.25 * ((1 - p) ** 2.0) * (-(p + 1e-8).log())
                C = 5 * torch.cdist(ob, tb, p=1) + 2 * (pos_c[:, ids] - neg_c[:, ids]) + \
                    2 * (-generalized_box_iou(box_cxcywh_to_xyxy(ob), box_cxcywh_to_xyxy(tb)))
                put(d, f"{name}/C", C.view(B, Q, -1))
        losses = crit(outs, tg)
        total = sum(losses[k] * wd[k] for k in losses if k in wd)
        if not neg:
            total.backward()
            for k in outs:
                put(d, f"{name}/g_{k}", outs[k].grad)
        for k, v in outs.items():
            put(d, f"{name}/{k}", v)
        for b, t in enumerate(tg):
            put(d, f"{name}/tgt{b}", t["boxes"])
            put(d, f"{name}/idx_i{b}", idx[b][0]); put(d, f"{name}/idx_j{b}", idx[b][1])
        for k, v in losses.items():
            put(d, f"{name}/L_{k}", v)
        put(d, f"{name}/B", np.array(B))
    np.savez_compressed(os.path.join(OUT, fname), **d)


def g45_large_t():
    g45_matcher_criterion(G45_LARGE_T, "g45_large_t.npz")


# ----------------------------------------------------------------------------- G6 end-to-end tiny
def g6_e2e():
    from models import build_model
    from oracle.weights import model_schema, seeded_state_dict
    d = {}
    cases = [("b1_64x96", [(64, 96)], (9,), "learned", 300), ("b2_pad", [(128, 160), (96, 128)], (7, 13), "learned", 300),
             ("b1_grid20", [(96, 96)], (5,), "grid", 20)]
    rects_edge = torch.tensor([[.10, .10, .20, .20], [.40, .40, .50, .55], [.70, .20, .80, .30]])
    for name, sizes, Ts, prior, nq in cases:
        args = ref_args(spatial_prior=prior, num_query_position=nq)
        model, crit, _ = build_model(args)
        sd = seeded_state_dict(model_schema(num_position=nq, spatial_prior=prior))
        missing = model.load_state_dict(sd, strict=True)
        model.train(); crit.train()
        g = gen(4242 + len(name))
        imgs = [torch.randn(3, h, w, generator=g) for h, w in sizes]
        B = len(imgs)
        rects = rects_edge[None].repeat(B, 1, 1).clone()
        if B > 1:
            rects[1] = torch.tensor([[.3, .3, .4, .4], [.5, .1, .6, .2], [.2, .6, .3, .7]])  # must NOT matter
        tg = [synth_targets(T, g) for T in Ts]
        samples = torch.stack(imgs) if B == 1 else imgs
        out, ref = model(samples, rects=rects)
        losses = crit(out, tg)
        wd = crit.weight_dict
        total = sum(losses[k] * wd[k] for k in losses if k in wd)
        params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.AdamW([{"params": [p for n, p in model.named_parameters() if "backbone" not in n and p.requires_grad], "lr": 1e-4},
                                 {"params": [p for n, p in model.named_parameters() if "backbone" in n and p.requires_grad], "lr": 1e-5}],
                                lr=1e-4, weight_decay=1e-4)
        opt.zero_grad()
        total.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
        idx = crit.matcher({k: v for k, v in out.items()}, tg)
        names = [n for n, p in model.named_parameters()]
        gnorm = np.array([(p.grad.norm().item() if p.grad is not None else -1.0) for n, p in model.named_parameters()])
        opt.step()
        psum = np.array([p.detach().double().sum().item() for n, p in model.named_parameters()])
        for i, im in enumerate(imgs):
            put(d, f"{name}/img{i}", im)
        put(d, f"{name}/rects", rects)
        for b, t in enumerate(tg):
            put(d, f"{name}/tgt{b}", t["boxes"])
            put(d, f"{name}/idx_i{b}", idx[b][0]); put(d, f"{name}/idx_j{b}", idx[b][1])
        for k, v in out.items():
            put(d, f"{name}/{k}", v)
        put(d, f"{name}/ref", ref)
        for k, v in losses.items():
            put(d, f"{name}/L_{k}", v)
        put(d, f"{name}/grad_total_norm", gn)
        put(d, f"{name}/param_names", np.array(names))
        put(d, f"{name}/grad_norms_clipped", gnorm)
        put(d, f"{name}/param_sums_after_step", psum)
        put(d, f"{name}/cfg", np.array([B, nq, int(prior == "grid")]))
        print(name, {k: float(v) for k, v in losses.items()}, "gn", float(gn), missing)
    np.savez_compressed(os.path.join(OUT, "g6_e2e.npz"), **d)


# ----------------------------------------------------------------------------- G8 count rule
def g8_count():
    d = {}
    logits = torch.randn(3, 50, 2, generator=gen(77)) * 2
    logits[0, 0, 0] = 0.0      # sigmoid == 0.5 exactly -> counted (>=)
    prob = logits.sigmoid()[..., 0]
    put(d, "logits", logits)
    put(d, "counts", (prob >= 0.5).sum(-1))
    gt = np.array([20, 31, 7])
    pred = (prob >= 0.5).sum(-1).numpy()
    err = np.abs(gt - pred).astype(np.float64)
    put(d, "gt", gt)
    put(d, "metrics", np.array([err.mean(), np.sqrt((err ** 2).mean()), (err / gt).mean(), np.sqrt((err ** 2 / gt).mean())]))
    np.savez_compressed(os.path.join(OUT, "g8_count.npz"), **d)


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    torch.manual_seed(0)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g45", "g45L", "g6", "g8"]
    fns = {"g1": g1_rcda, "g2": g2_pos, "g3": g3_block, "g45": g45_matcher_criterion, "g45L": g45_large_t, "g6": g6_e2e, "g8": g8_count}
    for w in which:
        fns[w]()
        print("wrote", w)


if __name__ == "__main__":
    main()
