"""Pin the oracle (oracle/*.py CPU restatement) against golden vectors captured from the REAL reference
(oracle/gen_golden.py, run in the build container).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import criterion as OC
from oracle import model as OM
from oracle.weights import _make, model_schema, seeded_state_dict


def T(a):
    return torch.from_numpy(np.asarray(a))


def check_digest(z, key, t, rtol=2e-4, atol=2e-6):
    t = t.detach().to(torch.float64).reshape(-1)
    if f"{key}/full" in z:
        np.testing.assert_allclose(t.numpy(), z[f"{key}/full"], rtol=rtol, atol=atol)
        return
    stats = z[f"{key}/stats"]
    step = int(z[f"{key}/step"])
    np.testing.assert_allclose(t[::step][:4096].numpy(), z[f"{key}/sample"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(t.norm().item(), stats[0], rtol=1e-4)
    np.testing.assert_allclose(t.abs().sum().item(), stats[2], rtol=1e-4)


@pytest.mark.parametrize("name", ["enc_like", "dec_like_masked", "square_masked", "e256"])
def test_g1_rcda(golden, name):
    z = golden("g1_rcda.npz")
    E, nh = int(z[f"{name}/E"]), int(z[f"{name}/nh"])
    sd = {"a.in_proj_weight": _make(f"g1.{name}.in_w", (5 * E, E), "linear"),
          "a.in_proj_bias": _make(f"g1.{name}.in_b", (5 * E,), "bias"),
          "a.out_proj.weight": _make(f"g1.{name}.out_w", (E, E), "linear"),
          "a.out_proj.bias": _make(f"g1.{name}.out_b", (E,), "bias")}
    for v in sd.values():
        v.requires_grad_(True)
    ins = [T(z[f"{name}/{k}"]).clone().requires_grad_(True) for k in ("qr", "qc", "kr", "kc", "v")]
    mask = T(z[f"{name}/mask"])
    mask = mask if mask.numel() else None
    out = OM.rcda(*ins, sd, "a", mask=mask, nh=nh)
    np.testing.assert_allclose(out.detach().numpy(), z[f"{name}/out"], rtol=1e-4, atol=2e-6)
    out.backward(T(z[f"{name}/gout"]))
    for k, t in zip(("qr", "qc", "kr", "kc", "v"), ins):
        check_digest(z, f"{name}/g_{k}", t.grad)
    for pn in ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias"):
        check_digest(z, f"{name}/gp_{pn}", sd["a." + pn].grad)


def test_g2_positional(golden):
    z = golden("g2_pos.npz")
    y, x = OM.mask2pos(T(z["mask"]))
    np.testing.assert_allclose(y.numpy(), z["pos_col"], rtol=1e-6)
    np.testing.assert_allclose(x.numpy(), z["pos_row"], rtol=1e-6)
    np.testing.assert_allclose(OM.pos2posemb1d(T(z["p1"])).numpy(), z["emb1d"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(OM.pos2posemb2d(T(z["p2"])).numpy(), z["emb2d"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(OM.inverse_sigmoid(T(z["isig_in"])).numpy(), z["isig_out"], rtol=1e-6)


@pytest.mark.parametrize("name", ["s2_down", "dil2"])
def test_g3_bottleneck(golden, name):
    z = golden("g3_block.npz")
    inpl, planes, stride, dil, down = [int(v) for v in z[f"{name}/cfg"]]
    shapes = {"conv1.weight": (planes, inpl, 1, 1), "conv2.weight": (planes, planes, 3, 3),
              "conv3.weight": (planes * 4, planes, 1, 1)}
    if down:
        shapes["downsample.0.weight"] = (planes * 4, inpl, 1, 1)
    sd = {}
    for k, s in shapes.items():
        sd["b." + k] = _make(f"g3.{name}.{k}", s, "conv").requires_grad_(True)
    bns = [("bn1", planes), ("bn2", planes), ("bn3", planes * 4)] + ([("downsample.1", planes * 4)] if down else [])
    for bn, c in bns:
        for f in ("weight", "bias", "running_mean", "running_var"):
            sd[f"b.{bn}.{f}"] = _make(f"g3.{name}.{bn}.{f}", (c,), "bn_" + f)
    x = T(z[f"{name}/x"]).clone().requires_grad_(True)
    y = OM.bottleneck(x, sd, "b", stride, dil, bool(down))
    np.testing.assert_allclose(y.detach().numpy(), z[f"{name}/y"], rtol=1e-5, atol=1e-6)
    y.backward(T(z[f"{name}/gy"]))
    np.testing.assert_allclose(x.grad.numpy(), z[f"{name}/gx"], rtol=1e-4, atol=1e-5)
    for k in shapes:
        np.testing.assert_allclose(sd["b." + k].grad.numpy(), z[f"{name}/gp_{k}"], rtol=1e-4, atol=1e-5)


G45 = ["q300_t37", "q576_t200", "q900_t56", "q300_t450", "q900_t900", "b2_q40", "b2_q300", "negvar", "t0"]
G45L = ["q300_t3000", "q576_t3731", "q900_t3000", "b2_q300_t2100", "q300_t1100"]      # FSC-147's crowded images (up to 3731 targets)


@pytest.mark.parametrize("name", G45 + G45L)
def test_g45_matcher_criterion(golden, name):
    z = golden("g45_large_t.npz" if name in G45L else "g45_matcher_criterion.npz")
    B = int(z[f"{name}/B"])
    outs = {k: T(z[f"{name}/{k}"]).clone().requires_grad_(True) for k in ("pred_logits", "pred_boxes", "pred_vars")}
    tg = []
    for b in range(B):
        bx = T(z[f"{name}/tgt{b}"]).reshape(-1, 4)
        tg.append({"boxes": bx, "labels": torch.zeros(bx.shape[0], dtype=torch.int64)})
    idx = OC.hungarian_match(outs, tg)
    for b in range(B):   # bit-exact Hungarian indices (north_star)
        assert np.array_equal(idx[b][0].numpy(), z[f"{name}/idx_i{b}"])
        assert np.array_equal(idx[b][1].numpy(), z[f"{name}/idx_j{b}"])
    if f"{name}/C" in z:   # per-image block of the reference's full cost matrix
        C = z[f"{name}/C"]
        off = 0
        for b in range(B):
            t = tg[b]["boxes"].shape[0]
            c = OC.match_cost(outs["pred_logits"][b].detach(), outs["pred_boxes"][b].detach(), tg[b]["boxes"])
            np.testing.assert_allclose(c.numpy(), C[b][:, off:off + t], rtol=1e-5, atol=1e-6)
            off += t
    losses, _ = OC.set_criterion(outs, tg, indices=idx)
    for k in ("loss_ce", "class_error", "loss_bbox", "loss_giou", "cardinality_error", "loss_variance"):
        ref = z[f"{name}/L_{k}"]
        if np.isnan(ref):
            assert torch.isnan(losses[k]), k     # negative variance -> NaN, like the reference
        else:
            np.testing.assert_allclose(losses[k].item(), ref, rtol=2e-5, atol=1e-6, err_msg=k)
    if name != "negvar":
        OC.total_loss(losses).backward()
        for k in outs:
            np.testing.assert_allclose(outs[k].grad.numpy(), z[f"{name}/g_{k}"], rtol=1e-4, atol=1e-7, err_msg=k)


@pytest.mark.parametrize("name", ["b1_64x96", "b2_pad", "b1_grid20"])
def test_g6_end_to_end(golden, name):
    z = golden("g6_e2e.npz")
    B, nq, is_grid = [int(v) for v in z[f"{name}/cfg"]]
    prior = "grid" if is_grid else "learned"
    sd = seeded_state_dict(model_schema(num_position=nq, spatial_prior=prior))
    names = [str(n) for n in z[f"{name}/param_names"]]
    frozen = lambda n: n.startswith("backbone.body.conv1") or n.startswith("backbone.body.layer1")  # noqa: E731
    for n in names:
        if not frozen(n):
            sd[n].requires_grad_(True)
    # the 6 head aliases are ONE parameter in the reference (A2/models/transformer.py:104-107)
    for n in names:
        for fam in ("cls_embed", "bbox_embed", "bbox_variance"):
            if f"transformer.{fam}.0." in n:
                for i in range(1, 6):
                    sd[n.replace(f"{fam}.0.", f"{fam}.{i}.")] = sd[n]
    imgs = [T(z[f"{name}/img{i}"]) for i in range(B)]
    rects = T(z[f"{name}/rects"])
    tg = []
    for b in range(B):
        bx = T(z[f"{name}/tgt{b}"]).reshape(-1, 4)
        tg.append({"boxes": bx, "labels": torch.zeros(bx.shape[0], dtype=torch.int64)})
    out, ref = OM.forward(imgs, rects, sd, spatial_prior=prior, num_position=nq)
    for k in ("pred_logits", "pred_boxes", "pred_vars"):
        np.testing.assert_allclose(out[k].detach().numpy(), z[f"{name}/{k}"], rtol=1e-3, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(ref.detach().numpy(), z[f"{name}/ref"], rtol=1e-6)
    losses, idx = OC.set_criterion(out, tg)
    for b in range(B):
        assert np.array_equal(idx[b][0].numpy(), z[f"{name}/idx_i{b}"])
        assert np.array_equal(idx[b][1].numpy(), z[f"{name}/idx_j{b}"])
    for k in ("loss_ce", "loss_bbox", "loss_giou", "cardinality_error", "loss_variance"):
        np.testing.assert_allclose(losses[k].item(), z[f"{name}/L_{k}"], rtol=1e-4, err_msg=k)
    OC.total_loss(losses).backward()
    params = {n: sd[n] for n in names}
    grads = [p.grad for p in params.values() if p.grad is not None]
    total = torch.norm(torch.stack([g.norm() for g in grads]))
    np.testing.assert_allclose(total.item(), z[f"{name}/grad_total_norm"], rtol=1e-3)
    coef = min(1.0, 0.1 / (total.item() + 1e-6))
    gn_ref = z[f"{name}/grad_norms_clipped"]
    for n, r in zip(names, gn_ref):
        p = params[n]
        if r < 0:
            assert p.grad is None, n     # frozen stem/layer1 and the unused input_proj.* 
        else:
            np.testing.assert_allclose(p.grad.norm().item() * coef, r, rtol=5e-3, atol=1e-7, err_msg=n)


def test_g6_adamw_step(golden):
    """fwd + criterion + bwd + clip(0.1) + AdamW of the oracle reproduces the reference's post-step parameters."""
    from oracle.step import OracleTrainer
    z = golden("g6_e2e.npz")
    name = "b1_64x96"
    tr = OracleTrainer(num_position=300)
    imgs = [T(z[f"{name}/img0"])]
    bx = T(z[f"{name}/tgt0"]).reshape(-1, 4)
    tg = [{"boxes": bx, "labels": torch.zeros(bx.shape[0], dtype=torch.int64)}]
    _, _, _, gn = tr.step(imgs, T(z[f"{name}/rects"]), tg)
    np.testing.assert_allclose(gn.item(), z[f"{name}/grad_total_norm"], rtol=1e-3)
    names = [str(n) for n in z[f"{name}/param_names"]]
    sums = z[f"{name}/param_sums_after_step"]
    for n, s in zip(names, sums):
        np.testing.assert_allclose(tr.sd[n].detach().double().sum().item(), s, rtol=1e-5, atol=2e-4, err_msg=n)


def test_g8_count_rule(golden):
    z = golden("g8_count.npz")
    counts = OC.count_objects(T(z["logits"]))
    assert np.array_equal(counts.numpy(), z["counts"])
    m = OC.counting_metrics(counts.numpy(), z["gt"])
    np.testing.assert_allclose([m["MAE"], m["RMSE"], m["NAE"], m["SRE"]], z["metrics"], rtol=1e-12)


@pytest.mark.parametrize("name", ["n3", "n57"])
def test_g7_stage1(golden, name):
    """1st-stage variant: `defined` anchor points, no variance head, BoundingBoxCriterion."""
    from oracle.weights import stage1_schema
    z = golden("g7_stage1.npz")
    sd = seeded_state_dict(stage1_schema())
    names = [str(n) for n in z[f"{name}/param_names"]]
    for n in names:
        if not (n.startswith("backbone.body.conv1") or n.startswith("backbone.body.layer1")):
            sd[n].requires_grad_(True)
    for n in names:
        for fam in ("cls_embed", "bbox_embed"):
            if f"transformer.{fam}.0." in n:
                for i in range(1, 6):
                    sd[n.replace(f"{fam}.0.", f"{fam}.{i}.")] = sd[n]
    pts, whs = T(z[f"{name}/points"]), T(z[f"{name}/whs"])
    out = OM.forward_stage1(T(z[f"{name}/img"]), pts, sd)
    for k in ("pred_logits", "pred_wh", "pred_points"):
        np.testing.assert_allclose(out[k].detach().numpy(), z[f"{name}/{k}"], rtol=1e-3, atol=2e-5, err_msg=k)
    losses = OC.bbox_criterion(out, {"points": pts, "whs": whs})
    for k in ("loss_wh", "loss_giou"):
        np.testing.assert_allclose(losses[k].item(), z[f"{name}/L_{k}"], rtol=1e-4, err_msg=k)
    (losses["loss_wh"] * 1 + losses["loss_giou"] * 0.4).backward()
    for n, r in zip(names, z[f"{name}/grad_norms"]):
        if r < 0:
            assert sd[n].grad is None, n
        else:
            np.testing.assert_allclose(sd[n].grad.norm().item(), r, rtol=5e-3, atol=1e-7, err_msg=n)


# ------------------------------------------------------------------------------------------------ full-size vectors (G10, G11)
@pytest.mark.parametrize("name", ["small_b2", "aux", "shipped576", "cfg2", "lvis_wide"])
def test_g10_full_size_reference_runs(golden, name):
    """The oracle vs the real reference at BASELINE's sizes (cfg2 = B=2 800x800 Q=300 T=(37,120), bench.py's batch), the shipped
    script's grid-576 shape, a padded batch and aux_loss=True: outputs, Hungarian indices (incl. every aux layer's), losses, total
    gradient norm and the digests of every intermediate (layer4, projection, 6 encoder outputs, 6 decoder states)."""
    from fullsize import TAP_KEYS, case_inputs, check_tap, rel_err
    from oracle.step import trainable_names
    z = golden("g10_full.npz")
    c = case_inputs(z, name)
    sd = seeded_state_dict(model_schema(num_position=c["nq"], spatial_prior=c["prior"]), heads="wide")
    names = trainable_names(sd)
    for n in names:
        sd[n].requires_grad_(True)
    for n in list(sd):
        for fam in ("cls_embed", "bbox_embed", "bbox_variance"):
            if f"transformer.{fam}.0." in n:
                for i in range(1, 6):
                    sd[n.replace(f"{fam}.0.", f"{fam}.{i}.")] = sd[n]
    taps = {}
    out, ref = OM.forward(c["images"], c["rects"], sd, spatial_prior=c["prior"], num_position=c["nq"], all_layers=c["aux"], taps=taps)
    for k in ("pred_logits", "pred_boxes", "pred_vars"):
        assert rel_err(out[k].detach().numpy(), z[f"{name}/{k}"]) < 1e-4, k
    np.testing.assert_allclose(ref.detach().numpy(), z[f"{name}/ref"], rtol=1e-6)
    for k in TAP_KEYS:
        check_tap(z, f"{name}/tap_{k}", taps[k], 2e-4)
    if c["aux"]:
        losses, all_idx = OC.set_criterion_aux(out, c["targets"])
        wd = OC.aux_weight_dict()
        for i in range(5):
            for b in range(c["B"]):
                assert np.array_equal(all_idx[i][b][0].numpy(), z[f"{name}/aux{i}/idx_i{b}"])
                assert np.array_equal(all_idx[i][b][1].numpy(), z[f"{name}/aux{i}/idx_j{b}"])
        idx = all_idx[-1]
    else:
        losses, idx = OC.set_criterion(out, c["targets"])
        wd = OC.WEIGHT_DICT
    for b in range(c["B"]):
        assert np.array_equal(idx[b][0].numpy(), z[f"{name}/idx_i{b}"]) and np.array_equal(idx[b][1].numpy(), z[f"{name}/idx_j{b}"])
    lkeys = [k[len(name) + 3:] for k in z.files if k.startswith(f"{name}/L_")]
    assert sorted(lkeys) == sorted(losses), (sorted(lkeys), sorted(losses))
    for k in lkeys:
        np.testing.assert_allclose(losses[k].item(), z[f"{name}/L_{k}"], rtol=2e-4, atol=1e-6, err_msg=k)
    total = OC.total_loss(losses, wd)
    np.testing.assert_allclose(total.item(), z[f"{name}/loss_total"], rtol=2e-4)
    total.backward()
    gn = torch.norm(torch.stack([sd[n].grad.norm() for n in names if sd[n].grad is not None]))
    np.testing.assert_allclose(gn.item(), z[f"{name}/grad_total_norm"], rtol=2e-3)
    # conditioning diagnostics the generator stored (round 6): the oracle's own outputs give the same distances from a tie / from the L1 kink
    i_, j_ = idx[0][0].numpy(), idx[0][1].numpy()
    if len(i_):
        marg = float((out["pred_boxes"][0][i_].detach() - c["targets"][0]["boxes"][j_]).abs().min())
        np.testing.assert_allclose(marg, z[f"{name}/min_l1_margin"][0], atol=5e-6, rtol=1e-2)
    # ... and, where the gradient is a well-posed demand (fullsize.grads_well_posed: no matched coordinate on the kink), the oracle's gradient
    # agrees with the reference's ELEMENT by element on the stored sample (the 4 largest-|gradient| elements of every parameter)
    from fullsize import grads_well_posed
    if grads_well_posed(z, name) and not c["aux"]:
        pnames = [str(n) for n in z[f"{name}/param_names"]]
        pidx, fidx, gref = z[f"{name}/sample_pidx"], z[f"{name}/sample_fidx"], z[f"{name}/sample_grad"].astype(np.float64)
        tn = float(z[f"{name}/grad_total_norm"])
        worst = 0.0
        for k in range(len(pidx)):
            g = sd[pnames[int(pidx[k])]].grad.reshape(-1)[int(fidx[k])].item()
            if abs(gref[k]) >= 1e-5 * tn:
                worst = max(worst, abs(g - gref[k]) / abs(gref[k]))
            else:
                assert abs(g - gref[k]) <= 1e-6 * tn
        assert worst <= 5e-3, f"{name}: oracle gradient element off by {worst:.2e}"
        print(f"{name}: oracle vs reference, worst sampled gradient element {worst:.2e}")


def test_g11_stage1_900_points(golden):
    """BASELINE config 5 at its stated size: the 1st-stage forward with 900 `defined` anchor points on an 800x800 image."""
    from fullsize import check_tap, rel_err
    from oracle.step import stage1_inputs
    from oracle.weights import stage1_schema
    z = golden("g11_stage1_n900.npz")
    H, W, n, seed = [int(v) for v in z["n900/cfg"]]
    img, pts, whs = stage1_inputs(H, W, n, seed)
    sd = seeded_state_dict(stage1_schema())
    with torch.no_grad():
        out = OM.forward_stage1(img, pts, sd)
    for k in ("pred_logits", "pred_wh", "pred_points"):
        assert rel_err(out[k].numpy(), z[f"n900/{k}"]) < 1e-4, k
    losses = OC.bbox_criterion(out, {"points": pts, "whs": whs})
    for k in ("loss_wh", "loss_giou"):
        np.testing.assert_allclose(losses[k].item(), z[f"n900/L_{k}"], rtol=1e-4, err_msg=k)
