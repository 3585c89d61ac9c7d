"""End-to-end parity of the MI355X path (model + criterion + step) against golden vectors captured from the real
reference (tests/golden/g6_e2e.npz) -- needs an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(params=[0, 1], ids=["fp32mfma", "bf16x3"])
def precision(request):
    from counting_detr_amd import ops
    old = ops.PRECISION
    ops.PRECISION = request.param
    yield request.param
    ops.PRECISION = old


def T(a):
    return torch.from_numpy(np.asarray(a))


def build(nq=300, prior="learned"):
    import counting_detr_amd
    from counting_detr_amd.args import default_args
    from oracle.weights import model_schema, seeded_state_dict
    args = default_args(device=DEV, num_query_position=nq, spatial_prior=prior)
    model, crit, _ = counting_detr_amd.build_model(args)
    model.load_state_dict(seeded_state_dict(model_schema(num_position=nq, spatial_prior=prior)), strict=True)
    return model.to(DEV), crit, args


def load_case(z, name):
    B, nq, is_grid = [int(v) for v in z[f"{name}/cfg"]]
    imgs = [T(z[f"{name}/img{i}"]).to(DEV) for i in range(B)]
    rects = T(z[f"{name}/rects"]).to(DEV)
    tg = []
    for b in range(B):
        bx = T(z[f"{name}/tgt{b}"]).reshape(-1, 4).to(DEV)
        tg.append({"boxes": bx, "labels": torch.zeros(bx.shape[0], dtype=torch.int64, device=DEV)})
    return B, nq, ("grid" if is_grid else "learned"), imgs, rects, tg


@pytest.mark.parametrize("name", ["b1_64x96", "b2_pad", "b1_grid20"])
def test_forward_losses_grads_vs_reference(golden, name, precision):
    z = golden("g6_e2e.npz")
    B, nq, prior, imgs, rects, tg = load_case(z, name)
    model, crit, args = build(nq, prior)
    model.train()
    samples = torch.stack(imgs) if B == 1 else imgs
    out, ref = model(samples, rects=rects)
    for k in ("pred_logits", "pred_boxes", "pred_vars"):      # north_star: within 1e-3 rel of the fp32 reference
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), z[f"{name}/{k}"], rtol=1e-3, atol=1e-4, err_msg=k)
    np.testing.assert_allclose(ref.detach().cpu().numpy(), z[f"{name}/ref"], rtol=1e-6)
    idx = crit.matcher(out, tg)                                 # bit-exact Hungarian indices
    for b in range(B):
        assert np.array_equal(idx[b][0].numpy(), z[f"{name}/idx_i{b}"])
        assert np.array_equal(idx[b][1].numpy(), z[f"{name}/idx_j{b}"])
    losses = crit(out, tg)
    for k in ("loss_ce", "loss_bbox", "loss_giou", "cardinality_error", "loss_variance", "class_error"):
        np.testing.assert_allclose(float(losses[k]), z[f"{name}/L_{k}"], rtol=1e-3, atol=1e-5, err_msg=k)
    total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
    total.backward()
    names = [str(n) for n in z[f"{name}/param_names"]]
    params = dict(model.named_parameters())
    grads = [p.grad for p in params.values() if p.grad is not None]
    tn = torch.norm(torch.stack([g.norm() for g in grads])).item()
    np.testing.assert_allclose(tn, z[f"{name}/grad_total_norm"], rtol=2e-3)
    coef = min(1.0, 0.1 / (tn + 1e-6))
    for n, r in zip(names, z[f"{name}/grad_norms_clipped"]):
        p = params[n]
        if r < 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
        else:
            np.testing.assert_allclose(p.grad.norm().item() * coef, r, rtol=1e-2, atol=1e-6, err_msg=n)


def test_train_step_matches_reference_adamw(golden, precision):
    """Trainer (flat arenas, device matcher, flat clip + AdamW) reproduces the reference's post-step parameters."""
    from counting_detr_amd.engine import Trainer
    z = golden("g6_e2e.npz")
    name = "b1_64x96"
    B, nq, prior, imgs, rects, tg = load_case(z, name)
    model, crit, args = build(nq, prior)
    model.train()
    tr = Trainer(model, crit, args, device=DEV)
    out = tr.train_step(torch.stack(imgs), rects, tg)
    np.testing.assert_allclose(float(out["grad_norm"]), z[f"{name}/grad_total_norm"], rtol=2e-3)
    names = [str(n) for n in z[f"{name}/param_names"]]
    sums = z[f"{name}/param_sums_after_step"]
    params = dict(model.named_parameters())
    for n, s in zip(names, sums):
        np.testing.assert_allclose(params[n].detach().double().sum().item(), s, rtol=1e-4, atol=5e-3, err_msg=n)


def test_graph_replay_equals_eager():
    """The whole step captured in a HIP graph gives the same losses as the eager step (same weights, same batch)."""
    from counting_detr_amd.engine import Trainer
    from oracle.step import synthetic_batch
    images, rects, targets = synthetic_batch(B=2, H=128, W=160, Ts=(7, 13))
    images, rects = images.to(DEV), rects.to(DEV)
    targets = [{k: v.to(DEV) for k, v in t.items()} for t in targets]
    res = []
    for use_graph in (False, True):
        model, crit, args = build()
        model.train()
        tr = Trainer(model, crit, args, device=DEV)
        if use_graph:
            tr.capture(images, rects, targets, warmup=0)            # records the step (weights untouched) ...
            out = tr.replay()                                       # ... and this executes it once
        else:
            out = tr.train_step(images, rects, targets)
        res.append({k: float(v) for k, v in out.items()})
    for k in res[0]:
        np.testing.assert_allclose(res[1][k], res[0][k], rtol=1e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("name", ["n3", "n57"])
def test_stage1_vs_reference(golden, name, precision):
    """1st-stage model on the HIP kernels vs golden vectors of the real 1st-stage reference."""
    from counting_detr_amd import stage1
    from counting_detr_amd.args import default_args
    from oracle.weights import seeded_state_dict, stage1_schema
    z = golden("g7_stage1.npz")
    args = default_args(device=DEV, spatial_prior="defined")
    model, crit, _ = stage1.build(args)
    model.load_state_dict(seeded_state_dict(stage1_schema()), strict=True)
    model.to(DEV).train()
    pts, whs = T(z[f"{name}/points"]).to(DEV), T(z[f"{name}/whs"]).to(DEV)
    out = model(T(z[f"{name}/img"]).to(DEV), pts)
    for k in ("pred_logits", "pred_wh", "pred_points"):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), z[f"{name}/{k}"], rtol=1e-3, atol=1e-4, err_msg=k)
    losses = crit(out, {"points": pts, "whs": whs})
    for k in ("loss_wh", "loss_giou"):
        np.testing.assert_allclose(float(losses[k]), z[f"{name}/L_{k}"], rtol=1e-3, err_msg=k)
    sum(losses[k] * crit.weight_dict[k] for k in losses).backward()
    params = dict(model.named_parameters())
    for n, r in zip([str(x) for x in z[f"{name}/param_names"]], z[f"{name}/grad_norms"]):
        if r >= 0:
            np.testing.assert_allclose(params[n].grad.norm().item(), r, rtol=1e-2, atol=1e-6, err_msg=n)


def test_dataset_train_epoch_and_infer(tmp_path):
    """SURVEY 8f rows 1-2 end to end on the tiny FSC-147-shaped dataset: DataLoader + collate + Prefetcher feed the trainer
    (a padded two-image batch), then infer.py writes the reference's prediction json and its counts agree with the model."""
    import argparse, json, os
    from torch.utils.data import DataLoader
    from counting_detr_amd import build_model, data
    from counting_detr_amd.args import default_args
    from counting_detr_amd.engine import Trainer, train_one_epoch
    from counting_detr_amd.misc import NestedTensor
    from oracle.weights import seeded_state_dict
    import infer as infer_mod
    here = os.path.dirname(os.path.abspath(__file__))
    args = default_args()
    args.data_path, args.scale_factor = os.path.join(here, "golden", "fsc147_tiny"), 32
    model, criterion, _ = build_model(args)
    model.load_state_dict(seeded_state_dict(), strict=True)
    model.to(DEV); criterion.to(DEV)
    trainer = Trainer(model, criterion, args, device=DEV)
    dl = DataLoader(data.build_dataset(args), batch_size=2, shuffle=False, collate_fn=data.collate)
    logs = []
    stats = train_one_epoch(trainer, data.Prefetcher(dl, DEV), 0, print_freq=1, log=logs.append)
    assert np.isfinite(stats["loss"]) and stats["loss"] > 0 and np.isfinite(stats["grad_norm"])
    # inference on the val split: json wire format + counts
    vl = DataLoader(data.build_test_dataset(args, "val"), batch_size=1, shuffle=False, collate_fn=data.collate)
    metrics, pred = infer_mod.infer(model, criterion, vl, torch.device(DEV), str(tmp_path), split="val")
    assert metrics["images"] == 2 and os.path.isfile(tmp_path / "predictions_val.json")
    pj = json.load(open(tmp_path / "predictions_val.json"))
    assert pj["categories"] == [{"name": "fg", "id": 1}] and len(pj["images"]) == 2
    for a in pj["annotations"]:
        assert set(a) == {"id", "image_id", "area", "bbox", "category_id", "score", "point"} and a["score"] >= 0.5
        assert all(isinstance(v, int) for v in a["bbox"] + a["point"])
    # the json-level evaluator reproduces the counts the model produced
    m2 = infer_mod.counting_metrics_from_json(str(tmp_path / "predictions_val.json"), os.path.join(args.data_path, "instances_val.json"))
    for k in ("MAE", "RMSE", "NAE", "SRE"):
        np.testing.assert_allclose(m2[k], metrics[k], rtol=1e-12)
    model.eval()
    with torch.no_grad():
        b = next(iter(vl))
        out, _ = model(NestedTensor(b["image"].to(DEV), b["mask"].to(DEV)), rects=b["ex_rects"].to(DEV))
    n0 = int((out["pred_logits"].sigmoid()[0, :, 0] >= 0.5).sum())
    assert n0 == sum(1 for a in pj["annotations"] if a["image_id"] == int(b["image_id"][0]))


def test_stage1_to_stage2_handoff(tmp_path):
    """the 1st-stage model writes pseudo_bbox_<split>.json (A1/engine.py:124-187 format) and the 2nd-stage
    training reader consumes that very file."""
    import argparse, json, os, shutil
    from PIL import Image
    from counting_detr_amd import data, stage1
    from counting_detr_amd.args import default_args
    from oracle.weights import seeded_state_dict, stage1_schema
    args = default_args()
    args.spatial_prior, args.num_query_pattern = "defined", 1
    model, _, _ = stage1.build(args)
    model.load_state_dict(seeded_state_dict(stage1_schema()), strict=True)
    model.to(DEV)
    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(here, "golden", "fsc147_tiny")
    root = tmp_path / "ds"
    shutil.copytree(src, root)
    names = {7: "1.png", 9: "2.png"}
    anno = json.load(open(root / "annotation_FSC147_384.json"))
    samples = []
    for im_id, fn in names.items():
        img = Image.open(root / "images_384_VarV2" / fn)
        w, h = img.size
        shutil.copy(root / "images_384_VarV2" / fn, root / "images_384_VarV2" / f"{im_id}.jpg")     # the hand-off names files <im_id>.jpg
        anno[f"{im_id}.jpg"] = anno[fn]
        pts = torch.tensor(anno[fn]["points"], dtype=torch.float32) / torch.tensor([w, h], dtype=torch.float32)
        t = data.to_normalized_tensor(img.resize((32 * (w // 32), 32 * (h // 32))))
        samples.append({"image": t[None], "points": pts[None], "orig_size": torch.tensor([[w, h]]), "im_id": torch.tensor(im_id)})
    json.dump(anno, open(root / "annotation_FSC147_384.json", "w"))
    ann = stage1.write_pseudo_labels(model, samples, "train", str(root / "annotations"), device=DEV)
    assert [im["file_name"] for im in ann["images"]] == ["7.jpg", "9.jpg"] and [im["id"] for im in ann["images"]] == [1, 2]
    assert len(ann["annotations"]) == sum(len(anno[fn]["points"]) for fn in names.values())
    for a in ann["annotations"]:
        assert set(a) == {"id", "image_id", "area", "bbox", "category_id", "iscrowd"} and all(isinstance(v, int) for v in a["bbox"])
        assert a["bbox"][2] >= 0 and a["bbox"][3] >= 0
    # the 2nd-stage reader opens the file the 1st stage wrote
    ds = data.FSC147Dataset(argparse.Namespace(data_path=str(root)), split="train")
    assert len(ds) == 2
    s0 = ds[0]
    n0 = len(anno["1.png"]["points"])
    assert s0["boxes"].shape == (n0, 4) and s0["ex_rects"].shape == (3, 4)
    w0, h0 = Image.open(root / "images_384_VarV2" / "7.jpg").size
    want = np.array([a["bbox"] for a in ann["annotations"] if a["image_id"] == 1], dtype=np.float32) / np.array([w0, h0, w0, h0], dtype=np.float32)
    np.testing.assert_allclose(s0["boxes"], want, rtol=0, atol=1e-7)
