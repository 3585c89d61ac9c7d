"""Checkpoint interchange: the checkpoint dict main.py writes is the reference's
{"model", "optimizer", "lr_scheduler", "epoch", "args"} (A2/main.py:222-236), `model` is key-for-key the reference's
state dict, and --resume applies the reference's key filter (A2/main.py:195-209)."""
import torch


def test_state_dict_is_reference_schema_and_resume_filter(tmp_path):
    from counting_detr_amd import build_model
    from counting_detr_amd.args import default_args
    from oracle.weights import model_schema, seeded_state_dict
    args = default_args()
    args.device = "cpu"
    model, _, _ = build_model(args)
    names = [n for n, _, _ in model_schema()]
    sd = model.state_dict()
    assert sorted(sd.keys()) == sorted(names) and len(names) == 547      # the reference's AnchorDETR.state_dict() key set
    for n, shape, _ in model_schema():
        assert tuple(sd[n].shape) == tuple(shape), n
    ref = seeded_state_dict()
    ckpt = {"model": ref, "optimizer": {}, "lr_scheduler": {"last_epoch": 3, "step_size": 20, "gamma": 0.1}, "epoch": 3, "args": args}
    path = tmp_path / "detr_retrain.pth"
    torch.save(ckpt, path)
    loaded = torch.load(path, map_location="cpu", weights_only=False)
    # the resume filter of A2/main.py:199-201: keys that exist, except the query pattern embedding
    own = model.state_dict()
    before = own["transformer.pattern.weight"].clone()
    pre = {k: v for k, v in loaded["model"].items() if k in own and "transformer.pattern." not in k}
    missing, unexpected = model.load_state_dict(pre, strict=False)
    assert missing == ["transformer.pattern.weight"] and not unexpected
    after = model.state_dict()
    assert torch.equal(after["transformer.pattern.weight"], before)
    for k in pre:
        assert torch.equal(after[k].reshape(-1), ref[k].reshape(-1)), k
    # and a checkpoint written from this model loads strictly into a fresh one (what a reference user would do)
    torch.save({"model": model.state_dict()}, tmp_path / "out.pth")
    m2, _, _ = build_model(args)
    m2.load_state_dict(torch.load(tmp_path / "out.pth", weights_only=False)["model"], strict=True)


def _cpu_model(**kw):
    from counting_detr_amd import build_model
    from counting_detr_amd.args import default_args
    args = default_args(**kw)
    args.device = "cpu"
    model, crit, _ = build_model(args)
    return model, crit, args


def _torchvision_resnet50_state_dict(seed=0):
    """A synthetic state dict in torchvision's resnet50 key layout (what `resnet50-0676ba61.pth` holds): conv / bn / downsample
    keys of the four stages, `num_batches_tracked` counters and the `fc` head (A2/models/resnet.py:163-280)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def bn(p, c):
        sd[p + ".weight"] = torch.rand(c, generator=g) + 0.5
        sd[p + ".bias"] = torch.randn(c, generator=g) * 0.1
        sd[p + ".running_mean"] = torch.randn(c, generator=g) * 0.1
        sd[p + ".running_var"] = torch.rand(c, generator=g) + 0.5
        sd[p + ".num_batches_tracked"] = torch.tensor(0)

    sd["conv1.weight"] = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    bn("bn1", 64)
    inpl = 64
    for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), start=1):
        for b in range(nb):
            p = f"layer{li}.{b}"
            sd[p + ".conv1.weight"] = torch.randn(planes, inpl, 1, 1, generator=g) * 0.05
            bn(p + ".bn1", planes)
            sd[p + ".conv2.weight"] = torch.randn(planes, planes, 3, 3, generator=g) * 0.05
            bn(p + ".bn2", planes)
            sd[p + ".conv3.weight"] = torch.randn(planes * 4, planes, 1, 1, generator=g) * 0.05
            bn(p + ".bn3", planes * 4)
            if b == 0:
                sd[p + ".downsample.0.weight"] = torch.randn(planes * 4, inpl, 1, 1, generator=g) * 0.05
                bn(p + ".downsample.1", planes * 4)
            inpl = planes * 4
    sd["fc.weight"] = torch.randn(1000, 2048, generator=g) * 0.01
    sd["fc.bias"] = torch.zeros(1000)
    return sd


def test_torchvision_resnet50_layout_loads_into_the_backbone(tmp_path):
    """A2/models/backbone.py:153-155 -> resnet.py:292-297: the reference starts from the torchvision ImageNet weights."""
    import pytest
    from counting_detr_amd.checkpoint import load_backbone_pretrained
    model, _, _ = _cpu_model()
    tv = _torchvision_resnet50_state_dict()
    path = tmp_path / "resnet50-0676ba61.pth"
    torch.save(tv, path)
    n = load_backbone_pretrained(model, str(path))
    own = model.state_dict()
    assert n == len([k for k in tv if not k.startswith("fc.") and not k.endswith("num_batches_tracked")])
    for k, v in tv.items():
        if k.startswith("fc.") or k.endswith("num_batches_tracked"):
            assert "backbone.body." + k not in own
            continue
        assert torch.equal(own["backbone.body." + k], v), k
    w = model.backbone.body.layer2[0].conv2.weight        # storage stays channels_last (a filter tap = a contiguous K-run)
    assert w.is_contiguous(memory_format=torch.channels_last)
    bad = dict(tv)
    del bad["layer3.2.conv2.weight"]
    with pytest.raises(RuntimeError):                     # the reference's load is strict
        load_backbone_pretrained(model, bad)
    bad = dict(tv)
    bad["layer1.0.conv1.weight"] = torch.zeros(64, 32, 1, 1)
    with pytest.raises(RuntimeError):
        load_backbone_pretrained(model, bad)


def test_resume_from_an_anchor_detr_coco_shaped_checkpoint(tmp_path):
    """The shipped script fine-tunes from `AnchorDETR_r50_c5.pth` (A2/scripts/var_wh_laplace_600.sh:13, A2/main.py:195-209): a
    detector without aggr_input_proj / bbox_variance, with a 91-class head and learned 300 x 2 anchor positions, loaded into
    the grid-prior counting model."""
    import pytest
    from counting_detr_amd.checkpoint import resume_model
    model, _, _ = _cpu_model(spatial_prior="grid", num_query_position=600)
    own = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    coco = {}
    for k, v in own.items():
        if k.startswith("aggr_input_proj.") or "bbox_variance" in k:
            continue                                       # Anchor-DETR has neither
        if "cls_embed" in k:
            coco[k] = torch.randn((91,) + tuple(v.shape[1:]), generator=g)
        elif v.is_floating_point():
            coco[k] = torch.randn(v.shape, generator=g)
        else:
            coco[k] = v.clone()
    import re
    for k in list(coco):                                   # the heads are ONE module aliased six times: a real checkpoint holds identical copies
        m = re.match(r"(transformer\.(?:cls_embed|bbox_embed)\.)(\d)(\..*)", k)
        if m and m.group(2) != "0":
            coco[k] = coco[m.group(1) + "0" + m.group(3)].clone()
    coco["transformer.position.weight"] = torch.rand(300, 2, generator=g)      # learned prior there, grid prior here: not a key of ours
    coco["transformer.pattern.weight"] = torch.randn(3, 256, generator=g)       # filtered by name
    path = tmp_path / "AnchorDETR_r50_c5.pth"
    torch.save({"model": coco, "epoch": 49}, path)
    with pytest.raises(RuntimeError):                      # torch (and therefore the reference) refuses the 91-class head
        resume_model(model, str(path), log=lambda *_: None)
    model, _, _ = _cpu_model(spatial_prior="grid", num_query_position=600)
    own = {k: v.clone() for k, v in model.state_dict().items()}
    msgs = []
    ckpt, missing, skipped = resume_model(model, str(path), skip_mismatch=True, log=msgs.append)
    after = model.state_dict()
    assert ckpt["epoch"] == 49
    assert sorted(skipped) == sorted(k for k in own if "cls_embed" in k)
    assert all(k.startswith("aggr_input_proj.") or "bbox_variance" in k or "cls_embed" in k or k == "transformer.pattern.weight"
               for k in missing) and "transformer.pattern.weight" in missing
    for k in own:
        if k in missing:
            assert torch.equal(after[k], own[k]), k        # untouched
        else:
            assert torch.equal(after[k].reshape(-1), coco[k].reshape(-1)), k
    assert any("Missing Keys" in m for m in msgs) and any("Skipped" in m for m in msgs)


def test_optimizer_state_is_torch_adamw_layout_and_round_trips():
    """The checkpoint's "optimizer" / "lr_scheduler" entries use torch's own layouts with the reference's three parameter groups
    (A2/main.py:157-189,228-232): compared key-for-key with a real torch.optim.AdamW / StepLR over the same model."""
    from counting_detr_amd.engine import Trainer
    model, crit, args = _cpu_model()
    tr = Trainer(model, crit, args, device="cpu")
    tr.exp_avg.copy_(torch.arange(tr.exp_avg.numel(), dtype=torch.float32) * 1e-6)
    tr.exp_avg_sq.copy_(torch.arange(tr.exp_avg.numel(), dtype=torch.float32) * 1e-7)
    tr.opt_state[0] = 7.0
    tr.epoch = 21
    tr.opt_state[1] = 0.1
    sd = tr.state_dict()
    # the reference's construction
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [{"params": [p for n, p in named if "backbone" not in n], "lr": args.lr},
              {"params": [p for n, p in named if "backbone" in n], "lr": args.lr_backbone},
              {"params": [], "lr": args.lr * args.lr_linear_proj_mult}]
    opt = torch.optim.AdamW(groups, lr=args.lr, weight_decay=args.weight_decay)
    sched = torch.optim.lr_scheduler.StepLR(opt, args.lr_drop)
    ref = opt.state_dict()
    assert [g["params"] for g in sd["param_groups"]] == [g["params"] for g in ref["param_groups"]]
    for a, b in zip(sd["param_groups"], ref["param_groups"]):
        for k in ("betas", "eps", "weight_decay", "amsgrad"):
            assert a[k] == b[k], k
        assert abs(a["initial_lr"] - b["initial_lr"]) < 1e-12 and abs(a["lr"] - 0.1 * b["lr"]) < 1e-12
    flat = [p for g in groups for p in g["params"]]
    no_grad_names = {n for n, _ in model.named_parameters() if n.startswith("input_proj.")}
    names = [n for g in ([n for n, _ in named if "backbone" not in n], [n for n, _ in named if "backbone" in n]) for n in g]
    for i, (n, p) in enumerate(zip(names, flat)):
        if n in no_grad_names:
            assert i not in sd["state"]                   # torch keeps no state for a parameter that never had a gradient
        else:
            st = sd["state"][i]
            assert tuple(st["exp_avg"].shape) == tuple(p.shape) and tuple(st["exp_avg_sq"].shape) == tuple(p.shape)
            assert float(st["step"]) == 7.0
    opt.load_state_dict(sd)                                # torch itself accepts it
    assert set(tr.lr_scheduler_state_dict()) >= set(sched.state_dict()) - {"_is_initial"}
    # round trip into a fresh trainer
    model2, crit2, args2 = _cpu_model()
    tr2 = Trainer(model2, crit2, args2, device="cpu")
    tr2.load_state_dict(sd, tr.lr_scheduler_state_dict())
    assert torch.equal(tr2.exp_avg, tr.exp_avg) and torch.equal(tr2.exp_avg_sq, tr.exp_avg_sq)
    assert float(tr2.opt_state[0]) == 7.0 and tr2.epoch == 21 and abs(float(tr2.opt_state[1]) - 0.1) < 1e-7


def test_count_rule_and_metrics_match_the_golden_vector(golden):
    """SURVEY a14 on the PRODUCT's functions: A2/infer.py:75-81 counting rule and A2/eval_all.py:252-270 metrics vs G8."""
    import numpy as np
    from counting_detr_amd.engine import count_from_logits, counting_metrics
    z = golden("g8_count.npz")
    counts, keep, prob = count_from_logits(torch.from_numpy(z["logits"]))
    assert np.array_equal(counts.numpy(), z["counts"])
    assert bool(keep[0, 0]) and float(prob[0, 0]) == 0.5  # exactly 0.5 counts (>=)
    m = counting_metrics(counts.tolist(), z["gt"].tolist())
    np.testing.assert_allclose([m["MAE"], m["RMSE"], m["NAE"], m["SRE"]], z["metrics"], rtol=1e-12)
