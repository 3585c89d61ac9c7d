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
