"""Checkpoint interchange with the reference.

Three layouts a reference user holds:
  * the torchvision ResNet-50 file the reference loads into its backbone at construction
    (`pretrained_models/resnet50-0676ba61.pth`, A2/models/resnet.py:292-297 via A2/models/backbone.py:153-155): flat
    torchvision keys `conv1.weight`, `bn1.running_mean`, `layer3.2.conv2.weight`, `fc.weight`, ...;
  * a detector checkpoint `{"model": state_dict, ...}` passed to `--resume` (A2/main.py:195-209), e.g. the Anchor-DETR COCO
    model `AnchorDETR_r50_c5.pth` the shipped script starts from (A2/scripts/var_wh_laplace_600.sh:13): the reference keeps the
    keys that exist in its model except `transformer.pattern.*` and loads them with strict=False;
  * the checkpoints this build's main.py writes: {"model", "optimizer", "lr_scheduler", "epoch", "args"} with the reference's
    547-key model state dict and torch's own AdamW / StepLR state layouts (engine.Trainer.state_dict).
"""
import torch

BODY_PREFIX = "backbone.body."


def invalidate_caches(model):
    """Drop every device table derived from parameter VALUES: the FrozenBN folds (`_cache`) and the packed stem images
    (`_stem_w4` / `_stem_wr`).  Call after anything that overwrites weights outside the optimizer step (checkpoint loads,
    replica broadcast) -- a forward that ran before would otherwise keep using images of the old values."""
    for m in model.modules():
        if hasattr(m, "_cache"):
            m._cache = None
        if hasattr(m, "_stem_w4"):
            m._stem_w4 = None
        if hasattr(m, "_stem_wr"):
            m._stem_wr = None
    # captured steps / forwards hold the addresses of those tables: their owners (engine.Trainer, engine.InferenceEngine) drop them
    for ref in list(model.__dict__.get("_graph_cache_owners", [])):
        owner = ref()
        if owner is None:
            model.__dict__["_graph_cache_owners"].remove(ref)
        else:
            owner.clear_graph_cache()


def _read(path_or_dict):
    if isinstance(path_or_dict, dict):
        return path_or_dict
    return torch.load(path_or_dict, map_location="cpu", weights_only=False)


def load_backbone_pretrained(model, path_or_state_dict):
    """Load a torchvision-layout ResNet-50 state dict into `model.backbone.body` (A2/models/resnet.py:292-297).
    The reference loads it strictly into a full torchvision ResNet (fc included) and then keeps the layers up to layer4
    (IntermediateLayerGetter, A2/models/backbone.py:101-104): here `fc.*` and `num_batches_tracked` are dropped and every other
    key must exist with the right shape -- a missing / unexpected / mis-shaped key raises, like the reference's strict load.
    Conv weights are copied into this build's channels_last storage (same logical [Cout,Cin,kh,kw] values)."""
    sd = _read(path_or_state_dict)
    if "model" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]
    body = model.backbone.body
    own = body.state_dict()
    src = {k: v for k, v in sd.items() if not k.startswith("fc.") and not k.endswith("num_batches_tracked")}
    missing = sorted(set(own) - set(src))
    unexpected = sorted(set(src) - set(own))
    if missing or unexpected:
        raise RuntimeError(f"torchvision ResNet-50 state dict does not match the backbone: missing {missing[:5]}"
                           f"{'...' if len(missing) > 5 else ''}, unexpected {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
    with torch.no_grad():
        for k, t in own.items():
            if tuple(t.shape) != tuple(src[k].shape):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(src[k].shape)} vs model {tuple(t.shape)}")
            t.copy_(src[k].to(t.device, t.dtype))        # copy_ honours the destination's (channels_last) strides
    invalidate_caches(model)                              # cached FrozenBN folds / padded stem images are stale now
    return len(own)


def resume_model(model, path_or_ckpt, skip_mismatch=False, log=print):
    """`--resume` exactly as A2/main.py:195-209: keep the checkpoint's keys that exist in the model except
    `transformer.pattern.*`, load with strict=False, report missing / unexpected keys.  A key whose SHAPE differs (a COCO
    class head [91,256], another `position` count) makes torch -- hence the reference -- raise; `skip_mismatch=True`
    (`--resume_skip_mismatch`) drops such keys instead and reports them.  Returns (checkpoint dict, missing, skipped)."""
    ckpt = _read(path_or_ckpt)
    pretrained = ckpt["model"] if "model" in ckpt else ckpt
    own = model.state_dict()
    keep = {k: v for k, v in pretrained.items() if k in own and "transformer.pattern." not in k}
    skipped = []
    if skip_mismatch:
        skipped = sorted(k for k, v in keep.items() if tuple(v.shape) != tuple(own[k].shape))
        for k in skipped:
            del keep[k]
    missing, unexpected = model.load_state_dict(keep, strict=False)
    unexpected = [k for k in unexpected if not (k.endswith("total_params") or k.endswith("total_ops"))]
    if missing:
        log("Missing Keys: {}".format(missing))
    if unexpected:
        log("Unexpected Keys: {}".format(unexpected))
    if skipped:
        log("Skipped (shape mismatch): {}".format(skipped))
    invalidate_caches(model)
    return ckpt, list(missing), skipped
