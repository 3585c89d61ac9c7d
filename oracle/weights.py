"""State-dict schema of the 2nd-stage model + a name-seeded weight generator (TEST INFRA).

Schema restates what `AnchorDETR.state_dict()` contains (A2/models/anchor_detr.py:34-92,
A2/models/transformer.py:21-107, A2/models/resnet.py:163-280, A2/models/backbone.py:22-60);
SURVEY.md section 8(b) lists it.  Weights are never stored in fixtures: both the golden generator
(running the real reference) and the tests regenerate them from `seeded_state_dict`.
"""
import math
import zlib

import torch

RESNET50_LAYERS = (3, 4, 6, 3)


def _bn(prefix, c, out):
    for k in ("weight", "bias", "running_mean", "running_var"):
        out.append((f"{prefix}.{k}", (c,), "bn_" + k))


def backbone_schema():
    """resnet50 body up to layer4 (A2/models/resnet.py:163-280, Bottleneck :105-160)."""
    out = [("backbone.body.conv1.weight", (64, 3, 7, 7), "conv")]
    _bn("backbone.body.bn1", 64, out)
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), RESNET50_LAYERS), start=1):
        for b in range(blocks):
            p = f"backbone.body.layer{li}.{b}"
            width = planes
            out.append((f"{p}.conv1.weight", (width, inplanes, 1, 1), "conv"))
            _bn(f"{p}.bn1", width, out)
            out.append((f"{p}.conv2.weight", (width, width, 3, 3), "conv"))
            _bn(f"{p}.bn2", width, out)
            out.append((f"{p}.conv3.weight", (planes * 4, width, 1, 1), "conv"))
            _bn(f"{p}.bn3", planes * 4, out)
            if b == 0:
                out.append((f"{p}.downsample.0.weight", (planes * 4, inplanes, 1, 1), "conv"))
                _bn(f"{p}.downsample.1", planes * 4, out)
            inplanes = planes * 4
    return out


def _linear(prefix, nout, nin, out, kind="linear"):
    out.append((f"{prefix}.weight", (nout, nin), kind))
    out.append((f"{prefix}.bias", (nout,), "bias"))


def _ln(prefix, c, out):
    out.append((f"{prefix}.weight", (c,), "ln_weight"))
    out.append((f"{prefix}.bias", (c,), "ln_bias"))


def transformer_schema(d=256, dff=1024, enc=6, dec=6, num_pattern=1, num_position=300, spatial_prior="learned",
                       stage=2):
    out = []
    t = "transformer"
    for i in range(enc):
        p = f"{t}.encoder_layers.{i}"
        out.append((f"{p}.self_attn.in_proj_weight", (5 * d, d), "linear"))
        out.append((f"{p}.self_attn.in_proj_bias", (5 * d,), "bias"))
        _linear(f"{p}.self_attn.out_proj", d, d, out)
        _ln(f"{p}.norm1", d, out)
        _linear(f"{p}.ffn.linear1", dff, d, out)
        _linear(f"{p}.ffn.linear2", d, dff, out)
        _ln(f"{p}.ffn.norm2", d, out)
    for i in range(dec):
        p = f"{t}.decoder_layers.{i}"
        out.append((f"{p}.cross_attn.in_proj_weight", (5 * d, d), "linear"))
        out.append((f"{p}.cross_attn.in_proj_bias", (5 * d,), "bias"))
        _linear(f"{p}.cross_attn.out_proj", d, d, out)
        _ln(f"{p}.norm1", d, out)
        out.append((f"{p}.self_attn.in_proj_weight", (3 * d, d), "linear"))
        out.append((f"{p}.self_attn.in_proj_bias", (3 * d,), "bias"))
        _linear(f"{p}.self_attn.out_proj", d, d, out)
        _ln(f"{p}.norm2", d, out)
        _linear(f"{p}.ffn.linear1", dff, d, out)
        _linear(f"{p}.ffn.linear2", d, dff, out)
        _ln(f"{p}.ffn.norm2", d, out)
    out.append((f"{t}.pattern.weight" if stage == 2 else f"{t}.modify_pattern.weight", (num_pattern, d), "embed"))
    if spatial_prior == "learned":
        out.append((f"{t}.position.weight", (num_position, 2), "position"))
    for name in ("adapt_pos2d", "adapt_pos1d"):
        _linear(f"{t}.{name}.0", d, d, out)
        _linear(f"{t}.{name}.2", d, d, out)
    # heads: ONE module each, aliased `dec` times (A2/models/transformer.py:104-107) -> 6 identical copies
    for i in range(dec):
        out.append((f"{t}.cls_embed.{i}.weight", (2, d), "alias:cls_w"))
        out.append((f"{t}.cls_embed.{i}.bias", (2 if stage == 2 else 1,), "alias:cls_b"))   # A1/models/transformer.py:82-86
        for j, (no, ni) in enumerate(((d, d), (d, d), (4, d))):
            out.append((f"{t}.bbox_embed.{i}.layers.{j}.weight", (no, ni), f"alias:box_w{j}"))
            out.append((f"{t}.bbox_embed.{i}.layers.{j}.bias", (no,), f"alias:box_b{j}"))
        if stage == 2:
            for j, (no, ni) in enumerate(((d, d), (d, d), (2, d))):
                out.append((f"{t}.bbox_variance.{i}.layers.{j}.weight", (no, ni), f"alias:var_w{j}"))
                out.append((f"{t}.bbox_variance.{i}.layers.{j}.bias", (no,), f"alias:var_b{j}"))
    return out


def stage1_schema(num_pattern=1, spatial_prior="defined", enc=6, dec=6, d=256, dff=1024):
    """State dict of the 1st-stage model (A1/models/anchor_detr.py:34-79): backbone + input_proj + transformer (stage 1)."""
    out = backbone_schema()
    out += [("input_proj.0.0.weight", (d, 2048, 1, 1), "conv_xavier"), ("input_proj.0.0.bias", (d,), "bias"),
            ("input_proj.0.1.weight", (d,), "ln_weight"), ("input_proj.0.1.bias", (d,), "ln_bias")]
    out += transformer_schema(d, dff, enc, dec, num_pattern, 0, spatial_prior, stage=1)
    return out


def model_schema(num_position=300, num_pattern=1, spatial_prior="learned", enc=6, dec=6, d=256, dff=1024):
    out = backbone_schema()
    # input_proj: built but never used on the stage-2 path (A2/models/anchor_detr.py:68-74 vs :119)
    out += [("input_proj.0.0.weight", (d, 2048, 1, 1), "conv_xavier"), ("input_proj.0.0.bias", (d,), "zeros"),
            ("input_proj.0.1.weight", (d,), "ln_weight"), ("input_proj.0.1.bias", (d,), "ln_bias")]
    out += [("aggr_input_proj.0.0.weight", (d, 4096, 1, 1), "conv_xavier"), ("aggr_input_proj.0.0.bias", (d,), "bias"),
            ("aggr_input_proj.0.1.weight", (d,), "ln_weight"), ("aggr_input_proj.0.1.bias", (d,), "ln_bias")]
    out += transformer_schema(d, dff, enc, dec, num_pattern, num_position, spatial_prior)
    return out


def _gen(name):
    return torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)


def _make(name, shape, kind):
    g = _gen(name)
    r = lambda: torch.randn(shape, generator=g, dtype=torch.float32)  # noqa: E731
    if kind == "conv":  # kaiming-normal fan_out, like A2/models/resnet.py:231-233, slightly damped
        cout, _, kh, kw = shape
        return r() * math.sqrt(2.0 / (cout * kh * kw)) * 0.9
    if kind == "conv_xavier":
        cout, cin, kh, kw = shape
        return r() * math.sqrt(2.0 / ((cin + cout) * kh * kw))
    if kind == "bn_weight":
        return 1.0 + 0.1 * r()
    if kind == "bn_bias":
        return 0.1 * r()
    if kind == "bn_running_mean":
        return 0.1 * r()
    if kind == "bn_running_var":
        return 1.0 + 0.2 * torch.rand(shape, generator=g, dtype=torch.float32)
    if kind == "linear":
        no, ni = shape
        return r() * math.sqrt(2.0 / (no + ni))
    if kind == "bias":
        return 0.02 * r()
    if kind == "zeros":
        return torch.zeros(shape)
    if kind == "ln_weight":
        return 1.0 + 0.05 * r()
    if kind == "ln_bias":
        return 0.05 * r()
    if kind == "embed":
        return r()
    if kind == "position":
        return torch.rand(shape, generator=g, dtype=torch.float32)
    raise KeyError(kind)


def _make_alias(tag, shape, heads="narrow"):
    """Head weights (shared across decoder layers).  Non-degenerate but in the reference's spirit
    (A2/models/transformer.py:86-103): class bias -log(99); box size bias -2; variance head positive.
    heads="wide": last-layer weights of O(1/sqrt(d)) so that the outputs are dominated by the trunk's features, not by the
    biases -- an error anywhere in backbone / encoder / decoder then shows up at full size in logits, boxes and variances
    (with the "narrow" heads the golden logits sit within 0.6 of the bias and hide a 1 % trunk error)."""
    g = _gen("head:" + tag)
    r = lambda: torch.randn(shape, generator=g, dtype=torch.float32)  # noqa: E731
    if heads == "wide":
        if tag == "cls_w":
            return r() * 0.12
        if tag == "cls_b":
            return torch.tensor([-1.5, 1.0])[: shape[0]].clone()
        if tag == "box_w2":
            return r() * 0.0625
        if tag == "var_w2":
            return r() * 0.03
        if tag == "var_b2":
            return torch.full(shape, 1.0)
    if tag == "cls_w":
        return r() * 0.05
    if tag == "cls_b":
        return torch.full(shape, -math.log(99.0)) + 0.1 * torch.randn((2,), generator=g, dtype=torch.float32)[: shape[0]]
    if tag in ("box_w0", "box_w1", "var_w0", "var_w1"):
        return r() * math.sqrt(2.0 / (shape[0] + shape[1]))
    if tag in ("box_b0", "box_b1", "var_b0", "var_b1"):
        return 0.02 * r()
    if tag == "box_w2":
        return r() * 0.02
    if tag == "box_b2":
        return torch.tensor([0.0, 0.0, -2.0, -2.0])[: shape[0]] + 0.05 * r()
    if tag == "var_w2":
        return 0.01 * (1.0 + 0.2 * torch.rand(shape, generator=g, dtype=torch.float32))
    if tag == "var_b2":
        return torch.full(shape, 0.01)
    raise KeyError(tag)


def seeded_state_dict(schema=None, heads="narrow", **kw):
    """name -> fp32 CPU tensor; deterministic function of (name, shape, heads) only."""
    schema = schema if schema is not None else model_schema(**kw)
    sd = {}
    alias_cache = {}
    for name, shape, kind in schema:
        if kind.startswith("alias:"):
            tag = kind[6:]
            if tag not in alias_cache:
                alias_cache[tag] = _make_alias(tag, shape, heads)
            sd[name] = alias_cache[tag].clone()
        else:
            sd[name] = _make(name, shape, kind)
    return sd
