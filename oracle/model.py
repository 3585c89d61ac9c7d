"""CPU restatement of the Counting-DETR 2nd-stage forward (TEST INFRA, see oracle/__init__.py).

Functional style over a flat state dict `sd` (name -> tensor; schema in oracle/weights.py).  Plain
torch ops in the reference's op order so fp32 results agree with the imported reference to ~1e-6;
pass float64 tensors for a high-precision checker.  Citations: A2/ = src/CountDETR_147_2nd_stage/.
"""
import math

import torch
import torch.nn.functional as F

from .weights import RESNET50_LAYERS


# ----------------------------------------------------------------------------- backbone (a1)
def frozen_bn(x, sd, p):
    """A2/models/backbone.py:50-60 -- fixed affine, eps inside rsqrt."""
    w, b = sd[p + ".weight"], sd[p + ".bias"]
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    scale = w * (rv + 1e-5).rsqrt()
    bias = b - rm * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def bottleneck(x, sd, p, stride, dilation, has_down):
    """A2/models/resnet.py:140-160 (stride on the 3x3, v1.5)."""
    out = F.relu(frozen_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
    out = F.conv2d(out, sd[p + ".conv2.weight"], stride=stride, padding=dilation, dilation=dilation)
    out = F.relu(frozen_bn(out, sd, p + ".bn2"))
    out = frozen_bn(F.conv2d(out, sd[p + ".conv3.weight"]), sd, p + ".bn3")
    if has_down:
        x = frozen_bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), sd, p + ".downsample.1")
    return F.relu(out + x)


def resnet50_dc5(x, sd, dilation=True, p="backbone.body"):
    """A2/models/resnet.py:261-271 up to layer4; layer4 stride replaced by dilation
    (A2/models/backbone.py:153-155, resnet.py:217-258: first block keeps the previous dilation)."""
    x = F.relu(frozen_bn(F.conv2d(x, sd[p + ".conv1.weight"], stride=2, padding=3), sd, p + ".bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    cur_dil = 1
    for li, blocks in enumerate(RESNET50_LAYERS, start=1):
        stride = 1 if li == 1 else 2
        prev_dil = cur_dil
        if li == 4 and dilation:
            cur_dil *= stride
            stride = 1
        for b in range(blocks):
            x = bottleneck(x, sd, f"{p}.layer{li}.{b}", stride if b == 0 else 1, prev_dil if b == 0 else cur_dil,
                           has_down=(b == 0))
    return x


def exemplar_centres(rects0, h, w):
    """A2/models/backbone.py:122-127: truncating int() of the box centre in feature cells."""
    out = []
    for rect in rects0.to(torch.float32):       # fp32 tensor arithmetic, like the reference
        x1, y1, x2, y2 = rect
        nx1, ny1, nx2, ny2 = x1 * w, y1 * h, x2 * w, y2 * h
        out.append((int((ny1 + ny2) / 2), int((nx1 + nx2) / 2)))
    return out


def extract_feature(images, mask, rects, sd, dilation=True, exemplar_mode="reference"):
    """A2/models/backbone.py:116-145.  exemplar_mode "reference": only `rects[0]` (image 0's exemplars) is used for the whole
    batch, scaled by the PADDED feature-map size -- what the reference does (it only ever runs batch 1, where this is exact).
    exemplar_mode "per_image" (the batched trainer's rule): image b is conditioned on ITS exemplars `rects[b]`, whose
    normalised coordinates are scaled by that image's own un-padded extent in feature cells (rows / columns of the
    down-sampled padding mask that are not padding); rows with x2 < 0 are absent exemplars (FSCD-LVIS has "at most 3").
    Identical to "reference" for a batch of one un-padded image."""
    x = resnet50_dc5(images, sd, dilation)
    h, w = x.shape[-2:]
    m = F.interpolate(mask[None].float(), size=(h, w)).to(torch.bool)[0]
    if exemplar_mode == "reference":
        pfs = [x[:, :, yc, xc][:, :, None, None] for (yc, xc) in exemplar_centres(rects[0], h, w)]
        pf = torch.stack(pfs).mean(0)
    else:
        assert exemplar_mode == "per_image"
        rows = []
        for b in range(x.shape[0]):
            hv, wv = int((~m[b, :, 0]).sum()), int((~m[b, 0, :]).sum())
            rb = rects[b][rects[b][:, 2] >= 0]
            cs = exemplar_centres(rb, hv, wv)
            acc = x[b, :, cs[0][0], cs[0][1]]
            for (yc, xc) in cs[1:]:
                acc = acc + x[b, :, yc, xc]
            rows.append(acc / len(cs))
        pf = torch.stack(rows)[:, :, None, None]
    feat = torch.cat([x, x * pf], dim=1)
    return feat, m


def aggr_input_proj(feat, sd, p="aggr_input_proj.0"):
    """A2/models/anchor_detr.py:78-84,119: 1x1 conv 4096->256 + GroupNorm(32, 256)."""
    y = F.conv2d(feat, sd[p + ".0.weight"], sd[p + ".0.bias"])
    return F.group_norm(y, 32, sd[p + ".1.weight"], sd[p + ".1.bias"], eps=1e-5)


# ----------------------------------------------------------------------------- positional (a3)
def mask2pos(mask):
    """A2/models/transformer.py:497-503."""
    nm = ~mask
    y = nm[:, :, 0].cumsum(1, dtype=torch.float32)
    x = nm[:, 0, :].cumsum(1, dtype=torch.float32)
    return (y - 0.5) / y[:, -1:], (x - 0.5) / x[:, -1:]


def _sine(pos, nfeat, temperature=10000):
    dim_t = torch.arange(nfeat, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / nfeat)
    px = (pos * (2 * math.pi))[..., None] / dim_t.to(pos.dtype)
    return torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=-1).flatten(-2)


def pos2posemb1d(pos):
    """A2/models/transformer.py:487-494."""
    return _sine(pos, 256)


def pos2posemb2d(pos):
    """A2/models/transformer.py:474-484 -- (y, x) concatenation order."""
    return torch.cat((_sine(pos[..., 1], 128), _sine(pos[..., 0], 128)), dim=-1)


def inverse_sigmoid(x, eps=1e-5):
    """A2/util/misc.py:475-479."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def mlp2(x, sd, p):
    """nn.Sequential(Linear, ReLU, Linear)  (A2/models/transformer.py:73-74)."""
    return F.linear(F.relu(F.linear(x, sd[p + ".0.weight"], sd[p + ".0.bias"])), sd[p + ".2.weight"], sd[p + ".2.bias"])


def mlp3(x, sd, p):
    """MLP(.., 3)  (A2/models/transformer.py:429-439)."""
    for j in range(3):
        x = F.linear(x, sd[f"{p}.layers.{j}.weight"], sd[f"{p}.layers.{j}.bias"])
        if j < 2:
            x = F.relu(x)
    return x


def layer_norm(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


# ----------------------------------------------------------------------------- RCDA (a5)
def rcda(q_row_in, q_col_in, k_row_in, k_col_in, v_in, sd, p, mask=None, nh=8, return_attn=False):
    """A2/models/row_column_decoupled_attention.py:24-321 in the reference's op order
    (project -> mean -> scale -> logits -> mask -> softmax -> short-edge-first contraction -> out_proj).
    q_*: [N,L,E]; k_*, v: [N,H,W,E]; mask: bool [N,H,W].  Returns [L,N,E]."""
    N, L, E = q_row_in.shape
    H, W = v_in.shape[1:3]
    d = E // nh
    Wi, bi = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    lin = lambda x, i: F.linear(x, Wi[i * E:(i + 1) * E], bi[i * E:(i + 1) * E])  # noqa: E731
    q_row, q_col = lin(q_row_in, 0), lin(q_col_in, 1)                 # :165-181
    k_row, k_col, v = lin(k_row_in, 2), lin(k_col_in, 3), lin(v_in, 4)  # :183-208
    k_row = k_row.mean(1)                                             # [N,W,E]  :212 (unmasked mean)
    k_col = k_col.mean(2)                                             # [N,H,E]  :213
    q_row = q_row * (float(d) ** -0.5)                                # :215-216
    q_col = q_col * (float(d) ** -0.5)
    hs = lambda t: t.reshape(N, -1, nh, d).permute(0, 2, 1, 3)        # noqa: E731  [N,nh,*,d]
    qr, qc, kr, kc = hs(q_row), hs(q_col), hs(k_row), hs(k_col)
    vv = v.reshape(N, H, W, nh, d).permute(0, 3, 1, 2, 4)             # [N,nh,H,W,d]
    s_row = qr @ kr.transpose(-1, -2)                                 # [N,nh,L,W]  :233
    s_col = qc @ kc.transpose(-1, -2)                                 # [N,nh,L,H]  :234
    if mask is not None:                                              # :238-249 first row / first column rule
        s_row = s_row.masked_fill(mask[:, 0, :][:, None, None, :], float("-inf"))
        s_col = s_col.masked_fill(mask[:, :, 0][:, None, None, :], float("-inf"))
    a_col = s_col.softmax(-1)
    a_row = s_row.softmax(-1)
    if H < W:                                                         # :261-276
        t = torch.einsum("bnqw,bnhwc->bnqhc", a_row, vv)
        o = torch.einsum("bnqh,bnqhc->bnqc", a_col, t)
    else:                                                             # :277-294
        t = torch.einsum("bnqh,bnhwc->bnqwc", a_col, vv)
        o = torch.einsum("bnqw,bnqwc->bnqc", a_row, t)
    o = o.permute(2, 0, 1, 3).reshape(L, N, E)
    out = F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])  # :311
    if return_attn:
        return out, a_row, a_col
    return out


def mha_self(q_in, k_in, v_in, sd, p, nh=8):
    """torch.nn.MultiheadAttention forward as used at A2/models/transformer.py:337,369-370.
    Inputs [N,L,E] (batch-first here); returns [N,L,E]."""
    N, L, E = q_in.shape
    d = E // nh
    Wi, bi = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(q_in, Wi[:E], bi[:E])
    k = F.linear(k_in, Wi[E:2 * E], bi[E:2 * E])
    v = F.linear(v_in, Wi[2 * E:], bi[2 * E:])
    hs = lambda t: t.reshape(N, L, nh, d).permute(0, 2, 1, 3)  # noqa: E731
    a = ((hs(q) * (float(d) ** -0.5)) @ hs(k).transpose(-1, -2)).softmax(-1)
    o = (a @ hs(v)).permute(0, 2, 1, 3).reshape(N, L, E)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


# ----------------------------------------------------------------------------- encoder / decoder (a4, a6, a7)
def ffn(x, sd, p):
    """A2/models/transformer.py:412-426 (post-norm, dropout 0)."""
    y = F.linear(F.relu(F.linear(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                 sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return layer_norm(x + y, sd, p + ".norm2")


def encoder_layer(src, mask, posemb_row, posemb_col, sd, p):
    """A2/models/transformer.py:242-279.  src [N,H,W,C] (NHWC here)."""
    N, H, W, C = src.shape
    kr = src + posemb_row[:, None]          # broadcast over h
    kc = src + posemb_col[:, :, None]       # broadcast over w
    a = rcda(kr.reshape(N, H * W, C), kc.reshape(N, H * W, C), kr, kc, src, sd, p + ".self_attn", mask)
    src = layer_norm(src + a.transpose(0, 1).reshape(N, H, W, C), sd, p + ".norm1")
    return ffn(src, sd, p + ".ffn")


def decoder_layer(tgt, ref, memory, mask, posemb_row, posemb_col, sd, p):
    """A2/models/transformer.py:352-409.  tgt [N,Q,C]; memory [N,H,W,C]."""
    t = "transformer"
    query_pos = mlp2(pos2posemb2d(ref), sd, t + ".adapt_pos2d")                    # :366-367
    qk = tgt + query_pos
    tgt = layer_norm(tgt + mha_self(qk, qk, tgt, sd, p + ".self_attn"), sd, p + ".norm2")   # :369-372
    qpx = mlp2(pos2posemb1d(ref[..., 0]), sd, t + ".adapt_pos1d")                  # :378
    qpy = mlp2(pos2posemb1d(ref[..., 1]), sd, t + ".adapt_pos1d")                  # :379
    kr = memory + posemb_row[:, None]
    kc = memory + posemb_col[:, :, None]
    a = rcda(tgt + qpx, tgt + qpy, kr, kc, memory, sd, p + ".cross_attn", mask)    # :385-392
    tgt = layer_norm(tgt + a.transpose(0, 1), sd, p + ".norm1")
    return ffn(tgt, sd, p + ".ffn")


def reference_points(sd, bs, spatial_prior, num_position, num_pattern, points=None):
    """A2/models/transformer.py:114-135."""
    if spatial_prior == "learned":
        return sd["transformer.position.weight"][None].repeat(bs, num_pattern, 1)
    if spatial_prior == "grid":
        n = round(math.sqrt(num_position))
        x = (torch.arange(n) + 0.5) / n
        xy = torch.meshgrid(x, x, indexing="ij")
        ref = torch.cat([xy[0].reshape(-1)[..., None], xy[1].reshape(-1)[..., None]], -1)
        return ref[None].repeat(bs, num_pattern, 1)
    if spatial_prior == "defined":
        return torch.as_tensor(points, dtype=torch.float32)[None].repeat(bs, num_pattern, 1)
    raise ValueError(spatial_prior)


def transformer(src, mask, sd, spatial_prior="learned", num_position=300, num_pattern=1, enc=6, dec=6,
                points=None, all_layers=False, stage=2, taps=None):
    """A2/models/transformer.py:109-215 for num_feature_levels == 1.  src: [N,C,H,W]."""
    N, C, H, W = src.shape
    t = "transformer"
    ref = reference_points(sd, N, spatial_prior, num_position, num_pattern, points).to(src.dtype)
    Qp = ref.shape[1] // num_pattern
    pat = sd[t + (".pattern.weight" if stage == 2 else ".modify_pattern.weight")]     # A1/models/transformer.py:66
    tgt = pat.reshape(1, num_pattern, 1, C).repeat(N, 1, Qp, 1).reshape(N, num_pattern * Qp, C)
    pos_col, pos_row = mask2pos(mask)
    posemb_row = mlp2(pos2posemb1d(pos_row).to(src.dtype), sd, t + ".adapt_pos1d")     # [N,W,C]
    posemb_col = mlp2(pos2posemb1d(pos_col).to(src.dtype), sd, t + ".adapt_pos1d")     # [N,H,C]
    x = src.permute(0, 2, 3, 1)
    for i in range(enc):
        x = encoder_layer(x, mask, posemb_row, posemb_col, sd, f"{t}.encoder_layers.{i}")
        if taps is not None:
            taps[f"enc{i}"] = x.permute(0, 3, 1, 2)          # NCHW like the reference's encoder layer output
    memory = x
    outs = []
    out = tgt
    inv_ref = inverse_sigmoid(ref)
    for i in range(dec):
        out = decoder_layer(out, ref, memory, mask, posemb_row, posemb_col, sd, f"{t}.decoder_layers.{i}")
        if taps is not None:
            taps[f"hs{i}"] = out
        logits = F.linear(out, sd[f"{t}.cls_embed.{i}.weight"], sd[f"{t}.cls_embed.{i}.bias"])
        tmp = mlp3(out, sd, f"{t}.bbox_embed.{i}")
        tmp = torch.cat([tmp[..., :2] + inv_ref, tmp[..., 2:]], -1)                     # :200
        boxes = tmp.sigmoid()
        var = mlp3(out, sd, f"{t}.bbox_variance.{i}") if stage == 2 else None
        outs.append((logits, boxes, var))
    res = {"pred_logits": outs[-1][0], "pred_boxes": outs[-1][1], "pred_vars": outs[-1][2]}
    if all_layers:       # aux_loss=True (A2/models/anchor_detr.py:129-140) -- with pred_vars added, which the reference forgets
        res["all_layers"] = outs
        res["aux_outputs"] = [{"pred_logits": a, "pred_boxes": b, "pred_vars": c} for a, b, c in outs[:-1]]
    res["memory"] = memory
    return res, ref


def nested(images):
    """A2/util/misc.py:291-310 for a list of [3,h,w] tensors (or a [B,3,H,W] tensor)."""
    if isinstance(images, torch.Tensor):
        images = list(images)
    hh = max(i.shape[1] for i in images)
    ww = max(i.shape[2] for i in images)
    t = torch.zeros((len(images), images[0].shape[0], hh, ww), dtype=images[0].dtype)
    m = torch.ones((len(images), hh, ww), dtype=torch.bool)
    for i, img in enumerate(images):
        t[i, :, : img.shape[1], : img.shape[2]] = img
        m[i, : img.shape[1], : img.shape[2]] = False
    return t, m


def forward(images, rects, sd, mask=None, **kw):
    """AnchorDETR.forward, A2/models/anchor_detr.py:94-133 -> (out dict, reference_points)."""
    if mask is None:
        images, mask = nested(images)
    feat, m = extract_feature(images, mask, rects, sd, exemplar_mode=kw.pop("exemplar_mode", "reference"))
    src = aggr_input_proj(feat, sd)
    taps = kw.get("taps")
    if taps is not None:
        taps["layer4"] = feat[:, : feat.shape[1] // 2]
        taps["proj"] = src
    return transformer(src, m, sd, **kw)


def forward_stage1(images, points, sd, mask=None, **kw):
    """1st-stage AnchorDETR.forward, A1/models/anchor_detr.py:80-113: plain backbone + input_proj, `defined` anchor points
    (A1/models/transformer.py:114-121), no variance head -> {"pred_logits","pred_wh","pred_points"}."""
    if mask is None:
        images, mask = nested(images)
    x = resnet50_dc5(images, sd)
    h, w = x.shape[-2:]
    m = F.interpolate(mask[None].float(), size=(h, w)).to(torch.bool)[0]
    src = aggr_input_proj(x, sd, p="input_proj.0")
    pts = torch.as_tensor(points, dtype=torch.float32).reshape(-1, 2)
    out, _ = transformer(src, m, sd, spatial_prior="defined", num_position=pts.shape[0], points=pts, stage=1, **kw)
    b = out["pred_boxes"]
    return {"pred_logits": out["pred_logits"], "pred_wh": b[..., 2:], "pred_points": b[..., :2]}
