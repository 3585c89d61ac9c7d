"""Anchor-DETR transformer (RCDA encoder / decoder, shared heads) on the HIP kernels.

API mirror of A2/models/transformer.py (Transformer :21-215, TransformerEncoderLayerSpatial :218-279,
TransformerDecoderLayer :316-409, FFN :412-426, MLP :429-439, pos2posemb* / mask2pos :474-503) and of
A2/models/row_column_decoupled_attention.py (MultiheadRCDA :324-537): same parameter names / shapes, same math.
Every dense contraction goes through the MFMA implicit-GEMM kernel (bias / ReLU / residual fused in the epilogue) and
the attention core through the fused RCDA kernels; feature maps stay NHWC end to end (the reference permutes
NCHW<->NHWC around every encoder layer and materialises four `.repeat`-ed [B,h,w,256] temporaries per layer).
k_row / k_col use mean-before-project (linear o mean = mean o linear): only 1/h resp. 1/w of the key projections remain.
"""
import math

import torch
from torch import nn

from . import ops


def _stack(xs):
    return xs[0].unsqueeze(0) if len(xs) == 1 else torch.stack(xs)


def inverse_sigmoid(x, eps=1e-5):
    """A2/util/misc.py:475-479."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def pos2posemb1d(pos, num_pos_feats=256, temperature=10000):
    """A2/models/transformer.py:487-494."""
    return ops.sine_embed(pos, num_pos_feats, temperature)


def pos2posemb2d(pos, num_pos_feats=128, temperature=10000):
    """A2/models/transformer.py:474-484 ((y, x) concatenation order)."""
    return ops.sine_embed(pos, num_pos_feats, temperature, two_d=True)


def mask2pos(mask):
    """A2/models/transformer.py:497-503."""
    nm = ~mask
    y = nm[:, :, 0].cumsum(1, dtype=torch.float32)
    x = nm[:, 0, :].cumsum(1, dtype=torch.float32)
    return (y - 0.5) / y[:, -1:], (x - 0.5) / x[:, -1:]


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm parameters; forward on the fused HIP kernels (C multiple of 256), in-place parameter gradients."""

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)


class Linear(nn.Linear):
    def forward(self, x, relu=False, resid=None):
        return ops.linear(x, self.weight, self.bias, relu=relu, resid=resid)


class PosMLP(nn.Sequential):
    """nn.Sequential(Linear, ReLU, Linear) with keys `0.*` and `2.*` (A2/models/transformer.py:73-74)."""

    def __init__(self, d):
        super().__init__(Linear(d, d), nn.ReLU(), Linear(d, d))

    def forward(self, x):
        return self[2](self[0](x, relu=True))


class MLP(nn.Module):
    """A2/models/transformer.py:429-439."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = layer(x, relu=(i < self.num_layers - 1))
        return x


def pos_mlp_many(mlp, xs):
    """[mlp(x) for x in xs] for a PosMLP, level by level: the first linears of all inputs share one grouped submission
    (ops.gemm_queue), then the second ones."""
    if not (xs[0].is_cuda and torch.is_grad_enabled()):
        return [mlp(x) for x in xs]
    if ops.FUSED_HEADS:             # one autograd node: the ReLU masks ride in the data-gradient epilogues of its backward
        return list(ops.PosMlpFn.apply(mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, *xs))
    with ops.gemm_queue():
        hs = [mlp[0](x, relu=True) for x in xs]
    with ops.gemm_queue():
        return [mlp[2](h) for h in hs]


def mlps_levelwise(mlps, x):
    """[m(x) for m in mlps] for MLP / Linear heads on the same input, level by level (sibling layers grouped)."""
    if not (x.is_cuda and torch.is_grad_enabled()):
        return [m(x) for m in mlps]
    chains = [list(m.layers) if isinstance(m, MLP) else [m] for m in mlps]
    cur = [x] * len(mlps)
    for lvl in range(max(len(c) for c in chains)):
        with ops.gemm_queue():
            for i, c in enumerate(chains):
                if lvl < len(c):
                    cur[i] = c[lvl](cur[i], relu=(lvl < len(c) - 1))
    return cur


class FFN(nn.Module):
    """A2/models/transformer.py:412-426 (post-norm; the residual add is fused into linear2's epilogue)."""

    def __init__(self, d_model=256, d_ffn=1024):
        super().__init__()
        self.linear1 = Linear(d_model, d_ffn)
        self.linear2 = Linear(d_ffn, d_model)
        self.norm2 = LayerNorm(d_model)

    def forward(self, src):
        return self.norm2(self.linear2(self.linear1(src, relu=True), resid=src))


class MultiheadRCDA(nn.Module):
    """Drop-in for A2/models/row_column_decoupled_attention.py:324-537 (same parameters: in_proj_weight [5E,E],
    in_proj_bias [5E], out_proj).  forward(...) -> (attn_output [L,N,E], None)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, bias=True):
        super().__init__()
        assert dropout == 0.0, "the fused kernels implement the reference's shipped dropout=0 path"
        assert embed_dim == num_heads * 32, "the RCDA kernels are specialised for head_dim 32"
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(5 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(5 * embed_dim))
        self.out_proj = Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)                     # :420-432
        nn.init.constant_(self.out_proj.bias, 0.0)

    def _proj(self, x, i):
        E = self.embed_dim
        return ops.linear(x, self.in_proj_weight, self.in_proj_bias, rows=(i * E, (i + 1) * E))

    def attend(self, query_row, query_col, key_row_mean, key_col_mean, value, mask_row, mask_col, resid=None):
        """Batch-first core: query_* [N,L,E]; key_row_mean [N,W,E] (mean over H of the row keys), key_col_mean [N,H,E];
        value [N,H,W,E]; masks uint8 [N,W] / [N,H].  Returns out_proj(attn) (+ resid) as [N,L,E]."""
        q_row, q_col = self._proj(query_row, 0), self._proj(query_col, 1)
        k_row, k_col = self._proj(key_row_mean, 2), self._proj(key_col_mean, 3)
        v = self._proj(value, 4)
        o = ops.rcda_core(q_row, q_col, k_row, k_col, v, mask_row, mask_col, self.num_heads)
        return self.out_proj(o, resid=resid)

    def forward(self, query_row, query_col, key_row, key_col, value, key_padding_mask=None, need_weights=False,
                attn_mask=None):
        assert attn_mask is None and not need_weights
        mr = mc = None
        if key_padding_mask is not None:                                  # :238-249 first row / first column rule
            mr = key_padding_mask[:, 0, :].to(torch.uint8).contiguous()
            mc = key_padding_mask[:, :, 0].to(torch.uint8).contiguous()
        out = self.attend(query_row, query_col, key_row.mean(1), key_col.mean(2), value, mr, mc)
        return out.transpose(0, 1), None


class MultiheadSelfAttention(nn.Module):
    """nn.MultiheadAttention(256, 8) as used at A2/models/transformer.py:337,369-370 (same parameter names).  Q x Q
    logits are tiny (8*300^2); projections run on the MFMA GEMM kernel, the softmax(QK^T)V core on batched GEMMs."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def forward(self, qk_in, v_in, resid=None):
        """qk_in, v_in: [N,L,E] batch-first.  Returns out_proj(attn) (+ resid) [N,L,E]."""
        N, L, E = qk_in.shape
        nh, d = self.num_heads, E // self.num_heads
        qk = ops.linear(qk_in, self.in_proj_weight, self.in_proj_bias, rows=(0, 2 * E))        # q and k in one GEMM
        v = ops.linear(v_in, self.in_proj_weight, self.in_proj_bias, rows=(2 * E, 3 * E))
        o = ops.mha_core(qk, v, nh)                       # fused scale / QK^T / softmax / PV kernel
        return self.out_proj(o, resid=resid)


class TransformerEncoderLayerSpatial(nn.Module):
    """A2/models/transformer.py:218-279, NHWC in / NHWC out."""

    def __init__(self, d_model=256, d_ffn=1024, n_heads=8):
        super().__init__()
        self.self_attn = MultiheadRCDA(d_model, n_heads)
        self.norm1 = LayerNorm(d_model)
        self.ffn = FFN(d_model, d_ffn)

    fused = True     # one autograd node with a hand-scheduled backward (ops.EncoderLayerFn); False = op-by-op autograd

    def forward(self, src, mask_row, mask_col, posemb_row, posemb_col):
        if self.fused and torch.is_grad_enabled():
            return ops.EncoderLayerFn.apply(src, posemb_row, posemb_col, mask_row, mask_col, self, self.norm1.weight)
        return self.forward_unfused(src, mask_row, mask_col, posemb_row, posemb_col)

    def forward_unfused(self, src, mask_row, mask_col, posemb_row, posemb_col):
        N, H, W, Cc = src.shape
        q_row = (src + posemb_row[:, None]).reshape(N, H * W, Cc)        # broadcast over h  (:248)
        q_col = (src + posemb_col[:, :, None]).reshape(N, H * W, Cc)     # broadcast over w  (:249)
        k_row_mean = src.mean(1) + posemb_row                            # mean_H(src + pos_row)
        k_col_mean = src.mean(2) + posemb_col
        a = self.self_attn.attend(q_row, q_col, k_row_mean, k_col_mean, src, mask_row, mask_col,
                                  resid=src.reshape(N, H * W, Cc))
        src = self.norm1(a).reshape(N, H, W, Cc)
        return self.ffn(src)


class TransformerDecoderLayer(nn.Module):
    """A2/models/transformer.py:316-409 (single feature level)."""

    def __init__(self, d_model=256, d_ffn=1024, n_heads=8):
        super().__init__()
        self.cross_attn = MultiheadRCDA(d_model, n_heads)
        self.norm1 = LayerNorm(d_model)
        self.self_attn = MultiheadSelfAttention(d_model, n_heads)
        self.norm2 = LayerNorm(d_model)
        self.ffn = FFN(d_model, d_ffn)

    def forward(self, tgt, query_pos, query_pos_x, query_pos_y, memory, k_row_mean, k_col_mean, mask_row, mask_col):
        tgt = self.norm2(self.self_attn(tgt + query_pos, tgt, resid=tgt))                        # :369-372
        a = self.cross_attn.attend(tgt + query_pos_x, tgt + query_pos_y, k_row_mean, k_col_mean, memory,
                                   mask_row, mask_col, resid=tgt)                                # :385-403
        tgt = self.norm1(a)
        return self.ffn(tgt)


class Transformer(nn.Module):
    """A2/models/transformer.py:21-215 for num_feature_levels == 1, attention_type == 'RCDA'."""

    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024,
                 dropout=0.0, activation="relu", num_feature_levels=1, num_query_position=300, num_query_pattern=3,
                 spatial_prior="learned", attention_type="RCDA", stage=2):
        super().__init__()
        assert num_feature_levels == 1 and attention_type == "RCDA" and dropout == 0.0 and activation == "relu"
        self.d_model, self.nhead, self.stage = d_model, nhead, stage
        self.all_layer_heads = True     # AnchorDETR sets this to its aux_loss flag
        self.taps = None                # see AnchorDETR.taps
        import os
        self.fused_decoder = os.environ.get("CDETR_FUSED_DECODER", "1") != "0"   # all decoder layers as one autograd node (ops.DecoderStackFn); False = op-by-op autograd
        self.encoder_layers = nn.ModuleList(
            TransformerEncoderLayerSpatial(d_model, dim_feedforward, nhead) for _ in range(num_encoder_layers))
        self.decoder_layers = nn.ModuleList(
            TransformerDecoderLayer(d_model, dim_feedforward, nhead) for _ in range(num_decoder_layers))
        self.spatial_prior = spatial_prior
        self.num_pattern = num_query_pattern
        if stage == 2:
            self.pattern = nn.Embedding(self.num_pattern, d_model)
        else:
            self.modify_pattern = nn.Embedding(self.num_pattern, d_model)      # A1/models/transformer.py:66
        self.num_position = num_query_position
        if spatial_prior == "learned":
            self.position = nn.Embedding(self.num_position, 2)
        self.adapt_pos2d = PosMLP(d_model)
        self.adapt_pos1d = PosMLP(d_model)
        self.num_layers = num_decoder_layers
        cls_embed = Linear(d_model, 2)
        bbox_embed = MLP(d_model, d_model, 4, 3)
        prior_prob = 0.01
        # stage 2: bias [2] (:90-92); stage 1: ONE bias element broadcast over the 2 logits (A1/models/transformer.py:82-86)
        cls_embed.bias = nn.Parameter(torch.ones(2 if stage == 2 else 1) * (-math.log((1 - prior_prob) / prior_prob)))
        nn.init.constant_(bbox_embed.layers[-1].weight.data, 0)                                       # :94-95
        nn.init.constant_(bbox_embed.layers[-1].bias.data, 0)
        nn.init.constant_(bbox_embed.layers[-1].bias.data[2:], -2.0)                                  # :103
        if spatial_prior == "learned":
            nn.init.uniform_(self.position.weight.data, 0, 1)                                         # :101
        # ONE module each, aliased num_pred times (:104-107): 6 identical state-dict copies, grads accumulate
        self.cls_embed = nn.ModuleList([cls_embed for _ in range(num_decoder_layers)])
        self.bbox_embed = nn.ModuleList([bbox_embed for _ in range(num_decoder_layers)])
        if stage == 2:
            bbox_variance = MLP(d_model, d_model, 2, 3)
            nn.init.constant_(bbox_variance.layers[-1].weight.data, 0.01)                             # :97-98
            nn.init.constant_(bbox_variance.layers[-1].bias.data, 0.01)
            self.bbox_variance = nn.ModuleList([bbox_variance for _ in range(num_decoder_layers)])

    def reference_points(self, bs, device, points=None):
        """:114-135."""
        if self.spatial_prior == "learned":
            if self.position.weight.is_cuda:
                return ops.TileParamFn.apply(self.position.weight, bs, self.num_pattern, self.num_position, "tile")
            return self.position.weight.unsqueeze(0).repeat(bs, self.num_pattern, 1)
        if self.spatial_prior == "grid":
            nx = ny = round(math.sqrt(self.num_position))
            self.num_position = nx * ny
            x = (torch.arange(nx, device=device) + 0.5) / nx
            y = (torch.arange(ny, device=device) + 0.5) / ny
            xy = torch.meshgrid(x, y, indexing="ij")
            ref = torch.cat([xy[0].reshape(-1)[..., None], xy[1].reshape(-1)[..., None]], -1)
            return ref.unsqueeze(0).repeat(bs, self.num_pattern, 1)
        if self.spatial_prior == "defined":
            assert points is not None, "defined, provide points"
            pts = torch.as_tensor(points, dtype=torch.float32, device=device).reshape(-1, 2)
            self.num_position = pts.shape[0]
            return pts.unsqueeze(0).repeat(bs, self.num_pattern, 1)
        raise ValueError(f"unknown {self.spatial_prior} spatial prior")

    def forward(self, src, mask, points=None):
        """src: NHWC [B,h,w,C] (the product path keeps NHWC); mask: bool [B,h,w] or the ops.MaskInfo the model derived from the
        image-level padding mask.  Returns (([classes [B,Q,ncls]] , [coords [B,Q,4]], [vars [B,Q,2]]) -- one entry per decoder layer
        whose heads are evaluated (all with aux losses, else the last) --, reference_points [B,Q,2])."""
        bs, h, w, c = src.shape
        mi = mask if isinstance(mask, ops.MaskInfo) else ops.mask_prep(mask, h, w)
        reference_points = self.reference_points(bs, src.device, points)
        pattern = self.pattern if self.stage == 2 else self.modify_pattern
        if src.is_cuda:
            tgt = ops.TileParamFn.apply(pattern.weight, bs, self.num_pattern, self.num_position, "repeat")
        else:
            tgt = (pattern.weight.reshape(1, self.num_pattern, 1, c).repeat(bs, 1, self.num_position, 1)
                   .reshape(bs, self.num_pattern * self.num_position, c))
        # the four 1-d positional MLP applications (key rows / columns, query x / y) run level by level, grouped
        emb_x, emb_y = ops.sine_embed_xy(reference_points, c)
        posemb_row, posemb_col, query_pos_x, query_pos_y = pos_mlp_many(
            self.adapt_pos1d, [pos2posemb1d(mi.pos_row), pos2posemb1d(mi.pos_col), emb_x, emb_y])          # [B,w,C], [B,h,C], 2 x [B,Q,C]
        mask_row, mask_col = mi.mask_row, mi.mask_col
        grad = torch.is_grad_enabled()

        enc = list(self.encoder_layers)
        if enc and enc[0].fused and src.is_cuda:
            if grad:
                # (the embeddings the decoder sees are the node's own outputs: see EncoderStackFn.forward)
                memory, posemb_row, posemb_col = ops.EncoderStackFn.apply(src, posemb_row, posemb_col, mask_row, mask_col, enc, enc[0].norm1.weight, self.taps)
            else:       # inference: the same fused forward bodies, nothing saved
                memory = src
                with ops.scope(RCDA_SAVE=False):      # no backward: the attention maps are not written
                    for li, layer in enumerate(enc):
                        memory = ops.EncoderLayerFn.forward(ops._Ctx(), memory, posemb_row, posemb_col, mask_row, mask_col, layer, None)
                        if self.taps is not None:
                            self.taps[f"enc{li}"] = memory
        else:
            memory = src
            for li, layer in enumerate(enc):
                memory = layer(memory, mask_row, mask_col, posemb_row, posemb_col)
                if self.taps is not None:
                    self.taps[f"enc{li}"] = memory.detach()

        # query positional terms do not depend on the layer (the reference recomputes them in every layer, :366-379)
        query_pos = pos_mlp_many(self.adapt_pos2d, [pos2posemb2d(reference_points)])[0]
        last = len(self.decoder_layers) - 1
        if self.fused_decoder and src.is_cuda:
            args = (tgt, query_pos, query_pos_x, query_pos_y, memory, None, None, mask_row, mask_col, list(self.decoder_layers),
                    pattern.weight, posemb_row, posemb_col)
            if grad:
                layer_outs = ops.DecoderStackFn.apply(*args)
            else:
                with ops.scope(RCDA_SAVE=False):
                    layer_outs = ops.DecoderStackFn.forward(ops._Ctx(), *args)
        else:
            k_row_mean = memory.mean(1) + posemb_row                      # shared by the 6 decoder layers
            k_col_mean = memory.mean(2) + posemb_col
            layer_outs, output = [], tgt
            for layer in self.decoder_layers:
                output = layer(output, query_pos, query_pos_x, query_pos_y, memory, k_row_mean, k_col_mean, mask_row, mask_col)
                layer_outs.append(output)
        if self.taps is not None:
            for lid, o in enumerate(layer_outs):
                self.taps[f"hs{lid}"] = o.detach()
        # the heads are ONE module aliased over the layers (:104-107): with aux losses all layers go through them in one pass
        lids = list(range(len(self.decoder_layers))) if self.all_layer_heads else [last]
        output = layer_outs[last] if len(lids) == 1 else torch.stack([layer_outs[i] for i in lids])      # [(Ld,) B, Q, C]
        if self.stage == 2 and output.is_cuda and torch.is_grad_enabled() and ops.FUSED_HEADS:
            ce, be, ve = self.cls_embed[last], self.bbox_embed[last], self.bbox_variance[last]
            hp = [ce.weight, ce.bias] + [t for l in be.layers for t in (l.weight, l.bias)] + [t for l in ve.layers for t in (l.weight, l.bias)]
            outputs_class, tmp, var = ops.HeadsFn.apply(output, *hp)       # one autograd node: 6 backward launches instead of 15
        elif self.stage == 2:
            outputs_class, tmp, var = mlps_levelwise([self.cls_embed[last], self.bbox_embed[last], self.bbox_variance[last]], output)
        else:
            ce = self.cls_embed[last]
            outputs_class = ops.linear(output, ce.weight, None) + ce.bias
            tmp, var = self.bbox_embed[last](output), None
        coord = ops.BoxHeadFn.apply(tmp, reference_points)               # sigmoid(tmp + [inverse_sigmoid(ref), 0, 0])   (:193-203)
        if len(lids) == 1:
            return ([outputs_class], [coord], [var] if self.stage == 2 else None), reference_points
        return (list(outputs_class.unbind(0)), list(coord.unbind(0)), list(var.unbind(0)) if self.stage == 2 else None), reference_points


def build_transformer(args):
    return Transformer(d_model=args.hidden_dim, nhead=args.nheads, num_encoder_layers=args.enc_layers,
                       num_decoder_layers=args.dec_layers, dim_feedforward=args.dim_feedforward, dropout=args.dropout,
                       activation="relu", num_feature_levels=args.num_feature_levels,
                       num_query_position=args.num_query_position, num_query_pattern=args.num_query_pattern,
                       spatial_prior=args.spatial_prior, attention_type=args.attention_type)
