"""AnchorDETR model, SetCriterion, PostProcess and build() -- the API surface of A2/models/anchor_detr.py
(AnchorDETR :34-140, SetCriterion :143-367, PostProcess :370-402, build :405-445) on the MI355X kernels.

`build(args)` returns `(model, criterion, postprocessors)` exactly like the reference; `model(samples, points=None,
rects=rects)` returns `({"pred_logits","pred_boxes","pred_vars"}, reference_points)`; `criterion(outputs, targets)`
returns the dict of 0-dim losses and exposes `.weight_dict` / `.matcher`.  The criterion is sync-free: the matcher
runs on the device and matched pairs are consumed as device index tensors.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import box_ops, ops
from .backbone import build_backbone
from .matcher import build_matcher
from .misc import NestedTensor, get_world_size, is_dist_avail_and_initialized, nested_tensor_from_tensor_list
from .transformer import build_transformer


CONCAT_FREE = os.environ.get("CDETR_CONCAT_FREE", "1") != "0"     # fold the exemplar product into the projection weight (ops.AggrProjFn)


class _Conv1x1(nn.Module):
    """nn.Conv2d(cin, cout, 1) parameters (`weight` [cout,cin,1,1], `bias`); runs as an MFMA GEMM on NHWC rows."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 1, 1))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.xavier_uniform_(self.weight, gain=1)                  # A2/models/anchor_detr.py:86-92

    def forward(self, x_nhwc):
        return ops.linear(x_nhwc, self.weight, self.bias)


class _ProjGN(nn.Sequential):
    """nn.Sequential(Conv2d 1x1, GroupNorm(32, d)) with state-dict keys `0.*`, `1.*`."""

    def __init__(self, cin, d):
        super().__init__(_Conv1x1(cin, d), nn.GroupNorm(32, d))

    def forward(self, x_nhwc):
        if isinstance(x_nhwc, tuple):                                 # (features, rects, extents, per-image flag): concat-free projection
            x, rects, extent, per_image = x_nhwc
            y = ops.AggrProjFn.apply(x, rects, extent, per_image, self[0].weight, self[0].bias)
        else:
            y = self[0](x_nhwc)                                       # [B,h,w,d]
        gn = self[1]
        cpg = y.shape[-1] // gn.num_groups
        if not (y.is_cuda and cpg % 4 == 0 and cpg <= 64):        # no fallback: the product path is the HIP kernel or nothing
            raise RuntimeError(f"GroupNorm({gn.num_groups}, {y.shape[-1]}) on {y.device}: cdetr_groupnorm_fwd needs a GPU tensor with "
                               "channels-per-group a multiple of 4 and <= 64")
        return ops.GroupNormNHWCFn.apply(y, gn.weight, gn.bias, gn.num_groups, gn.eps)          # statistics over (8 channels x h x w), NHWC


class AnchorDETR(nn.Module):
    """A2/models/anchor_detr.py:34-140."""

    def __init__(self, backbone, transformer, num_feature_levels, aux_loss=True):
        super().__init__()
        assert num_feature_levels == 1
        self.transformer = transformer
        hidden_dim = transformer.d_model
        self.num_feature_levels = num_feature_levels
        # built but never used on the stage-2 path (:68-74 vs :119): parameters with no gradient
        self.input_proj = nn.ModuleList([_ProjGN(backbone.num_channels[0], hidden_dim)])
        self.backbone = backbone
        self.aux_loss = aux_loss
        self.transformer.all_layer_heads = bool(aux_loss)
        self.aggr_input_proj = nn.ModuleList([_ProjGN(backbone.num_channels[0] * 2, hidden_dim)])
        self.taps = None      # a dict: forward() leaves the intermediates there (layer4 features, projected source, every encoder /
        #                       decoder layer's output) -- the full-size parity tests compare their digests with the reference's

    def forward(self, samples, points=None, rects=None):
        if not isinstance(samples, NestedTensor):
            samples = nested_tensor_from_tensor_list(samples)
        images, mask = samples.decompose()
        x, mi = self.backbone.features(images, mask)                          # NHWC [B,h,w,2048] + everything derived from the mask
        if ops.AFTER_BACKBONE is not None:
            ops.AFTER_BACKBONE()
        per_image = self.backbone.exemplar_mode == "per_image"
        if CONCAT_FREE:     # the exemplar product folds into the projection weight: no [B,h,w,4096] tensor
            src = self.aggr_input_proj[0]((x, rects, mi.extent, per_image))
        else:
            pf = ops.ExemplarFeatureFn.apply(x, rects, mi.extent, per_image)
            src = self.aggr_input_proj[0](torch.cat([x, x * pf[:, None, None, :]], dim=-1))
        self.transformer.taps = self.taps
        if self.taps is not None:
            self.taps["layer4"] = x.detach()
            self.taps["proj"] = src.detach()
        (outputs_class, outputs_coord, outputs_var), reference_points = self.transformer(src, mi, points)
        out = {"pred_logits": outputs_class[-1], "pred_boxes": outputs_coord[-1], "pred_vars": outputs_var[-1]}
        if self.aux_loss:
            # A2/models/anchor_detr.py:129-140.  The reference's _set_aux_loss leaves pred_vars out and its criterion then raises
            # KeyError in loss_variance on the first aux layer (the shipped scripts all pass --no_aux_loss); the variances of the
            # intermediate layers are added here so that aux_loss=True trains.
            out["aux_outputs"] = [{"pred_logits": a, "pred_boxes": b, "pred_vars": c}
                                  for a, b, c in zip(outputs_class[:-1], outputs_coord[:-1], outputs_var[:-1])]
        return out, reference_points


def sigmoid_focal_loss(inputs, targets, num_boxes, alpha: float = 0.25, gamma: float = 2):
    """A2/models/segmentation.py:198-223."""
    prob = inputs.sigmoid()
    ce_loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce_loss * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum() / num_boxes


class SetCriterion(nn.Module):
    """A2/models/anchor_detr.py:143-367 with losses [labels, boxes, cardinality, vars].  Matching happens on the
    device (`matcher.match_device`); `indices` are device tensors, so the whole criterion issues no host sync."""

    def __init__(self, num_classes, matcher, weight_dict, losses, focal_alpha=0.25):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict, self.losses, self.focal_alpha = \
            num_classes, matcher, weight_dict, losses, focal_alpha
        self.check_status = False   # set True to raise (with a host sync) on an invalid / infeasible cost matrix
        import os
        self.fused = os.environ.get("CDETR_FUSED_CRITERION", "1") != "0"   # losses + their gradients in one kernel (ops.CriterionFn); False = tensor-op composition
        self._nb_cache = {}
        self._w6_cache = {}
        self._plans = {}            # (target counts, Q, device) -> MatchPlan (device offset tables built once: graph-safe)
        self._pre = None            # (cost matrices, the logits they belong to): pre_match()

    # -- matched (batch, query, target-row) index tensors on the device
    def _matched(self, idx_i, idx_j, plan):
        dev = idx_i.device
        bidx = torch.cat([torch.full((m,), b, dtype=torch.int64, device=dev) for b, m in enumerate(plan.M)])
        sidx = torch.cat([idx_i[b, :m] for b, m in enumerate(plan.M)])
        tidx = torch.cat([idx_j[b, :m] + plan.tgt_off_host[b] for b, m in enumerate(plan.M)])
        return bidx, sidx, tidx

    def _targets_of(self, targets, Q, device):
        if isinstance(targets, ops.PackedTargets):     # fixed-address buffers + capacity plan (the graph-cached step): counts live on the device
            return targets.plan, targets.boxes, targets.labels, True
        sizes = tuple(len(t["boxes"]) for t in targets)
        return (self._plan(sizes, Q, device), torch.cat([t["boxes"] for t in targets]).to(torch.float32),
                torch.cat([t["labels"] for t in targets]), False)

    @torch.no_grad()
    def pre_match(self, outputs, targets):
        """The Hungarian cost matrices of the last layer, ahead of `forward` (which then starts with the assignment solve itself): the
        trainer captures this as its own small graph so that work for the next batch can be released exactly when the solve starts.
        -> False when the configuration matches several layers (aux_loss): nothing is computed ahead then."""
        if outputs.get("aux_outputs"):
            return False
        logits = outputs["pred_logits"]
        plan, tgt_boxes, _, _ = self._targets_of(targets, logits.shape[1], logits.device)
        self._pre = (self.matcher.cost_device(outputs, plan, tgt_boxes), logits)
        return True

    def forward(self, outputs, targets, num_boxes=None):
        """`num_boxes`: optional precomputed normaliser (device scalar) -- the data-parallel trainer all-reduces the target
        count BEFORE the (graph-captured) step; otherwise it is computed here as in the reference (:321-325).
        With `aux_outputs` (aux_loss=True, :334-350) every intermediate decoder layer gets its own Hungarian matching and the
        same losses under the key suffix `_i`; the matchings of all layers are ONE cost launch + ONE assignment launch
        (layers x images independent problems, one wavefront each), then one fused loss launch per layer."""
        out = {k: v for k, v in outputs.items() if k not in ("aux_outputs", "enc_outputs")}
        logits = out["pred_logits"]
        B, Q = logits.shape[:2]
        aux = outputs.get("aux_outputs")
        plan, tgt_boxes_all, tgt_labels_all, packed = self._targets_of(targets, Q, logits.device)
        sizes = tuple(plan.sizes)
        pre, self._pre = self._pre, None
        cost = pre[0] if (pre is not None and pre[1] is logits) else None
        if aux and packed:
            # capacity plan: the stacked form needs the targets packed tightly L times over (count-dependent offsets); the layers are
            # matched one after the other with the one device-resident plan instead -- 2 L launches, graph-replayable for any counts
            layers = list(aux) + [out]
            L = len(layers)
            per = [self.matcher.match_device(lo, None, plan, tgt_boxes=tgt_boxes_all) for lo in layers]
            idx_i, idx_j = torch.cat([r[0] for r in per]), torch.cat([r[1] for r in per])
            status = torch.cat([r[2] for r in per])
        elif aux:
            layers = list(aux) + [out]
            L = len(layers)
            plan_all = self._plan(sizes * L, Q, logits.device)
            stacked = {k: torch.cat([l[k] for l in layers]) for k in ("pred_logits", "pred_boxes")}       # [L*B, Q, .]
            idx_i, idx_j, status, _ = self.matcher.match_device(stacked, None, plan_all, tgt_boxes=tgt_boxes_all.repeat(L, 1))
        else:
            layers, L = [out], 1
            idx_i, idx_j, status, _ = self.matcher.match_device(out, targets, plan, tgt_boxes=tgt_boxes_all, cost=cost)
        if self.check_status and bool((status != 0).any()):
            raise ValueError("invalid or infeasible matching cost matrix")
        if num_boxes is not None:
            pass
        elif is_dist_avail_and_initialized():
            num_boxes = sum(plan.sizes)
            nb = torch.as_tensor([num_boxes], dtype=torch.float, device=logits.device)
            torch.distributed.all_reduce(nb)
            num_boxes = nb / get_world_size()
            num_boxes = torch.clamp(num_boxes, min=1)[0]                      # stays on the device (no .item())
        else:
            num_boxes = max(float(sum(plan.sizes)), 1.0)                      # :321-325
        fused = self.fused and logits.is_cuda and list(self.losses) == ["labels", "boxes", "cardinality", "vars"]
        nbt = None
        if fused:
            if not torch.is_tensor(num_boxes):
                nbt = self._nb_cache.get((float(num_boxes), str(logits.device)))
                if nbt is None:
                    if len(self._nb_cache) > 64:
                        self._nb_cache.clear()
                    nbt = self._nb_cache[(float(num_boxes), str(logits.device))] = torch.full((1,), float(num_boxes), device=logits.device)
            else:
                nbt = num_boxes.reshape(-1)[:1].to(torch.float32)
        losses = {}
        self.last_total = None      # weighted total (A2/engine.py:37) when the fused kernel produced it: the trainer backpropagates this
        totals = []
        for li, lo in enumerate(layers):
            ii, jj = idx_i[li * B:(li + 1) * B], idx_j[li * B:(li + 1) * B]
            last = li == L - 1
            if fused:
                vec, tot = ops.CriterionFn.apply(lo["pred_logits"], lo["pred_boxes"], lo["pred_vars"], tgt_boxes_all,
                                                 tgt_labels_all.to(torch.int64), plan, ii, jj, nbt, self.num_classes, self.focal_alpha,
                                                 self._weights6(logits.device, None if last else li))
                totals.append(tot)
                d = {"loss_ce": vec[0], "class_error": vec[1].detach(), "cardinality_error": vec[2].detach(), "loss_bbox": vec[3],
                     "loss_giou": vec[4], "loss_variance": vec[5]}
            else:
                bidx, sidx, tidx = self._matched(ii, jj, plan)
                d = {}
                for loss in self.losses:
                    d.update(getattr(self, "loss_" + loss)(lo, plan, bidx, sidx, tidx, tgt_boxes_all, tgt_labels_all, num_boxes))
            if last:
                losses.update(d)
            else:
                d.pop("class_error", None)          # logged for the last layer only (log=False, :343-345)
                losses.update({k + f"_{li}": v for k, v in d.items()})
        if totals:
            self.last_total = totals[0] if len(totals) == 1 else torch.stack(totals).sum()
        return losses

    def _weights6(self, device, layer):
        """weight_dict entries of one layer's six scalars, in the fused kernel's order (0 for the logged-only ones), on the device."""
        key = (str(device), layer)
        w = self._w6_cache.get(key)
        if w is None:
            suf = "" if layer is None else f"_{layer}"
            order = ("loss_ce", "class_error", "cardinality_error", "loss_bbox", "loss_giou", "loss_variance")
            w = self._w6_cache[key] = torch.tensor([float(self.weight_dict.get(k + suf, 0.0)) for k in order], device=device)
        return w

    def _plan(self, sizes, Q, device):
        key = (tuple(sizes), Q, str(device))
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) > 256:          # real data: a new plan per distinct tuple of target counts
                self._plans.clear()
            plan = self._plans[key] = ops.MatchPlan(key[0], Q, device)
        return plan

    def loss_labels(self, out, plan, bidx, sidx, tidx, tb, tl, num_boxes):
        src_logits = out["pred_logits"]                                       # :166-197
        target_classes = torch.full(src_logits.shape[:2], self.num_classes, dtype=torch.int64, device=src_logits.device)
        tco = tl[tidx]
        target_classes[bidx, sidx] = tco
        onehot = torch.zeros([src_logits.shape[0], src_logits.shape[1], src_logits.shape[2] + 1], dtype=src_logits.dtype,
                             device=src_logits.device)
        onehot.scatter_(2, target_classes.unsqueeze(-1), 1)
        onehot = onehot[:, :, :-1]
        loss_ce = sigmoid_focal_loss(src_logits, onehot, num_boxes, alpha=self.focal_alpha, gamma=2) * src_logits.shape[1]
        with torch.no_grad():
            if tco.numel() == 0:
                err = torch.full((), 100.0, device=src_logits.device)
            else:
                err = 100 - (src_logits[bidx, sidx].argmax(-1) == tco).float().sum() * (100.0 / tco.numel())
        return {"loss_ce": loss_ce, "class_error": err}

    @torch.no_grad()
    def loss_cardinality(self, out, plan, bidx, sidx, tidx, tb, tl, num_boxes):
        pred_logits = out["pred_logits"]                                      # :199-211
        tgt_lengths = plan.sizes_f
        card_pred = (pred_logits.argmax(-1) != pred_logits.shape[-1] - 1).sum(1)
        return {"cardinality_error": F.l1_loss(card_pred.float(), tgt_lengths)}

    def loss_boxes(self, out, plan, bidx, sidx, tidx, tb, tl, num_boxes):
        src_boxes = out["pred_boxes"][bidx, sidx]                             # :213-234
        target_boxes = tb[tidx]
        loss_bbox = (src_boxes - target_boxes).abs().sum() / num_boxes
        giou = box_ops.generalized_box_iou_pairs(box_ops.box_cxcywh_to_xyxy(src_boxes),
                                                 box_ops.box_cxcywh_to_xyxy(target_boxes))
        return {"loss_bbox": loss_bbox, "loss_giou": (1 - giou).sum() / num_boxes}

    def loss_vars(self, out, plan, bidx, sidx, tidx, tb, tl, num_boxes):
        src_boxes = out["pred_boxes"][bidx, sidx]                             # :264-289
        target_boxes = tb[tidx]
        pv = out["pred_vars"][bidx, sidx]
        lw = (src_boxes[:, 2] - target_boxes[:, 2]).abs().mean() / pv[:, 0].abs() + pv[:, 0].log().abs()
        lh = (src_boxes[:, 3] - target_boxes[:, 3]).abs().mean() / pv[:, 1].abs() + pv[:, 1].log().abs()
        return {"loss_variance": ((lw + lh) / num_boxes).sum()}


class PostProcess(nn.Module):
    """A2/models/anchor_detr.py:370-402 (returned by build() for API parity; unused by the 2nd-stage engine)."""

    @torch.no_grad()
    def forward(self, outputs, target_sizes):
        out_logits, out_bbox = outputs["pred_logits"], outputs["pred_boxes"]
        prob = out_logits.sigmoid()
        k = min(100, prob[0].numel())
        topk_values, topk_indexes = torch.topk(prob.view(out_logits.shape[0], -1), k, dim=1)
        topk_boxes = topk_indexes // out_logits.shape[2]
        labels = topk_indexes % out_logits.shape[2]
        boxes = box_ops.box_cxcywh_to_xyxy(out_bbox)
        boxes = torch.gather(boxes, 1, topk_boxes.unsqueeze(-1).repeat(1, 1, 4))
        img_h, img_w = target_sizes.unbind(1)
        boxes = boxes * torch.stack([img_w, img_h, img_w, img_h], dim=1)[:, None, :]
        return [{"scores": s, "labels": l, "boxes": b} for s, l, b in zip(topk_values, labels, boxes)]


def build(args):
    """A2/models/anchor_detr.py:405-445."""
    from . import _ffi
    _ffi.lib()                                # fail loudly now if the HIP library is missing -- there is no fallback
    num_classes = 1
    device = torch.device(args.device)
    backbone = build_backbone(args)
    transformer = build_transformer(args)
    model = AnchorDETR(backbone, transformer, num_feature_levels=args.num_feature_levels, aux_loss=args.aux_loss)
    matcher = build_matcher(args)
    weight_dict = {"loss_ce": args.cls_loss_coef, "loss_bbox": args.bbox_loss_coef, "loss_giou": args.giou_loss_coef,
                   "loss_variance": args.variance_loss_coef}
    if args.aux_loss:
        aux = {}
        for i in range(args.dec_layers - 1):
            aux.update({k + f"_{i}": v for k, v in weight_dict.items()})
        aux.update({k + "_enc": v for k, v in weight_dict.items()})
        weight_dict.update(aux)
    losses = ["labels", "boxes", "cardinality", "vars"]
    criterion = SetCriterion(num_classes, matcher, weight_dict, losses, focal_alpha=args.focal_alpha)
    criterion.to(device)
    return model, criterion, {"bbox": PostProcess()}
