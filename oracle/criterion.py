"""CPU restatement of the matcher + SetCriterion + step tail (TEST INFRA, see oracle/__init__.py)."""
import torch
import torch.nn.functional as F

from . import lsap as _lsap


# ----------------------------------------------------------------------------- box ops (a11)
def box_cxcywh_to_xyxy(x):
    """A2/util/box_ops.py:17-20."""
    xc, yc, w, h = x.unbind(-1)
    return torch.stack([xc - 0.5 * w, yc - 0.5 * h, xc + 0.5 * w, yc + 0.5 * h], dim=-1)


def generalized_box_iou(b1, b2):
    """A2/util/box_ops.py:30-67 pairwise [N,M] (asserts on degenerate boxes kept)."""
    assert (b1[:, 2:] >= b1[:, :2]).all() and (b2[:, 2:] >= b2[:, :2]).all()
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = a1[:, None] + a2 - inter
    iou = inter / union
    lt2 = torch.min(b1[:, None, :2], b2[:, :2])
    rb2 = torch.max(b1[:, None, 2:], b2[:, 2:])
    wh2 = (rb2 - lt2).clamp(min=0)
    area = wh2[..., 0] * wh2[..., 1]
    return iou - (area - union) / area


# ----------------------------------------------------------------------------- matcher (a8, a9)
def match_cost(pred_logits, pred_boxes, tgt_boxes, tgt_ids=None, w_class=2.0, w_bbox=5.0, w_giou=2.0):
    """Per-image cost block, A2/models/matcher.py:222-242 (same expression and summation order).
    pred_logits [Q,2], pred_boxes [Q,4], tgt_boxes [T,4] -> [Q,T] float32."""
    p = pred_logits.sigmoid()
    neg = (1 - 0.25) * (p ** 2.0) * (-(1 - p + 1e-8).log())
    pos = 0.25 * ((1 - p) ** 2.0) * (-(p + 1e-8).log())
    ids = tgt_ids if tgt_ids is not None else torch.zeros(tgt_boxes.shape[0], dtype=torch.int64)
    cost_class = pos[:, ids] - neg[:, ids]
    cost_bbox = torch.cdist(pred_boxes, tgt_boxes, p=1)
    cost_giou = -generalized_box_iou(box_cxcywh_to_xyxy(pred_boxes), box_cxcywh_to_xyxy(tgt_boxes))
    return w_bbox * cost_bbox + w_class * cost_class + w_giou * cost_giou


@torch.no_grad()
def hungarian_match(outputs, targets, solver=None, **w):
    """A2/models/matcher.py:197-247 -> list of (idx_i, idx_j) int64 CPU tensors, idx_i ascending."""
    solver = solver or _lsap.linear_sum_assignment
    res = []
    for b, t in enumerate(targets):
        c = match_cost(outputs["pred_logits"][b], outputs["pred_boxes"][b], t["boxes"], t.get("labels"), **w)
        i, j = solver(c.cpu().numpy())
        res.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
    return res


# ----------------------------------------------------------------------------- losses (a10)
def sigmoid_focal_loss(inputs, targets, num_boxes, alpha=0.25, gamma=2):
    """A2/models/segmentation.py:198-223."""
    prob = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum() / num_boxes


def set_criterion(outputs, targets, indices=None, num_classes=1, world_size=1, num_boxes=None):
    """A2/models/anchor_detr.py:308-367 with losses [labels, boxes, cardinality, vars], no aux."""
    if indices is None:
        indices = hungarian_match(outputs, targets)
    if num_boxes is None:
        num_boxes = max(float(sum(len(t["labels"]) for t in targets)) / world_size, 1.0)   # :321-325
    logits, boxes, pvars = outputs["pred_logits"], outputs["pred_boxes"], outputs["pred_vars"]
    bidx = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)])
    sidx = torch.cat([s for s, _ in indices])
    # labels :166-197
    tco = torch.cat([t["labels"][j] for t, (_, j) in zip(targets, indices)])
    tc = torch.full(logits.shape[:2], num_classes, dtype=torch.int64)
    tc[bidx, sidx] = tco
    onehot = torch.zeros(logits.shape[0], logits.shape[1], logits.shape[2] + 1, dtype=logits.dtype)
    onehot.scatter_(2, tc.unsqueeze(-1), 1)
    onehot = onehot[:, :, :-1]
    losses = {"loss_ce": sigmoid_focal_loss(logits, onehot, num_boxes) * logits.shape[1]}
    matched = logits[bidx, sidx]
    if tco.numel() == 0:
        losses["class_error"] = torch.tensor(100.0)
    else:                                                                                  # A2/util/misc.py:436-452
        losses["class_error"] = 100 - (matched.argmax(-1) == tco).float().sum() * (100.0 / tco.numel())
    # boxes :213-234
    sb = boxes[bidx, sidx]
    tb = torch.cat([t["boxes"][j] for t, (_, j) in zip(targets, indices)], dim=0)
    losses["loss_bbox"] = (sb - tb).abs().sum() / num_boxes
    giou = torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(sb), box_cxcywh_to_xyxy(tb)))
    losses["loss_giou"] = (1 - giou).sum() / num_boxes
    # cardinality :199-211
    with torch.no_grad():
        card = (logits.argmax(-1) != logits.shape[-1] - 1).sum(1).float()
        tl = torch.tensor([float(len(t["labels"])) for t in targets])
        losses["cardinality_error"] = (card - tl).abs().mean()
    # vars :264-289 (scalar-mean L1 over all matched, raw log of the possibly negative variance)
    pv = pvars[bidx, sidx]
    lw = (sb[:, 2] - tb[:, 2]).abs().mean() / pv[:, 0].abs() + pv[:, 0].log().abs()
    lh = (sb[:, 3] - tb[:, 3]).abs().mean() / pv[:, 1].abs() + pv[:, 1].log().abs()
    losses["loss_variance"] = ((lw + lh) / num_boxes).sum()
    return losses, indices


def set_criterion_aux(outputs, targets, world_size=1):
    """A2/models/anchor_detr.py:308-350 with aux_loss=True: the last layer's losses plus, for every intermediate decoder layer
    i, its OWN Hungarian matching and the same losses under the key suffix `_i` (class_error is logged for the last layer only,
    `log=False` :343-345).  `aux_outputs` must carry pred_vars (the reference's `_set_aux_loss` :136-140 omits it and its
    loss_variance then raises KeyError -- the golden generator patches that one line).  -> (losses, [indices per layer])."""
    main = {k: v for k, v in outputs.items() if k not in ("aux_outputs", "all_layers", "memory")}
    losses, idx = set_criterion(main, targets, world_size=world_size)
    all_idx = []
    for i, aux in enumerate(outputs["aux_outputs"]):
        l_i, idx_i = set_criterion(aux, targets, world_size=world_size)
        l_i.pop("class_error")
        losses.update({k + f"_{i}": v for k, v in l_i.items()})
        all_idx.append(idx_i)
    all_idx.append(idx)
    return losses, all_idx


def aux_weight_dict(dec_layers=6, base=None):
    """A2/models/anchor_detr.py:419-428."""
    base = dict(base or WEIGHT_DICT)
    wd = dict(base)
    for i in range(dec_layers - 1):
        wd.update({k + f"_{i}": v for k, v in base.items()})
    wd.update({k + "_enc": v for k, v in base.items()})
    return wd


WEIGHT_DICT = {"loss_ce": 2.0, "loss_bbox": 5.0, "loss_giou": 2.0, "loss_variance": 2.0}  # A2/main.py:105-120


def total_loss(losses, weight_dict=WEIGHT_DICT):
    """A2/engine.py:37."""
    return sum(losses[k] * weight_dict[k] for k in losses if k in weight_dict)


# ----------------------------------------------------------------------------- counting rule (a14)
def count_objects(pred_logits, thr=0.5):
    """A2/infer.py:75-81."""
    return (pred_logits.sigmoid()[..., 0] >= thr).sum(-1)


def counting_metrics(pred_counts, gt_counts):
    """A2/eval_all.py:252-270: MAE, RMSE, NAE, SRE."""
    p = torch.as_tensor(pred_counts, dtype=torch.float64)
    g = torch.as_tensor(gt_counts, dtype=torch.float64)
    err = (g - p).abs()
    return {"MAE": err.mean().item(), "RMSE": (err ** 2).mean().sqrt().item(),
            "NAE": (err / g).mean().item(), "SRE": ((err ** 2) / g).mean().sqrt().item()}


def bbox_criterion(outputs, targets):
    """1st-stage BoundingBoxCriterion, A1/models/anchor_detr.py:317-337 (weights loss_wh 1, loss_giou 0.4)."""
    tp = targets["points"].flatten(0, 1)
    sw = outputs["pred_wh"].flatten(0, 1)
    tw = targets["whs"].flatten(0, 1)
    giou = torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(torch.cat([tp, sw], -1)), box_cxcywh_to_xyxy(torch.cat([tp, tw], -1))))
    return {"loss_wh": F.l1_loss(sw, tw), "loss_giou": (1 - giou).sum() / tw.shape[0]}
