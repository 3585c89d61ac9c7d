"""A2/util/box_ops.py:17-67 (device-side, assertion-free so the step stays sync-free; degenerate boxes give inf/nan
exactly as the reference's formulas would after its host-side asserts)."""
import torch


def box_cxcywh_to_xyxy(x):
    x_c, y_c, w, h = x.unbind(-1)
    return torch.stack([(x_c - 0.5 * w), (y_c - 0.5 * h), (x_c + 0.5 * w), (y_c + 0.5 * h)], dim=-1)


def box_xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0), (y1 - y0)], dim=-1)


def box_area(b):
    return (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])


def generalized_box_iou_pairs(b1, b2):
    """Element-wise GIoU of matched pairs [M,4] x [M,4] -> [M]: the diagonal the reference extracts from its
    M x M matrix at A2/models/anchor_detr.py:228-232."""
    a1, a2 = box_area(b1), box_area(b2)
    lt = torch.max(b1[:, :2], b2[:, :2])
    rb = torch.min(b1[:, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = a1 + a2 - inter
    iou = inter / union
    lt2 = torch.min(b1[:, :2], b2[:, :2])
    rb2 = torch.max(b1[:, 2:], b2[:, 2:])
    wh2 = (rb2 - lt2).clamp(min=0)
    area = wh2[:, 0] * wh2[:, 1]
    return iou - (area - union) / area


def generalized_box_iou(boxes1, boxes2):
    """Pairwise [N,M] (API parity with A2/util/box_ops.py:48-67)."""
    a1, a2 = box_area(boxes1), box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = a1[:, None] + a2 - inter
    iou = inter / union
    lt2 = torch.min(boxes1[:, None, :2], boxes2[:, :2])
    rb2 = torch.max(boxes1[:, None, 2:], boxes2[:, 2:])
    wh2 = (rb2 - lt2).clamp(min=0)
    area = wh2[:, :, 0] * wh2[:, :, 1]
    return iou - (area - union) / area
