"""1st-stage Counting-DETR (point -> box pseudo-label generator) on the same MI355X kernels -- SURVEY.md row a15.

API mirror of A1/models/anchor_detr.py (A1 = src/CountDETR_147_1st_stage): `build(args) -> (model, criterion,
postprocessors)`, `model(samples, scaled_sample_points) -> {"pred_logits", "pred_wh", "pred_points"}` (:80-113),
`BoundingBoxCriterion` (:317-337: L1 on wh + mean(1 - GIoU) of boxes built from the GT points and the predicted wh; no
Hungarian matcher is involved in stage 1).  Differences to stage 2 (A1/models/transformer.py:60-214): the query embedding is
called `modify_pattern`, there is no variance head, the class bias has ONE element broadcast over 2 logits, the anchor
points are `defined` = the given points (Q = number of points), the backbone output goes through `input_proj` (no
exemplar aggregation).  State dict is key-compatible with the reference's stage-1 model.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import box_ops
from .anchor_detr import PostProcess, _ProjGN
from .backbone import BackboneAgg
from .misc import NestedTensor, nested_tensor_from_tensor_list
from .transformer import Transformer


class AnchorDETRStage1(nn.Module):
    def __init__(self, backbone, transformer, num_feature_levels=1):
        super().__init__()
        assert num_feature_levels == 1
        self.transformer = transformer
        self.num_feature_levels = num_feature_levels
        self.input_proj = nn.ModuleList([_ProjGN(backbone.num_channels[0], transformer.d_model)])
        self.backbone = backbone

    def forward(self, samples, scaled_sample_points):
        if not isinstance(samples, NestedTensor):
            samples = nested_tensor_from_tensor_list(samples)
        images, mask = samples.decompose()
        x = self.backbone.body.forward_nhwc(images)                          # NHWC [B,h,w,2048]
        m = F.interpolate(mask[None].float(), size=x.shape[1:3]).to(torch.bool)[0]
        src = self.input_proj[0](x)
        (cls, xywh, _), _ = self.transformer(src, m, scaled_sample_points)
        return {"pred_logits": cls[-1], "pred_wh": xywh[-1][..., 2:], "pred_points": xywh[-1][..., :2]}


class BoundingBoxCriterion(nn.Module):
    """A1/models/anchor_detr.py:317-337."""

    def __init__(self):
        super().__init__()
        self.weight_dict = {"loss_wh": 1, "loss_giou": 0.4}

    def forward(self, outputs, targets):
        tgt_points = targets["points"].flatten(0, 1)
        src_whs = outputs["pred_wh"].flatten(0, 1)
        tgt_whs = targets["whs"].flatten(0, 1)
        src_boxes = torch.cat([tgt_points, src_whs], dim=-1)
        tgt_boxes = torch.cat([tgt_points, tgt_whs], dim=-1)
        giou = box_ops.generalized_box_iou_pairs(box_ops.box_cxcywh_to_xyxy(src_boxes), box_ops.box_cxcywh_to_xyxy(tgt_boxes))
        return {"loss_wh": F.l1_loss(src_whs, tgt_whs), "loss_giou": (1 - giou).sum() / tgt_whs.shape[0]}


def build(args):
    """A1/models/anchor_detr.py:375-409."""
    from . import _ffi
    _ffi.lib()
    backbone = BackboneAgg(args.lr_backbone > 0, args.dilation)
    transformer = Transformer(d_model=args.hidden_dim, nhead=args.nheads, num_encoder_layers=args.enc_layers,
                              num_decoder_layers=args.dec_layers, dim_feedforward=args.dim_feedforward, dropout=args.dropout,
                              num_feature_levels=args.num_feature_levels, num_query_position=args.num_query_position,
                              num_query_pattern=args.num_query_pattern, spatial_prior=args.spatial_prior,
                              attention_type=args.attention_type, stage=1)
    transformer.all_layer_heads = False
    model = AnchorDETRStage1(backbone, transformer, args.num_feature_levels)
    criterion = BoundingBoxCriterion().to(torch.device(args.device))
    return model, criterion, {"bbox": PostProcess()}


@torch.no_grad()
def generate_pseudo_boxes(model, image, points):
    """A1/engine.py:124-187 core: all GT dots in, one [cx, cy, w, h] pseudo box per dot out (normalised)."""
    model.eval()
    out = model(image, points)
    return torch.cat([points.reshape(1, -1, 2).expand(image.shape[0], -1, -1), out["pred_wh"]], dim=-1)


@torch.no_grad()
def write_pseudo_labels(model, loader, split, output_dir, device="cuda"):
    """The 1st-stage -> 2nd-stage hand-off file (A1/engine.py:124-187): for every image, one pseudo box per annotated dot,
    written as the COCO-style `pseudo_bbox_<split>.json` that the 2nd-stage training reader opens
    (A2/data/fsc147.py:18-19; counting_detr_amd.data.FSC147Dataset): bbox = [cx, cy, w, h] in original pixels (ints),
    file_name = "<im_id>.jpg", ids counted from 1.  `loader` yields dicts with image [1,3,H,W], points [1,P,2] (normalised),
    orig_size [1,2] = (width, height), im_id.  Returns the annotation dict."""
    import json
    import os
    model.eval()
    ann = {"categories": [{"name": "fg", "id": 1}], "images": [], "annotations": []}
    img_id = anno_id = 1
    for ret in loader:
        image, points = ret["image"].to(device), ret["points"].to(device)
        wh = model(image, points)["pred_wh"]
        size = ret["orig_size"].reshape(-1).tolist()                      # (width, height)
        pts = points.reshape(-1, 2).cpu().numpy().copy()
        whs = wh.reshape(-1, 2).cpu().numpy().copy()
        whs[:, 0] *= size[0]; whs[:, 1] *= size[1]
        pts[:, 0] *= size[0]; pts[:, 1] *= size[1]
        for (x_cen, y_cen), (w, h) in zip(pts, whs):
            ann["annotations"].append({"id": anno_id, "image_id": img_id, "area": int(w * h),
                                       "bbox": [int(x_cen), int(y_cen), int(w), int(h)], "category_id": 1, "iscrowd": 0})
            anno_id += 1
        ann["images"].append({"id": img_id, "file_name": str(int(ret["im_id"])) + ".jpg", "height": int(size[1]), "width": int(size[0])})
        img_id += 1
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, "pseudo_bbox_" + split + ".json"), "w") as handle:
        json.dump(ann, handle)
    return ann
