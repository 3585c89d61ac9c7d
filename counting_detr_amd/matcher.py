"""Hungarian matcher on the device -- drop-in for A2/models/matcher.py:175-251 (OriginalHungarianMatcher, build_matcher).

`forward` keeps the reference contract (list of (index_i, index_j) int64 CPU tensors, index_i ascending).
`match_device` is what SetCriterion uses inside the train step: cost matrix + exact LSAP stay on the GPU, no host
sync (the reference drains the pipeline with `C.cpu()` and solves with scipy on the host every step)."""
import torch
from torch import nn

from . import ops


class OriginalHungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"

    @torch.no_grad()
    def cost_device(self, outputs, plan, tgt_boxes):
        """The cost matrices of A2/models/matcher.py:229-242, per image, in the solver's layout (ops.match_cost)."""
        logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
        return ops.match_cost(logits.detach().float(), boxes.detach().float(), tgt_boxes, plan, float(self.cost_class),
                              float(self.cost_bbox), float(self.cost_giou))

    @torch.no_grad()
    def match_device(self, outputs, targets, plan=None, tgt_boxes=None, cost=None):
        """-> (idx_i [B,Mmax], idx_j [B,Mmax] int64 device, status [B] int32 device, plan).  `tgt_boxes`: the targets' boxes already
        concatenated in plan order (the criterion needs the same tensor: one concatenation per step instead of two).  `cost`: the cost
        matrices when they were computed ahead (`cost_device`; SetCriterion.pre_match)."""
        logits = outputs["pred_logits"]
        B, Q = logits.shape[:2]
        if plan is None:
            plan = ops.MatchPlan([len(t["boxes"]) for t in targets], Q, logits.device)
        if cost is None:
            tgt = tgt_boxes if tgt_boxes is not None else torch.cat([t["boxes"] for t in targets]).to(torch.float32)
            cost = self.cost_device(outputs, plan, tgt)
        idx_i, idx_j, status = ops.lsap(cost, plan)
        return idx_i, idx_j, status, plan

    @torch.no_grad()
    def forward(self, outputs, targets):
        idx_i, idx_j, status, plan = self.match_device(outputs, targets)
        st = status.cpu()
        if (st != 0).any():
            # scipy raises ValueError for NaN / -inf / infeasible cost matrices
            raise ValueError("matrix contains invalid numeric entries" if (st == 2).any() else "cost matrix is infeasible")
        ii, jj = idx_i.cpu(), idx_j.cpu()
        return [(ii[b, : plan.M[b]].clone(), jj[b, : plan.M[b]].clone()) for b in range(plan.B)]


def build_matcher(args):
    return OriginalHungarianMatcher(args.cost_class, args.cost_bbox, args.cost_giou)
