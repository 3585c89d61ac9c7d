"""Train step / epoch and counting inference -- counterpart of A2/engine.py:14-67 (train_one_epoch) and
A2/infer.py:27-122, re-designed for one-process-per-GPU data parallelism on MI355X:

  * all trainable parameters live in ONE flat fp32 arena (views keep the reference's shapes / state-dict keys), their
    gradients in a second arena that the weight-gradient kernels accumulate into directly; clip_grad_norm_(0.1) and AdamW
    (A2/main.py:157-189: lr 1e-4, "backbone" 1e-5, wd 1e-4) are a handful of flat ops instead of ~250 per-tensor ones;
  * the arena is ordered [transformer+proj | layer4 | layer3 | layer2] = the order gradients become final in backward, so
    the data-parallel all-reduce (RCCL over xGMI) runs as 4 large buckets on a side stream, each launched the moment its
    segment is final and overlapped with the remaining backbone backward;
  * the step issues no host sync (device matcher, device loss normaliser), so it can be captured in a HIP graph.
"""
import math
import sys

import torch
import torch.distributed as dist

from . import backbone as _bb
from .misc import NestedTensor, get_world_size, is_dist_avail_and_initialized, nested_tensor_from_tensor_list, reduce_dict

UNUSED_PREFIXES = ("input_proj.",)   # built but never used on the stage-2 path: grad stays None in the reference


def _segment_of(name):
    if "backbone" in name:
        for li in (4, 3, 2):
            if f"layer{li}." in name:
                return {4: 1, 3: 2, 2: 3}[li]
        return 3
    return 0


class FlatGradExchange:
    """Bucketed SUM all-reduce of a flat gradient arena whose segments become final in order 0, 1, 2, ... during backward.
    Each bucket is launched on a side stream the moment its segment is final (event dependency on the compute stream) and
    overlaps with the rest of backward; `finish()` joins the side stream.  On CPU tensors (gloo, tests) it runs inline.
    Four large buckets (58 / 60 / 28 / 5 MB) instead of torch-DDP's 25 MB default: xGMI ring collectives are per-link
    latency/bandwidth bound, fewer and larger is better."""

    def __init__(self, flat_g, seg_bounds):
        self.flat_g, self.seg_bounds = flat_g, list(seg_bounds)
        self.stream = torch.cuda.Stream() if (flat_g.is_cuda and is_dist_avail_and_initialized()) else None
        self.launched = []

    def segment_done(self, seg):
        lo, hi = self.seg_bounds[seg], self.seg_bounds[seg + 1]
        self.launched.append(seg)
        if hi <= lo or get_world_size() < 2:
            return
        buf = self.flat_g[lo:hi]
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                dist.all_reduce(buf)
        else:
            dist.all_reduce(buf)

    def finish(self):
        """Reduce whatever segment was not announced (e.g. a model without the backbone hook), then join."""
        nseg = len(self.seg_bounds) - 1
        for seg in range(nseg):
            if seg not in self.launched:
                self.segment_done(seg)
        self.launched = []
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


class Trainer:
    def __init__(self, model, criterion, args, device=None):
        self.model, self.criterion, self.args = model, criterion, args
        self.device = torch.device(device or args.device)
        self.max_norm = args.clip_max_norm
        self.betas, self.eps, self.wd = (0.9, 0.999), 1e-8, args.weight_decay
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad and not n.startswith(UNUSED_PREFIXES)]
        for n, p in model.named_parameters():
            if n.startswith(UNUSED_PREFIXES):
                p.grad = None
        named.sort(key=lambda np_: _segment_of(np_[0]))            # stable: keeps definition order inside a segment
        self.names = [n for n, _ in named]
        sizes = [p.numel() for _, p in named]
        total = sum(sizes)
        self.flat_p = torch.zeros(total, device=self.device, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=self.device, dtype=torch.float32)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.lr_vec = torch.zeros(total, device=self.device, dtype=torch.float32)   # per-element base lr
        self.seg_bounds = [0, 0, 0, 0, 0]
        off = 0
        for (n, p), sz in zip(named, sizes):
            pv = self._view_like(self.flat_p[off:off + sz], p)
            pv.copy_(p.data)
            p.data = pv
            p.grad = self._view_like(self.flat_g[off:off + sz], p)
            lr = args.lr
            if any(k in n for k in args.lr_backbone_names):
                lr = args.lr_backbone
            elif any(k in n for k in args.lr_linear_proj_names):
                lr = args.lr * args.lr_linear_proj_mult
            self.lr_vec[off:off + sz] = lr
            off += sz
            self.seg_bounds[_segment_of(n) + 1] = off
        for i in range(1, 5):
            self.seg_bounds[i] = max(self.seg_bounds[i], self.seg_bounds[i - 1])
        # device-resident optimizer scalars (graph-replay safe): [step count, StepLR factor, last grad norm, spare]
        self.opt_state = torch.tensor([0.0, 1.0, 0.0, 0.0], device=self.device)
        self.sumsq = torch.zeros(1, device=self.device)
        self.sumsq_ws = torch.zeros(2048, device=self.device)       # CDETR_SUMSQ_WS_FLOATS: per-block partial sums
        self.epoch = 0
        self._graph = None
        self._static = None
        order = ("loss_ce", "class_error", "cardinality_error", "loss_bbox", "loss_giou", "loss_variance")
        self._w6 = torch.tensor([float(criterion.weight_dict.get(k, 0.0)) for k in order], device=self.device)
        self.mirror = self._build_mirror(named)
        self.exchange = FlatGradExchange(self.flat_g, self.seg_bounds)
        _bb.set_backward_hook(self._segment_done if get_world_size() > 1 else None)

    def _build_mirror(self, named):
        """Weight images (ops.WeightMirror), rewritten once per step: k-contiguous transposes of every trainable matrix that
        is a data-gradient operand (backbone convs with the FrozenBN scale folded in, 1x1 projections, linears with both dims
        >= 32) and pre-split forward operands of every conv / linear weight with K % 32 == 0 (frozen layers included)."""
        from . import ops
        entries, fwd, seen = [], [], set()
        for m in self.model.modules():
            if isinstance(m, _bb.Bottleneck):
                pairs = [(m.conv1, m.bn1), (m.conv2, m.bn2), (m.conv3, m.bn3)]
                if m.downsample is not None:
                    pairs.append((m.downsample[0], m.downsample[1]))
                for conv, bn in pairs:
                    w = conv.weight.data
                    seen.add(w.data_ptr())
                    if not w.is_cuda:
                        continue
                    if conv.weight.requires_grad:
                        entries.append((w, bn.affine()[0]))
                    if w.shape[1] % 32 == 0:
                        fwd.append((w, bn.affine()[0]))
        for _, p in named:
            if p.data_ptr() in seen or not p.is_cuda or min(p.shape[:2] if p.dim() >= 2 else (0,)) < 32:
                continue
            if p.dim() == 2 or (p.dim() == 4 and p.shape[2] == 1 and p.shape[3] == 1):
                entries.append((p.data, None))
                if p.shape[1] % 32 == 0:
                    fwd.append((p.data, None))
        return ops.WeightMirror(entries, fwd) if (entries or fwd) else None

    @staticmethod
    def _view_like(chunk, p):
        if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
            co, ci, kh, kw = p.shape
            return chunk.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return chunk.view(p.shape)

    # ------------------------------------------------------------------ data-parallel gradient exchange
    def _segment_done(self, seg):
        """Called from the backbone's backward: `seg` (0 = everything above the backbone, 1..3 = layer4..layer2) is final."""
        from . import ops
        ops.wgrad_flush()           # queued parameter gradients of the finished segment must land before its bucket ships
        self.exchange.segment_done(seg)

    def _finish_allreduce(self):
        self.exchange.finish()      # the 1/world average is folded into cdetr_adamw_step (grad_div)

    # ------------------------------------------------------------------ optimizer (flat clip + AdamW)
    def _optimizer_step(self):
        """clip_grad_norm_(max_norm) + AdamW over the flat arenas: one reduction pass + one fused update pass
        (cdetr_sumsq / cdetr_adamw_step); step count, StepLR factor and the norm stay on the device."""
        from . import _ffi
        n = self.flat_p.numel()
        b1, b2 = self.betas
        st = _ffi.stream_ptr()
        _ffi.check(_ffi.lib().cdetr_sumsq(self.flat_g.data_ptr(), n, self.sumsq.data_ptr(), self.sumsq_ws.data_ptr(), st), "cdetr_sumsq")
        _ffi.check(_ffi.lib().cdetr_adamw_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                                               self.exp_avg_sq.data_ptr(), self.lr_vec.data_ptr(), n, self.sumsq.data_ptr(),
                                               self.opt_state.data_ptr(), float(self.max_norm), b1, b2, self.eps, self.wd,
                                               1.0 / get_world_size(), st), "cdetr_adamw_step")
        return self.opt_state[2]

    def state_dict(self):
        """Optimizer state in a self-describing form (flat moments + names/offsets) for the checkpoint's "optimizer" key."""
        sizes = [dict(self.model.named_parameters())[n].numel() for n in self.names]
        return {"names": self.names, "sizes": sizes, "exp_avg": self.exp_avg.detach().cpu(),
                "exp_avg_sq": self.exp_avg_sq.detach().cpu(), "state": self.opt_state.detach().cpu(), "epoch": self.epoch}

    def load_state_dict(self, sd):
        assert sd["names"] == self.names, "optimizer state does not match this model's parameter layout"
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.opt_state.copy_(sd["state"])
        self.epoch = int(sd["epoch"])

    def lr_scheduler_step(self):
        """StepLR(step=lr_drop, gamma=0.1), stepped once per epoch (A2/main.py:189,219)."""
        self.epoch += 1
        self.opt_state[1] = 0.1 ** (self.epoch // self.args.lr_drop)

    # ------------------------------------------------------------------ one step
    def _num_boxes(self, targets):
        """Loss normaliser: sum of target counts over all ranks / world, clamped at 1 (A2/models/anchor_detr.py:321-325).
        Issued before the forward: it scales the loss, so it must be known before the loss is formed."""
        nb = float(sum(len(t["boxes"]) for t in targets))
        if get_world_size() > 1:
            t = torch.tensor([nb], dtype=torch.float32, device=self.device)
            dist.all_reduce(t)
            return torch.clamp(t / get_world_size(), min=1)[0]
        return max(nb, 1.0)

    def _fwd_bwd(self, images, mask, rects, targets, num_boxes):
        from .misc import NestedTensor
        from . import ops
        self.flat_g.zero_()
        # weight images (transposes for the data gradients, pre-split operands for both passes): one launch, armed for this step only
        if self.mirror is not None:
            self.mirror.refresh()
        ops.MIRROR = self.mirror
        try:
            outputs, _ = self.model(NestedTensor(images, mask), rects=rects)
            loss_dict = self.criterion(outputs, targets, num_boxes=num_boxes)
            wd = self.criterion.weight_dict
            vec = getattr(self.criterion, "last_vec", None)
            if vec is not None:          # fused criterion: one weighted reduction of its loss vector
                losses = (vec * self._w6).sum()                                       # A2/engine.py:37
            else:
                losses = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
            with ops.wgrad_queue():      # small parameter gradients outside the fused layer nodes (heads, positional MLPs): grouped
                losses.backward()
        finally:
            ops.MIRROR = None
        out = dict(loss_dict)
        out["loss"] = losses.detach()
        return out

    def _step_impl(self, images, mask, rects, targets, num_boxes):
        out = self._fwd_bwd(images, mask, rects, targets, num_boxes)
        if get_world_size() > 1:
            self._finish_allreduce()
        out["grad_norm"] = self._optimizer_step()
        return out

    def train_step(self, samples, rects, targets):
        """Eager step.  samples: [B,3,H,W] tensor, list of [3,h,w] tensors, or NestedTensor.  Returns device scalars.
        With world_size > 1 the gradient all-reduce runs as 4 buckets on a side stream, overlapped with backward."""
        nt = samples if hasattr(samples, "decompose") else nested_tensor_from_tensor_list(samples)
        images, mask = nt.decompose()
        return self._step_impl(images, mask, rects, targets, self._num_boxes(targets))

    # ------------------------------------------------------------------ HIP-graph replay of the step
    def _dry_run(self, st):
        """Forward + criterion without autograd: builds every lazily cached device table (frozen-BN folds, padded stem
        weight, match plans, LDS attributes) OUTSIDE the capture; parameters are untouched."""
        from .misc import NestedTensor
        with torch.no_grad():
            outputs, _ = self.model(NestedTensor(st["images"], st["mask"]), rects=st["rects"])
            self.criterion(outputs, st["targets"], num_boxes=1.0)

    def capture(self, samples, rects, targets, warmup=0):
        """Capture (record, not run) the step for fixed shapes / target counts; `replay()` executes it.
        world_size == 1: ONE graph = zero-grad + forward + device matcher + losses + backward + clip + AdamW.
        world_size  > 1: graph A = everything up to the gradients, then ONE eager flat all-reduce (RCCL) on the compute
        stream, then graph B = clip + AdamW (no collective inside a capture)."""
        nt = samples if hasattr(samples, "decompose") else nested_tensor_from_tensor_list(samples)
        images, mask = nt.decompose()
        st = {"images": images.clone(), "mask": mask.clone(), "rects": rects.clone(),
              "targets": [{k: v.clone() for k, v in t.items()} for t in targets]}
        world = get_world_size()
        nb0 = self._num_boxes(targets)
        st["num_boxes"] = nb0.clone() if torch.is_tensor(nb0) else nb0
        hook = _bb._BACKWARD_HOOK
        _bb.set_backward_hook(None)                # no collectives inside the capture
        try:
            g_a, g_b, out = self._capture_graphs(st, world, warmup)
        finally:                                   # a failed capture must leave the stream-ordered step intact
            _bb.set_backward_hook(hook)
        self._graph, self._graph_b, self._static, self._static_out = g_a, g_b, st, out
        return out

    def _capture_graphs(self, st, world, warmup):
        self._dry_run(st)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._fwd_bwd(st["images"], st["mask"], st["rects"], st["targets"], st["num_boxes"])
                if world > 1:
                    dist.all_reduce(self.flat_g)
                self._optimizer_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g_a = torch.cuda.CUDAGraph()
        if world == 1:
            with torch.cuda.graph(g_a):
                out = self._fwd_bwd(st["images"], st["mask"], st["rects"], st["targets"], st["num_boxes"])
                out["grad_norm"] = self._optimizer_step()
            g_b = None
        else:
            with torch.cuda.graph(g_a):
                out = self._fwd_bwd(st["images"], st["mask"], st["rects"], st["targets"], st["num_boxes"])
            g_b = torch.cuda.CUDAGraph()           # (capturing records work, it does not run it)
            with torch.cuda.graph(g_b, pool=g_a.pool()):
                out["grad_norm"] = self._optimizer_step()
        return g_a, g_b, out

    def replay(self, samples=None, rects=None, targets=None):
        st = self._static
        if samples is not None:
            nt = samples if hasattr(samples, "decompose") else nested_tensor_from_tensor_list(samples)
            images, mask = nt.decompose()
            st["images"].copy_(images)
            st["mask"].copy_(mask)
            st["rects"].copy_(rects)
            for s_t, t in zip(st["targets"], targets):
                for k in s_t:
                    s_t[k].copy_(t[k])
            if torch.is_tensor(st["num_boxes"]):
                st["num_boxes"].copy_(self._num_boxes(targets))
        self._graph.replay()
        if self._graph_b is not None:
            dist.all_reduce(self.flat_g)
            self._graph_b.replay()
        return self._static_out


def train_one_epoch(trainer, data_loader, epoch, print_freq=100, log=print):
    """A2/engine.py:14-67: iterate, step, abort on a non-finite loss.  The loss is read back every `print_freq`
    iterations only (the reference syncs every step with .item())."""
    trainer.model.train()
    trainer.criterion.train()
    stats = {}
    n = 0
    for it, ret in enumerate(data_loader):
        samples = NestedTensor(ret["image"], ret["mask"]) if "mask" in ret else ret["image"]     # data.collate pads + masks
        out = trainer.train_step(samples, ret["ex_rects"], ret["targets"])
        if it % print_freq == 0:
            red = reduce_dict({k: v for k, v in out.items() if torch.is_tensor(v)})
            vals = {k: float(v) for k, v in red.items()}
            if not math.isfinite(vals["loss"]):
                log("Loss is {}, stopping training".format(vals["loss"]))
                log(vals)
                sys.exit(1)
            for k, v in vals.items():
                stats[k] = stats.get(k, 0.0) + v
            n += 1
            log(f"Epoch: [{epoch}] it {it} " + "  ".join(f"{k}: {v:.4f}" for k, v in vals.items()))
    return {k: v / max(n, 1) for k, v in stats.items()}


@torch.no_grad()
def count_objects(model, samples, rects, threshold=0.5):
    """The counting rule of A2/infer.py:75-81: #queries with sigmoid(logit[...,0]) >= 0.5; also returns the kept boxes."""
    model.eval()
    outputs, ref_points = model(samples, rects=rects)
    prob = outputs["pred_logits"].sigmoid()[..., 0]
    keep = prob >= threshold
    return keep.sum(-1), keep, outputs, ref_points


def counting_metrics(pred_counts, gt_counts):
    """MAE / RMSE / NAE / SRE exactly as A2/eval_all.py:252-270."""
    cnt = len(gt_counts)
    sae = sse = nae = sre = 0.0
    for p, g in zip(pred_counts, gt_counts):
        err = abs(float(g) - float(p))
        sae += err
        sse += err ** 2
        nae += err / g
        sre += err ** 2 / g
    return {"MAE": sae / cnt, "RMSE": (sse / cnt) ** 0.5, "NAE": nae / cnt, "SRE": (sre / cnt) ** 0.5}
