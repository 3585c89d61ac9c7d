"""ResNet-50-DC5 backbone with frozen BN and exemplar feature aggregation on the HIP implicit-GEMM kernels.

API mirror of the reference's A2/models/backbone.py (FrozenBatchNorm2d :22-60, BackboneAgg.extract_feature :116-145,
build_backbone :174-179) and A2/models/resnet.py (Bottleneck :105-160, ResNet :163-280): same parameter/buffer names
and shapes (state-dict compatible), same math.  Differences are in HOW it runs:
  * activations are NHWC fp32; every conv is one implicit-GEMM launch with FrozenBN folded into the weight load and
    bias / residual / ReLU fused in the epilogue (no im2col buffer, no separate BN / ReLU / add passes);
  * stem + layer1 are frozen and run without autograd; layer2-4 are ONE autograd node whose hand-scheduled backward
    fuses each ReLU mask, BN scale and residual add into the data-gradient epilogues;
  * the exemplar centres are computed on the device (the reference does 6 host syncs via int() on device scalars).
"""
import torch
from torch import nn

from . import ops

RESNET50_LAYERS = (3, 4, 6, 3)


class ConvW(nn.Module):
    """Holds a conv weight with the reference's logical shape in channels_last memory ([Cout][kh][kw][Cin])."""

    def __init__(self, cin, cout, k):
        super().__init__()
        w = torch.empty(cout, cin, k, k).contiguous(memory_format=torch.channels_last)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")     # A2/models/resnet.py:231-233
        self.weight = nn.Parameter(w)


class FrozenBatchNorm2d(nn.Module):
    """A2/models/backbone.py:22-60 -- buffers only; `affine()` returns the folded (scale, bias)."""

    def __init__(self, n, eps=1e-5):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.eps = eps
        self._cache = None

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        state_dict.pop(prefix + "num_batches_tracked", None)
        self._cache = None
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def affine(self):
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version,
               self.weight.data_ptr())
        if self._cache is None or self._cache[0] != key:
            with torch.no_grad():
                scale = self.weight * (self.running_var + self.eps).rsqrt()
                bias = self.bias - self.running_mean * scale
            self._cache = (key, scale.contiguous(), bias.contiguous())
        return self._cache[1], self._cache[2]

    def forward(self, x):   # NCHW reference semantics (not used by the fused path; kept for API parity)
        s, b = self.affine()
        return x * s.reshape(1, -1, 1, 1) + b.reshape(1, -1, 1, 1)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, dilation, has_down):
        super().__init__()
        self.conv1, self.bn1 = ConvW(inplanes, planes, 1), FrozenBatchNorm2d(planes)
        self.conv2, self.bn2 = ConvW(planes, planes, 3), FrozenBatchNorm2d(planes)
        self.conv3, self.bn3 = ConvW(planes, planes * 4, 1), FrozenBatchNorm2d(planes * 4)
        self.downsample = nn.Sequential(ConvW(inplanes, planes * 4, 1), FrozenBatchNorm2d(planes * 4)) if has_down else None
        self.stride, self.dilation = stride, dilation

    def forward_fused(self, x, save=None, x16=None, twins=False, twin_out=False, out_to=None, out_groups=False):
        """twins: every activation a weight gradient of this block will read (a1, a2) and the block's output (the next block's x) also
        leave their producing epilogue as a bf16 copy (ops.bf16_twins); x16 = the twin of x, from the previous block.
        twin_out: only the output gets a twin (the last frozen block in front of the trainable ones).
        out_to = (y, y16): caller-owned buffers the block's output (and its twin) are written to (fixed addresses: ResNetBody.frozen_stage).
        out_groups: the output leaves conv3's epilogue as interleaved split-bf16 groups (ops.Groups, 4 bytes per element, no fp32 tensor):
        the next block's conv1 streams it in full lines on the direct-to-LDS kernel, its conv3 adds hi + lo as the residual; the twin still
        serves the backward (weight-gradient operand, ReLU mask).  `x` may itself be such a tensor."""
        twin_out = twin_out or twins
        s1, b1 = self.bn1.affine()
        s2, b2 = self.bn2.affine()
        s3, b3 = self.bn3.affine()
        idn, br = x, None
        if self.downsample is not None:       # the shortcut convolution: beside conv1 -> conv2 (ops.fork_branch), joined before the residual add
            sd, bd = self.downsample[1].affine()
            wd = self.downsample[0].weight
            if ops.BRANCH_BESIDE and x.is_cuda:
                Hd, Wd = (x.shape[1] - 1) // self.stride + 1, (x.shape[2] - 1) // self.stride + 1
                idn = torch.empty((x.shape[0], Hd, Wd, wd.shape[0]), device=x.device, dtype=torch.float32)
                with ops.fork_branch() as br:
                    ops.conv_fwd(x, wd, sd, bd, stride=self.stride, out=idn)
            else:
                idn = ops.conv_fwd(x, wd, sd, bd, stride=self.stride)
        a1 = ops.conv_fwd(x, self.conv1.weight, s1, b1, relu=True, twin=twins)
        a1, a1_16 = a1 if twins else (a1, None)
        # the 3x3's output feeds the EXPANDING 1x1 (K = planes, N = 4 planes): the one shape class where the direct-to-LDS kernel beats the
        # register-staged one in the split-bf16 forward (tools/dl_sweep.py: 1.05-1.2x) -- its A operand is the smallest tensor of the block, so
        # the lo plane costs next to nothing; the hi plane is the backward's twin anyway
        a2l = None
        if ops.expand_planes() and x.is_cuda:
            a2, a2_16, a2l = ops.conv_fwd(a1, self.conv2.weight, s2, b2, stride=self.stride, pad=self.dilation, dil=self.dilation, relu=True, split=True)
        else:
            a2 = ops.conv_fwd(a1, self.conv2.weight, s2, b2, stride=self.stride, pad=self.dilation, dil=self.dilation, relu=True, twin=twins)
            a2, a2_16 = a2 if twins else (a2, None)
        if br is not None:
            br.join()
        out = ops.conv_fwd(a2, self.conv3.weight, s3, b3, relu=True, resid=idn, twin=twin_out, xs=(a2_16, a2l) if a2l is not None else None,
                           out=out_to[0] if out_to is not None else None, out16=out_to[1] if out_to is not None else None, out_groups=out_groups)
        out, out16 = out if twin_out else (out, None)
        if not twins:
            a2_16 = None
        if save is not None:
            save.append((x, a1, a2, out, x16, a1_16, a2_16, out16))
        return (out, out16) if twin_out else out

    def backward_fused(self, saved, dz, need_dx, dz16=None):
        """dz = gradient w.r.t. the pre-ReLU output of this block (already masked by out > 0); dz16 its bf16 twin or None.
        Returns the gradient w.r.t. the pre-ReLU output of the PREVIOUS block (masked by x > 0) -- with its twin when dz16 is given."""
        x, a1, a2, out, x16, a1_16, a2_16, _ = saved
        tw = dz16 is not None
        s1, _ = self.bn1.affine()
        s2, _ = self.bn2.affine()
        s3, _ = self.bn3.affine()
        w1, w2, w3 = self.conv1.weight, self.conv2.weight, self.conv3.weight
        st, dl = self.stride, self.dilation
        d_idn, br = dz, None
        if self.downsample is not None and need_dx:      # the shortcut's data gradient: beside the main branch's chain, joined before the last add
            wd0 = self.downsample[0].weight
            sd0, _ = self.downsample[1].affine()
            if ops.BRANCH_BESIDE and dz.is_cuda:
                d_idn = torch.empty(x.shape, device=dz.device, dtype=torch.float32)
                with ops.fork_branch() as br:
                    ops.conv_dgrad(dz, wd0, sd0, x.shape[1:3], stride=st, dz16=dz16, out=d_idn)
            else:
                d_idn = ops.conv_dgrad(dz, wd0, sd0, x.shape[1:3], stride=st, dz16=dz16)
        if w3.requires_grad:
            ops.conv_wgrad_(dz, a2, w3, s3, dz16=dz16, x16=a2_16 if tw else None)
        # dz2 / dz1 are read by the next data gradient and one weight gradient only -- plain-bf16 contractions fed by the twin: no fp32 copy
        dz2 = ops.conv_dgrad(dz, w3, s3, a2.shape[1:3], gate=a2, twin=tw, dz16=dz16, gate16=a2_16 if tw else None, twin_only=True)      # masked by relu(a2)
        dz2, dz2_16 = dz2 if tw else (dz2, None)
        if w2.requires_grad:
            ops.conv_wgrad_(dz2, a1, w2, s2, stride=st, pad=dl, dil=dl, dz16=dz2_16, x16=a1_16 if tw else None)
        dz1 = ops.conv_dgrad(dz2, w2, s2, a1.shape[1:3], stride=st, pad=dl, dil=dl, gate=a1, twin=tw, dz16=dz2_16, gate16=a1_16 if tw else None,
                             twin_only=True)
        dz1, dz1_16 = dz1 if tw else (dz1, None)
        if w1.requires_grad:
            ops.conv_wgrad_(dz1, x, w1, s1, dz16=dz1_16, x16=x16 if tw else None)
        if self.downsample is not None:
            wd = self.downsample[0].weight
            sd, _ = self.downsample[1].affine()
            if wd.requires_grad:
                ops.conv_wgrad_(dz, x, wd, sd, stride=st, dz16=dz16, x16=x16 if tw else None)
        if not need_dx:
            return None
        if br is not None:
            br.join()
        # x = relu(previous pre-activation): the gate applies the previous block's ReLU mask in the same epilogue
        return ops.conv_dgrad(dz1, w1, s1, x.shape[1:3], gate=x, resid=d_idn, twin=tw, dz16=dz1_16, gate16=x16 if tw else None)


GROUPS_PLANES = tuple(64 << (int(c) - 1) for c in __import__("os").environ.get("CDETR_GROUPS_LAYERS", "2,3,4").split(",") if c.strip())


def groups_between(blocks, i, x):
    """Whether block i's output goes to block i + 1 as interleaved groups (ops.Groups) instead of an fp32 tensor: inside one backbone stage only
    (the stage's last block feeds a strided / long-reduction shortcut convolution and, at the trunk's end, the projection: fp32), stride-1 block,
    and ops.use_groups (split-bf16 forward with weight images, >= 4096 pixel rows).  Measured per shape: profiles/r4_fwd_split_groups.txt."""
    if i + 1 >= len(blocks) or blocks[i + 1].downsample is not None:
        return False
    blk = blocks[i]
    planes = blk.conv3.weight.shape[1]
    if planes not in GROUPS_PLANES or not x.is_cuda:
        return False
    Nb, H, W, _ = x.shape
    Ho, Wo = (H - 1) // blk.stride + 1, (W - 1) // blk.stride + 1
    return ops.use_groups(Nb * Ho * Wo, 4 * planes)


_BACKWARD_HOOK = None
_DEFER = None      # a list while the trainer wants the trunk's backward as an explicit object (TrunkBackward) instead of inline


def set_backward_hook(fn):
    """fn(segment) is called from the trunk's backward when a gradient segment is final: 0 = every parameter above the
    backbone, 1 / 2 / 3 = layer4 / layer3 / layer2 (used by the trainer to launch bucketed all-reduces early)."""
    global _BACKWARD_HOOK
    _BACKWARD_HOOK = fn


class defer_trunk_backward:
    """Context manager: inside it `_TrunkFn.backward` only applies the last ReLU mask and parks the rest of its work as a
    `TrunkBackward` in `.pending`; the caller then runs `pending[0].run(segment)` for segment 1 (layer4), 2 (layer3),
    3 (layer2) itself, on its own thread.  The data-parallel graph replay needs this: each segment is captured as its own HIP
    graph so that the RCCL all-reduce of a finished gradient bucket can be issued BETWEEN two graph launches, on a side
    stream, and overlap the remaining segments (a collective cannot sit inside a captured graph, and stream capture cannot
    be ended / restarted from the autograd engine's thread in the middle of a backward)."""

    def __enter__(self):
        global _DEFER
        assert _DEFER is None
        self.pending = _DEFER = []
        return self

    def __exit__(self, *exc):
        global _DEFER
        _DEFER = None
        return False


class TrunkBackward:
    def __init__(self, blocks, saved, dz, dz16=None):
        self.blocks, self.saved, self.dz, self.dz16 = blocks, saved, dz, dz16
        n2, n3 = RESNET50_LAYERS[1], RESNET50_LAYERS[2]
        n = len(blocks)
        self.ranges = {1: (n - 1, n2 + n3), 2: (n2 + n3 - 1, n2), 3: (n2 - 1, 0)}      # block indices, high -> low inclusive

    def run(self, seg):
        hi, lo = self.ranges[seg]
        for k in range(hi, lo - 1, -1):
            r = self.blocks[k].backward_fused(self.saved[k], self.dz, need_dx=(k > 0), dz16=self.dz16)
            self.dz, self.dz16 = r if (self.dz16 is not None and r is not None) else (r, None)
            self.saved[k] = None
            if ops.WGRAD_EVERY and (k - lo) % ops.WGRAD_EVERY == 0:
                ops.wgrad_flush(overlap=True)    # these blocks' weight gradients run beside the next blocks' data-gradient chain


class _TrunkFn(torch.autograd.Function):
    """layer2..layer4 as one autograd node (explicit backward schedule instead of ~100 tiny autograd nodes)."""

    @staticmethod
    def forward(ctx, x, anchor, blocks, x16):
        saved = []
        twins = x16 is not None
        with torch.no_grad():
            for i, blk in enumerate(blocks):
                og = twins and groups_between(blocks, i, x)
                if twins:
                    x, x16 = blk.forward_fused(x, saved, x16=x16, twins=True, out_groups=og)
                else:
                    x = blk.forward_fused(x, saved)
        ctx.blocks, ctx.saved_acts = blocks, saved
        ctx.twins = twins and ops.bf16_twins()
        return x

    @staticmethod
    def backward(ctx, d_out):
        blocks, saved = ctx.blocks, ctx.saved_acts
        out_last = saved[-1][3]
        dz16 = None
        if ctx.twins:
            dz, dz16 = ops.relu_mask(out_last, d_out, twin=True)
        else:
            dz = ops.relu_mask(out_last, d_out)      # gradient through the last block's ReLU: one pass
        if _DEFER is not None:                       # the trainer runs the segments itself (graph replay with bucketed exchange)
            _DEFER.append(TrunkBackward(blocks, saved, dz, dz16))
            ctx.saved_acts = None
            return None, None, None, None
        hook = _BACKWARD_HOOK
        if hook is not None:
            hook(0)          # autograd runs this node last: every gradient above the backbone is final
        tb = TrunkBackward(blocks, saved, dz, dz16)
        for seg in (1, 2, 3):
            tb.run(seg)
            if hook is not None:
                hook(seg)    # layer4 / layer3 / layer2 done
        ctx.saved_acts = None
        return None, None, None, None


class ResNetBody(nn.Module):
    """Children named as torchvision's resnet50 up to layer4 (state-dict keys `conv1.weight`, `layer3.2.bn1.bias`...)."""

    def __init__(self, dilation=True):
        super().__init__()
        self.conv1, self.bn1 = ConvW(3, 64, 7), FrozenBatchNorm2d(64)
        inplanes, cur_dil = 64, 1
        for li, (planes, nblocks) in enumerate(zip((64, 128, 256, 512), RESNET50_LAYERS), start=1):
            stride = 1 if li == 1 else 2
            prev_dil = cur_dil
            if li == 4 and dilation:            # DC5: A2/models/backbone.py:153-155, resnet.py:217-224
                cur_dil *= stride
                stride = 1
            blocks = []
            for b in range(nblocks):
                blocks.append(Bottleneck(inplanes, planes, stride if b == 0 else 1, prev_dil if b == 0 else cur_dil, b == 0))
                inplanes = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self._stem_w4 = None
        self._stem_wr = None
        self.frozen_input = None      # see forward_nhwc
        self.stem_packed = True       # row-packed stem (stem_rows); False = one tap per filter element (stem_weight4)

    def stem_weight4(self):
        """conv1 weight padded to 4 input channels ([64][7][7][4]) so a tap is one aligned 16-byte K-run."""
        w = self.conv1.weight
        key = (w._version, w.data_ptr())
        if self._stem_w4 is None or self._stem_w4[0] != key:
            with torch.no_grad():
                w4 = torch.zeros(64, 4, 7, 7, device=w.device).contiguous(memory_format=torch.channels_last)
                w4[:, :3] = w
            self._stem_w4 = (key, w4)
        return self._stem_w4[1]

    def stem_weight_rows(self):
        """conv1 weight as 7 row taps of K = 32: [64][ky][kx*4 + c] (kx < 7, c < 3; the 8th pixel and the 4th channel are zero).
        A filter row of the 4-channel NHWC image is 28 contiguous floats, so the 7x7 stem becomes a 7-tap convolution whose
        k-tiles are full 32-float runs -> the MFMA tile kernel (K % 32 == 0) instead of the generic 4-floats-per-tap path."""
        w = self.conv1.weight
        key = (w._version, w.data_ptr())
        if self._stem_wr is None or self._stem_wr[0] != key:
            with torch.no_grad():
                wr = torch.zeros(64, 7, 8, 4, device=w.device)                # [o][ky][kx][c]
                wr[:, :, :7, :3] = w.permute(0, 2, 3, 1)
                wr = wr.reshape(64, 7, 1, 32).permute(0, 3, 1, 2)             # logical [64, 32, 7, 1], channels_last memory
            self._stem_wr = (key, wr)
        return self._stem_wr[1]

    def stem_rows(self, images):
        """conv1 + bn1 + relu through the row-packed form: the image is written once into a zero-padded NHWC buffer (3 rows / 3
        columns in front, enough behind for the 8-pixel runs), a GEMM row is the 32-float run starting at padded pixel (2y+ky, 2x)."""
        B, _, H, W = images.shape
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        Ha, Wa = H + 6, W + 8 + (W & 1)
        xp = torch.empty((B, Ha, Wa, 4), device=images.device, dtype=torch.float32)
        ops.check(ops.lib().cdetr_stem_pack(ops.ptr(images.contiguous()), ops.ptr(xp), B, H, W, Ha, Wa, 3, 3, ops.stream_ptr()), "cdetr_stem_pack")
        s, b = self.bn1.affine()
        wr = self.stem_weight_rows()
        y = torch.empty((B, Ho, Wo, 64), device=images.device, dtype=torch.float32)
        g = ops._geom(ops._ffi.ROWS_CONV_FWD, Ha, Wa, Ho, Wo, 7, 1, 2, 0, 1)
        ops.gemm_raw(xp, 4, wr, 7 * 32, y, 64, B * Ho * Wo, 64, 32, taps=7, w_scale=s, bias=b, relu=True, geom=g)
        return y

    def frozen_stage(self, images, twins, out_to=None):
        """The part of the trunk that never trains and whose input needs no gradient (A2/models/backbone.py:93-95: conv1 + layer1
        frozen): stem + max-pool + layer1, images [B,3,H,W] -> (x NHWC [B,H/4,W/4,256], bf16 twin of x or None).
        It depends on nothing a training step updates, so a trainer may run it for the NEXT batch while the current step is in its
        latency-bound phases (engine.Trainer: frozen-stage prefetch).  out_to = (x, x16) caller-owned output buffers."""
        B, _, H, W = images.shape
        with torch.no_grad():
            if self.stem_packed:
                x = self.stem_rows(images)
            else:
                x = torch.zeros((B, H, W, 4), device=images.device, dtype=torch.float32)
                x[..., :3] = images.permute(0, 2, 3, 1)
                s, b = self.bn1.affine()
                x = ops.conv_fwd(x, self.stem_weight4(), s, b, stride=2, pad=3, relu=True)
            x16 = None
            x = ops.maxpool3x3s2(x)
            for i, blk in enumerate(self.layer1):
                if i == len(self.layer1) - 1 and (twins or out_to is not None):
                    want16 = twins or (out_to is not None and out_to[1] is not None)      # (inference: out_to = (x, None), no twin)
                    r = blk.forward_fused(x, twin_out=want16, out_to=out_to)
                    x, x16 = r if want16 else (r, None)
                else:
                    x = blk.forward_fused(x)
        return x, x16

    @staticmethod
    def frozen_out_hw(H, W):
        """Spatial size of the frozen stage's output: 7x7 / 2 stem (pad 3), then 3x3 / 2 max-pool (pad 1)."""
        return ((H - 1) // 2 + 1 - 1) // 2 + 1, ((W - 1) // 2 + 1 - 1) // 2 + 1

    def frozen_stage_is_frozen(self):
        return not any(p.requires_grad for m in (self.conv1, self.layer1) for p in m.parameters())

    def forward_nhwc(self, images):
        """images [B,3,H,W] (NCHW, as the reference API) -> layer4 features NHWC [B,H/16,W/16,2048].
        `self.frozen_input` = (x, x16) (set by a trainer around its captured forward): the frozen stage's output for THESE images has
        already been computed into those buffers (frozen_stage(..., out_to=...)) -- it is not run again."""
        B, _, H, W = images.shape
        blocks = list(self.layer2) + list(self.layer3) + list(self.layer4)
        train = torch.is_grad_enabled() and any(p.requires_grad for blk in blocks for p in blk.parameters())
        twins = train and ops.bf16_twins()            # layer1's output is layer2's first weight-gradient operand: it gets a twin too
        if self.frozen_input is not None:
            x, x16 = self.frozen_input
            assert tuple(x.shape) == (B,) + self.frozen_out_hw(H, W) + (256,), (tuple(x.shape), (B, H, W))
        else:
            x, x16 = self.frozen_stage(images, twins)
        anchor = self.layer4[-1].conv3.weight
        if train:
            return _TrunkFn.apply(x, anchor, blocks, x16)
        with torch.no_grad():
            for i, blk in enumerate(blocks):
                x = blk.forward_fused(x, out_groups=groups_between(blocks, i, x))
        return x


class BackboneAgg(nn.Module):
    """A2/models/backbone.py:90-159 (single feature level): body + exemplar aggregation."""

    def __init__(self, train_backbone=True, dilation=True):
        super().__init__()
        self.body = ResNetBody(dilation)
        for name, p in self.body.named_parameters():
            if not train_backbone or ("layer2" not in name and "layer3" not in name and "layer4" not in name):
                p.requires_grad_(False)                                   # :93-95
        self.strides = [16 if dilation else 32]
        self.num_channels = [2048]
        # "per_image" (default): image b is conditioned on rects[b], scaled by ITS un-padded extent -- what a batched trainer
        # needs.  "reference": A2/models/backbone.py:122 verbatim -- rects[0] for the whole batch, scaled by the padded map
        # (the reference only ever runs batch 1, where the two coincide; the golden vector `b2_pad` pins this mode).
        self.exemplar_mode = "per_image"

    def features(self, images, mask):
        """images [B,3,H,W], mask bool [B,H,W] -> (layer4 features NHWC [B,h,w,2048], ops.MaskInfo of the down-sampled mask)."""
        x = self.body.forward_nhwc(images)
        return x, ops.mask_prep(mask, x.shape[1], x.shape[2])

    def extract_feature(self, images, mask, rects):
        """images [B,3,H,W], mask bool [B,H,W], rects [B,K,4] normalised xyxy (device; rows with x2 < 0 = absent exemplar) ->
        (features NHWC [B,h,w,4096] = cat([x, x * exemplar feature]), mask [B,h,w])   (A2/models/backbone.py:116-145).
        The model itself never builds this tensor: AnchorDETR folds the product into its projection (ops.AggrProjFn)."""
        x, mi = self.features(images, mask)
        pf = ops.ExemplarFeatureFn.apply(x, rects, mi.extent, self.exemplar_mode == "per_image")
        return torch.cat([x, x * pf[:, None, None, :]], dim=-1), mi.m


def build_backbone(args):
    train_backbone = args.lr_backbone > 0
    assert not (args.masks or args.num_feature_levels > 1), "only the single-level 2nd-stage path is built"
    assert args.backbone == "resnet50"
    bb = BackboneAgg(train_backbone, args.dilation)
    bb.exemplar_mode = getattr(args, "exemplar_mode", "per_image")
    return bb
