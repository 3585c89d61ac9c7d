"""CPU restatement of one 2nd-stage training step (TEST INFRA; also the timed `cpu_baseline` of bench.py).

step = forward + criterion/matcher + backward + clip_grad_norm_(0.1) + AdamW, as A2/engine.py:24-57 with the
optimizer of A2/main.py:157-189 (lr 1e-4, params whose name contains "backbone" 1e-5, wd 1e-4).
"""
import torch

from . import criterion as OC
from . import model as OM
from .weights import model_schema, seeded_state_dict

FROZEN_PREFIXES = ("backbone.body.conv1", "backbone.body.layer1")   # A2/models/backbone.py:93-95
BUFFER_SUFFIXES = (".running_mean", ".running_var")


def is_frozen_bn(name):
    return name.startswith("backbone.") and (".bn" in name or ".downsample.1." in name)


def trainable_names(sd):
    """Names of tensors that are nn.Parameters with requires_grad in the reference model."""
    out = []
    for n in sd:
        if is_frozen_bn(n) or n.startswith(FROZEN_PREFIXES):
            continue
        if any(f"transformer.{fam}.{i}." in n for fam in ("cls_embed", "bbox_embed", "bbox_variance") for i in range(1, 6)):
            continue                                            # aliases of head copy 0
        out.append(n)
    return out


class OracleTrainer:
    def __init__(self, num_position=300, spatial_prior="learned", num_pattern=1, dtype=torch.float32):
        self.kw = dict(spatial_prior=spatial_prior, num_position=num_position, num_pattern=num_pattern)
        self.sd = {k: v.to(dtype) for k, v in seeded_state_dict(
            model_schema(num_position=num_position, spatial_prior=spatial_prior, num_pattern=num_pattern)).items()}
        self.names = trainable_names(self.sd)
        for n in self.names:
            self.sd[n].requires_grad_(True)
        self._alias()
        main = [self.sd[n] for n in self.names if "backbone" not in n]
        bb = [self.sd[n] for n in self.names if "backbone" in n]
        self.opt = torch.optim.AdamW([{"params": main, "lr": 1e-4}, {"params": bb, "lr": 1e-5}], lr=1e-4,
                                     weight_decay=1e-4)

    def _alias(self):
        for n in list(self.sd):
            for fam in ("cls_embed", "bbox_embed", "bbox_variance"):
                if f"transformer.{fam}.0." in n:
                    for i in range(1, 6):
                        self.sd[n.replace(f"{fam}.0.", f"{fam}.{i}.")] = self.sd[n]

    def forward_loss(self, images, rects, targets):
        out, ref = OM.forward(images, rects, self.sd, **self.kw)
        losses, idx = OC.set_criterion(out, targets)
        return out, losses, idx

    def step(self, images, rects, targets, max_norm=0.1):
        out, losses, idx = self.forward_loss(images, rects, targets)
        total = OC.total_loss(losses)
        self.opt.zero_grad()
        total.backward()
        params = [self.sd[n] for n in self.names]
        gn = torch.nn.utils.clip_grad_norm_(params, max_norm)
        self.opt.step()
        return out, losses, idx, gn


def synthetic_batch(B=2, H=800, W=800, Ts=(37, 120), seed=0):
    """SURVEY.md section 8(d) synthetic inputs (identical on CPU and GPU runs)."""
    g0 = torch.Generator().manual_seed(seed)
    images = torch.randn(B, 3, H, W, generator=g0)
    rects = torch.tensor([[.10, .10, .20, .20], [.40, .40, .50, .55], [.70, .20, .80, .30]])[None].repeat(B, 1, 1)
    g1 = torch.Generator().manual_seed(seed + 1)
    targets = []
    for b in range(B):
        T = Ts[b % len(Ts)]
        cxcy = torch.rand(T, 2, generator=g1) * 0.8 + 0.1
        wh = torch.rand(T, 2, generator=g1) * 0.10 + 0.02
        targets.append({"boxes": torch.cat([cxcy, wh], 1), "labels": torch.zeros(T, dtype=torch.int64)})
    return images, rects, targets


def synthetic_images(sizes, Ts, seed):
    """A batch of images of DIFFERENT sizes (list of [3,h,w]; the model pads them and builds the mask) with per-image exemplar
    rectangles and targets -- seeded like synthetic_batch, so golden inputs never have to be stored."""
    g0 = torch.Generator().manual_seed(seed)
    images = [torch.randn(3, h, w, generator=g0) for h, w in sizes]
    base = torch.tensor([[.10, .10, .20, .20], [.40, .40, .50, .55], [.70, .20, .80, .30]])
    rects = torch.stack([(base + 0.07 * b).clamp(max=0.95) for b in range(len(sizes))])
    g1 = torch.Generator().manual_seed(seed + 1)
    targets = []
    for b in range(len(sizes)):
        T = Ts[b % len(Ts)]
        cxcy = torch.rand(T, 2, generator=g1) * 0.8 + 0.1
        wh = torch.rand(T, 2, generator=g1) * 0.10 + 0.02
        targets.append({"boxes": torch.cat([cxcy, wh], 1), "labels": torch.zeros(T, dtype=torch.int64)})
    return images, rects, targets


def stage1_inputs(H, W, npts, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(1, 3, H, W, generator=g)
    pts = torch.rand(1, npts, 2, generator=g) * 0.8 + 0.1
    whs = torch.rand(1, npts, 2, generator=g) * 0.1 + 0.02
    return img, pts, whs
