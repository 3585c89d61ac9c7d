"""Deterministic, name-seeded random initialisation for benchmarking without checkpoints (no network in this environment):
every tensor of the state dict is filled from a generator seeded by the CRC of its name, with the scales of the reference's
own initialisers (kaiming conv / xavier linear / identity-ish norms, A2/models/resnet.py:231-233, transformer.py:86-103) and
non-degenerate heads (class bias -log 99, box-size bias -2, small positive variance head), so the step exercises realistic
matching costs.  Independent of the test oracle (which has its own generator for the golden vectors)."""
import math
import zlib

import torch


def _gen(name, seed):
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


@torch.no_grad()
def seeded_init_(model, seed=0):
    done = set()
    for name, t in model.state_dict().items():
        if t.data_ptr() in done or not t.is_floating_point():
            continue
        done.add(t.data_ptr())
        g = _gen(name, seed)
        shape = tuple(t.shape)
        n = lambda: torch.randn(shape, generator=g)          # noqa: E731
        u = lambda: torch.rand(shape, generator=g)           # noqa: E731
        if name.endswith("running_var"):
            v = 1.0 + 0.2 * u()
        elif name.endswith("running_mean"):
            v = 0.1 * n()
        elif "position.weight" in name:
            v = u()
        elif "pattern.weight" in name:
            v = n()
        elif "cls_embed" in name:
            v = 0.05 * n() if t.dim() == 2 else torch.full(shape, -math.log(99.0)) + 0.1 * n()
        elif "bbox_embed" in name and ".layers.2." in name:
            v = 0.02 * n() if t.dim() == 2 else torch.tensor([0.0, 0.0, -2.0, -2.0])[: shape[0]] + 0.05 * n()
        elif "bbox_variance" in name and ".layers.2." in name:
            v = 0.01 * (1.0 + 0.2 * u()) if t.dim() == 2 else torch.full(shape, 0.01)
        elif t.dim() == 4:
            co, ci, kh, kw = shape
            v = n() * (math.sqrt(2.0 / ((ci + co) * kh * kw)) if "input_proj" in name else 0.9 * math.sqrt(2.0 / (co * kh * kw)))
        elif t.dim() == 2:
            v = n() * math.sqrt(2.0 / (shape[0] + shape[1]))
        elif name.endswith("weight"):                        # norm scales (BN / LayerNorm / GroupNorm)
            v = 1.0 + (0.1 if ("bn" in name or "downsample" in name) else 0.05) * n()
        else:                                                # biases
            v = (0.1 if ("bn" in name or "downsample" in name) else 0.02) * n()
        t.copy_(v.reshape(shape).to(t.device))
    return model
