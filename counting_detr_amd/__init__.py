"""counting_detr_amd -- MI355X-native (gfx950) implementation of the Counting-DETR 2nd-stage train / inference hot path.

Public surface = the reference's (A2/models/__init__.py:15-16): `build_model(args) -> (model, criterion, postprocessors)`.
Heavy lifting: hand-written HIP kernels in csrc/ behind the C-ABI of include/cdetr_hip.h (ctypes, no torch types).
There is no CPU / eager fallback: importing is cheap, but building a model without lib/libcdetr_hip.so raises.
"""
from .anchor_detr import build as _build


def build_model(args):
    return _build(args)


__all__ = ["build_model"]
