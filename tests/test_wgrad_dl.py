"""Parity of the direct-to-LDS weight-gradient kernel (csrc/wgrad_dl.hip; entry cdetr_wgrad_dl, picked by cdetr_wgrad /
cdetr_wgrad_group for plain-bf16 problems with twins) against fp64 math on the SAME bf16 twins: dW[i][tap][c] += s[i] * sum_p
dY16[p][i] X16[row(p, tap)][c] -- autograd of F.conv2d / F.linear w.r.t. the weight (A2/models/resnet.py:140-160,
transformer.py:242-279).  Every tile x pixels-per-tile x ring depth, dense and 3x3 / strided / dilated rows, pixel counts that are not
multiples of the tile, accumulation into a non-zero gradient, the grouped launch.  Needs an MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
CFGS = [13, 14, 23, 22, 113, 114, 123, 213, 214, 223, 313, 314, 323, 324]


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("cfg", CFGS)
def test_wgrad_dl_dense(cfg):
    from counting_detr_amd import ops
    for (P, Nout, Cin) in [(5000, 256, 128), (333, 64, 192), (20000, 128, 512), (70, 320, 96)]:
        dy = torch.randn(P, Nout, generator=g(P)).to(DEV)
        x = torch.randn(P, Cin, generator=g(P + 1)).to(DEV)
        sc = (1 + 0.2 * torch.randn(Nout, generator=g(2))).to(DEV)
        dy16, x16 = dy.bfloat16(), x.bfloat16()
        dw = torch.randn(Nout, Cin, generator=g(3)).to(DEV)
        base = dw.clone()
        for target in (0, 64):                                   # default slice plan, and few slices (long k loops; `single` when one slice)
            ops.wgrad_raw(dy, Nout, x, Cin, dw, Cin, P, Nout, Cin, w_scale=sc, dY16=dy16, X16=x16, dl=(cfg, target), precision=3)
        ref = base.double() + 2 * sc.double()[:, None] * (dy16.double().t() @ x16.double())
        err = (dw.double() - ref).abs().max().item()
        assert err <= 2e-5 * (ref.abs().max().item() + 1.0), f"cfg {cfg} P={P} {Nout}x{Cin}: {err:.3e}"


@pytest.mark.parametrize("cfg", [13, 23, 113, 213, 313, 324])
@pytest.mark.parametrize("geom", [(3, 1, 1, 1), (3, 2, 1, 1), (3, 1, 2, 2), (1, 2, 0, 1)], ids=["3x3", "3x3s2", "3x3d2", "1x1s2"])
def test_wgrad_dl_conv(cfg, geom):
    from counting_detr_amd import ops
    kh, stride, pad, dil = geom
    Nb, H, W, Cin, Cout = 2, 19, 23, 64, 128
    x = torch.randn(Nb, H, W, Cin, generator=g(1)).to(DEV)
    geo, Ho, Wo = ops.conv_geom_fwd(H, W, kh, kh, stride, pad, dil)
    dz = torch.randn(Nb, Ho, Wo, Cout, generator=g(2)).to(DEV)
    w = torch.zeros(Cout, Cin, kh, kh, device=DEV).contiguous(memory_format=torch.channels_last)
    dw = torch.zeros_like(w)
    x16, dz16 = x.bfloat16(), dz.bfloat16()
    ops.wgrad_raw(dz, Cout, x, Cin, dw, kh * kh * Cin, Nb * Ho * Wo, Cout, Cin, taps=kh * kh, geom=geo, dY16=dz16, X16=x16, dl=(cfg, 0), precision=3)
    x64 = x16.double().permute(0, 3, 1, 2).cpu().requires_grad_(False)
    w64 = torch.zeros(Cout, Cin, kh, kh, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x64, w64, stride=stride, padding=pad, dilation=dil)
    y.backward(dz16.double().permute(0, 3, 1, 2).cpu())
    err = (dw.double().cpu() - w64.grad).abs().max().item()
    assert err <= 2e-5 * (w64.grad.abs().max().item() + 1.0), f"{err:.3e}"


def test_wgrad_dl_is_the_default_for_twin_problems_and_groups():
    """cdetr_wgrad / cdetr_wgrad_group route plain-bf16 problems with twins to the direct-to-LDS kernel (single and grouped launches):
    same sums as the forced configuration; problems without twins keep the register-staged kernels."""
    from counting_detr_amd import ops
    old = ops.PRECISION_BWD
    ops.PRECISION_BWD = 3
    try:
        probs = []
        for i, (P, Nout, Cin) in enumerate([(5000, 256, 1024), (5000, 1024, 256), (20000, 128, 128), (5000, 64, 64)]):
            dy = torch.randn(P, Nout, generator=g(10 + i)).to(DEV)
            x = torch.randn(P, Cin, generator=g(20 + i)).to(DEV)
            probs.append((dy, x, dy.bfloat16(), x.bfloat16(), torch.zeros(Nout, Cin, device=DEV), torch.zeros(Nout, Cin, device=DEV)))
        with ops.wgrad_queue():
            for dy, x, dy16, x16, dw, _ in probs:
                ops.wgrad_raw(dy, dy.shape[1], x, x.shape[1], dw, x.shape[1], dy.shape[0], dy.shape[1], x.shape[1], dY16=dy16, X16=x16, may_defer=True)
        for dy, x, dy16, x16, dw, dw1 in probs:
            ops.wgrad_raw(dy, dy.shape[1], x, x.shape[1], dw1, x.shape[1], dy.shape[0], dy.shape[1], x.shape[1], dY16=dy16, X16=x16)
            ref = dy16.double().t() @ x16.double()
            for got in (dw, dw1):
                err = (got.double() - ref).abs().max().item()
                assert err <= 2e-5 * (ref.abs().max().item() + 1.0), f"{tuple(dw.shape)}: {err:.3e}"
    finally:
        ops.PRECISION_BWD = old
