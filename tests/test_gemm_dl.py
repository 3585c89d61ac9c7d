"""Parity of the direct-to-LDS tile kernel (csrc/igemm_dl.hip, entry cdetr_gemm_dl / picked by cdetr_gemm for pre-split operands)
against fp64 math on the SAME split operands -- conv + FrozenBN fold + bias + residual + ReLU (A2/models/resnet.py:140-160,
backbone.py:50-60) and the data-gradient form with its ReLU gate.  Every tile (128x128, 128x64, 64x128, 64x64) x ring depth x
arithmetic (split-bf16 x3 from hi|lo planes, plain bf16 from the hi plane), ragged M / N, 1x1 / 3x3 / strided / dilated rows,
padding rows from the zero page.  Needs an MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"

CONFIGS = [(0, 2), (0, 3), (1, 2), (1, 3), (1, 4), (2, 2), (2, 3), (2, 4), (3, 2), (3, 3), (3, 4)]


def g(seed):
    return torch.Generator().manual_seed(seed)


def _mirror(w, scale):
    from counting_detr_amd import ops
    mir = ops.WeightMirror([], [(w, scale)])
    mir.refresh("fwd")
    return mir, mir.lookup_fwd(w, scale)


def _split_ref(x):
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    return hi, lo


def _check_planes(y, y16, y16lo):
    assert torch.equal(y16, y.bfloat16()), "hi plane != bf16(C)"
    assert torch.equal(y16lo, (y - y16.float()).bfloat16()), "lo plane != bf16(C - hi)"


@pytest.mark.parametrize("tile,stages", CONFIGS)
@pytest.mark.parametrize("precision", [1, 3], ids=["bf16x3", "bf16"])
def test_dl_dense_rows(tile, stages, precision):
    """Dense rows (1x1 conv / linear): ragged M and N (tails clamp to valid rows / the zero page), K = 1..5 k-tiles."""
    from counting_detr_amd import ops
    for (M, N, K, epi) in [(300, 132, 128, True), (1000, 256, 256, False), (129, 64, 64, True), (5000, 512, 320, True), (64, 4, 64, False)]:
        if precision == 3 and K % 64:
            continue
        x = torch.randn(M, K, generator=g(M + K)).to(DEV)
        w = (torch.randn(N, K, generator=g(N + K)) / K ** 0.5).to(DEV)
        sc = (1 + 0.2 * torch.randn(N, generator=g(7))).to(DEV)
        bias = torch.randn(N, generator=g(8)).to(DEV) if epi else None
        resid = torch.randn(M, N, generator=g(9)).to(DEV) if epi else None
        gate = torch.randn(M, N, generator=g(10)).to(DEV) if epi else None
        mir, sp = _mirror(w, sc)
        xh, xl = ops.split_planes(x)
        y = torch.empty(M, N, device=DEV)
        y16 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        y16lo = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm_raw(x, K, w, K, y, N, M, N, K, w_scale=sc, bias=bias, relu=epi, resid=resid, ldr=N, gate=gate, ldg=N, B_split=sp,
                     precision=precision, A16=xh, A16lo=xl, C16=y16, C16lo=y16lo, dl=(tile, stages), out_scale=0.5 if epi else 1.0)
        ws = (w * sc[:, None])
        wh, wl = _split_ref(ws)
        if precision == 1:
            ref = (xh.double() + xl.double()) @ (wh.double() + wl.double()).t() - xl.double() @ wl.double().t()     # the lo*lo term is dropped
        else:
            ref = xh.double() @ wh.double().t()
        if epi:
            ref = (ref + bias.double()) * 0.5 + resid.double()
            ref = torch.where(gate.double() > 0, ref, torch.zeros_like(ref)).clamp(min=0)
        err = (y.double() - ref).abs().max().item()
        assert err <= 3e-5 * (ref.abs().max().item() + 1.0), f"M={M} N={N} K={K}: {err:.3e}"
        _check_planes(y, y16, y16lo)


@pytest.mark.parametrize("tile", [0, 1, 2, 3])
@pytest.mark.parametrize("ring", [2, 3])
@pytest.mark.parametrize("slices", [2, 3, 4])
@pytest.mark.parametrize("precision", [1, 3], ids=["bf16x3", "bf16"])
def test_dl_split_reduction(tile, ring, slices, precision):
    """The reduction of an output tile cut into 2..4 slices (stages code 200 + 10 * slices + ring depth): dense rows with ragged M / N and a
    3x3 convolution (slices that start inside a tap and cross tap boundaries), full epilogue.  The result must not depend on which slice
    arrives last: two launches are bit-identical, and equal to fp64 math on the split operands within the accumulation tolerance; the
    unsplit kernels that share the arrival counters still run right afterwards (the counters are left zero)."""
    from counting_detr_amd import ops
    code = 200 + 10 * slices + ring
    for (M, N, K) in [(1000, 256, 512), (333, 132, 256), (5000, 256, 1024)]:
        x = torch.randn(M, K, generator=g(M + K)).to(DEV)
        w = (torch.randn(N, K, generator=g(N + K)) / K ** 0.5).to(DEV)
        sc = (1 + 0.2 * torch.randn(N, generator=g(7))).to(DEV)
        bias = torch.randn(N, generator=g(8)).to(DEV)
        resid = torch.randn(M, N, generator=g(9)).to(DEV)
        gate = torch.randn(M, N, generator=g(10)).to(DEV)
        mir, sp = _mirror(w, sc)
        xh, xl = ops.split_planes(x)
        outs = []
        for rep in range(2):
            y = torch.empty(M, N, device=DEV)
            y16 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            y16lo = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            ops.gemm_raw(x, K, w, K, y, N, M, N, K, w_scale=sc, bias=bias, relu=True, resid=resid, ldr=N, gate=gate, ldg=N, B_split=sp,
                         precision=precision, A16=xh, A16lo=xl, C16=y16, C16lo=y16lo, dl=(tile, code), out_scale=0.5,
                         gate16=gate.bfloat16() if rep else None)
            _check_planes(y, y16, y16lo)
            outs.append(y)
        assert torch.equal(outs[0], outs[1]), "the split sum depends on the arrival order"
        ws = (w * sc[:, None])
        wh, wl = _split_ref(ws)
        if precision == 1:
            ref = (xh.double() + xl.double()) @ (wh.double() + wl.double()).t() - xl.double() @ wl.double().t()
        else:
            ref = xh.double() @ wh.double().t()
        ref = (ref + bias.double()) * 0.5 + resid.double()
        ref = torch.where(gate.double() > 0, ref, torch.zeros_like(ref)).clamp(min=0)
        err = (outs[0].double() - ref).abs().max().item()
        assert err <= 3e-5 * (ref.abs().max().item() + 1.0), f"M={M} N={N} K={K}: {err:.3e}"
        # whatever cdetr_gemm picks for this problem (the register-staged kernels' own split reductions share the arrival counters) still
        # runs right on the same scratch afterwards
        y2 = torch.empty(M, N, device=DEV)
        ops.gemm_raw(x, K, w, K, y2, N, M, N, K, w_scale=sc, bias=bias, relu=True, resid=resid, ldr=N, gate=gate, ldg=N, B_split=sp, precision=1, out_scale=0.5)
        ref1 = (xh.double() + xl.double()) @ (wh.double() + wl.double()).t() - xl.double() @ wl.double().t()
        ref1 = (ref1 + bias.double()) * 0.5 + resid.double()
        ref1 = torch.where(gate.double() > 0, ref1, torch.zeros_like(ref1)).clamp(min=0)
        assert (y2.double() - ref1).abs().max().item() <= 3e-5 * (ref1.abs().max().item() + 1.0)
    # 3x3, dilation 2: 9 taps x (Cin / KT) k-tiles
    Nb, H, W, Cin, Cout = 2, 19, 23, 128, 132
    x = torch.randn(Nb, H, W, Cin, generator=g(1)).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g(2)) / (Cin * 9) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    sc = (1 + 0.2 * torch.randn(Cout, generator=g(3))).to(DEV)
    bias = torch.randn(Cout, generator=g(4)).to(DEV)
    geo, Ho, Wo = ops.conv_geom_fwd(H, W, 3, 3, 1, 2, 2)
    resid = torch.randn(Nb, Ho, Wo, Cout, generator=g(5)).to(DEV)
    mir, sp = _mirror(w, sc)
    xh, xl = ops.split_planes(x)
    y = torch.empty(Nb, Ho, Wo, Cout, device=DEV)
    y16 = torch.empty(Nb, Ho, Wo, Cout, device=DEV, dtype=torch.bfloat16)
    y16lo = torch.empty_like(y16)
    ops.gemm_raw(x, Cin, w, 9 * Cin, y, Cout, Nb * Ho * Wo, Cout, Cin, taps=9, w_scale=sc, bias=bias, relu=True, resid=resid,
                 ldr=Cout, geom=geo, B_split=sp, precision=precision, A16=xh, A16lo=xl, C16=y16, C16lo=y16lo, dl=(tile, code))
    wh, wl = _split_ref(w * sc.view(-1, 1, 1, 1))
    conv = lambda a, b: F.conv2d(a.double().permute(0, 3, 1, 2).cpu(), b.double().cpu(), stride=1, padding=2, dilation=2).permute(0, 2, 3, 1)   # noqa: E731
    ref = conv(xh.double() + xl.double(), wh.double() + wl.double()) - conv(xl, wl) if precision == 1 else conv(xh, wh)
    ref = (ref + bias.double().cpu() + resid.double().cpu()).clamp(min=0)
    err = (y.double().cpu() - ref).abs().max().item()
    assert err <= 3e-5 * (ref.abs().max().item() + 1.0), f"3x3: {err:.3e}"
    _check_planes(y, y16, y16lo)


@pytest.mark.parametrize("tile,stages", [(0, 3), (1, 3), (2, 2), (3, 4)])
@pytest.mark.parametrize("precision", [1, 3], ids=["bf16x3", "bf16"])
@pytest.mark.parametrize("geom", [(3, 1, 1, 1), (3, 2, 1, 1), (3, 1, 2, 2), (1, 2, 0, 1)], ids=["3x3", "3x3s2", "3x3d2", "1x1s2"])
def test_dl_conv_forward(tile, stages, precision, geom):
    """Convolution rows: k-tiles never straddle a tap, padding taps read the zero page, strided / dilated gathers."""
    from counting_detr_amd import ops
    kh, stride, pad, dil = geom
    Nb, H, W, Cin, Cout = 2, 19, 23, 64, 132
    x = torch.randn(Nb, H, W, Cin, generator=g(1)).to(DEV)
    w = (torch.randn(Cout, Cin, kh, kh, generator=g(2)) / (Cin * kh * kh) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    sc = (1 + 0.2 * torch.randn(Cout, generator=g(3))).to(DEV)
    bias = torch.randn(Cout, generator=g(4)).to(DEV)
    geo, Ho, Wo = ops.conv_geom_fwd(H, W, kh, kh, stride, pad, dil)
    resid = torch.randn(Nb, Ho, Wo, Cout, generator=g(5)).to(DEV)
    mir, sp = _mirror(w, sc)
    xh, xl = ops.split_planes(x)
    y = torch.empty(Nb, Ho, Wo, Cout, device=DEV)
    y16 = torch.empty(Nb, Ho, Wo, Cout, device=DEV, dtype=torch.bfloat16)
    y16lo = torch.empty_like(y16)
    ops.gemm_raw(x, Cin, w, kh * kh * Cin, y, Cout, Nb * Ho * Wo, Cout, Cin, taps=kh * kh, w_scale=sc, bias=bias, relu=True, resid=resid,
                 ldr=Cout, geom=geo, B_split=sp, precision=precision, A16=xh, A16lo=xl, C16=y16, C16lo=y16lo, dl=(tile, stages))
    ws = w * sc.view(-1, 1, 1, 1)
    wh, wl = _split_ref(ws)
    conv = lambda a, b: F.conv2d(a.double().permute(0, 3, 1, 2).cpu(), b.double().cpu(), stride=stride, padding=pad, dilation=dil).permute(0, 2, 3, 1)   # noqa: E731
    if precision == 1:
        ref = conv(xh.double() + xl.double(), wh.double() + wl.double()) - conv(xl, wl)
    else:
        ref = conv(xh, wh)
    ref = (ref + bias.double().cpu() + resid.double().cpu()).clamp(min=0)
    err = (y.double().cpu() - ref).abs().max().item()
    assert err <= 3e-5 * (ref.abs().max().item() + 1.0), f"{err:.3e}"
    _check_planes(y, y16, y16lo)


@pytest.mark.parametrize("precision", [1, 3], ids=["bf16x3", "bf16"])
def test_dl_conv_dgrad_rows(precision):
    """The data-gradient row mode (transposed gather, stride 2) through the same kernel, with the ReLU gate of the layer below."""
    from counting_detr_amd import ops, _ffi
    Nb, Hin, Win, Cin, Cout, kh, stride, pad, dil = 2, 20, 22, 64, 128, 3, 2, 1, 1
    Ho, Wo = (Hin + 2 * pad - dil * (kh - 1) - 1) // stride + 1, (Win + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    dz = torch.randn(Nb, Ho, Wo, Cout, generator=g(1)).to(DEV)
    w = (torch.randn(Cout, Cin, kh, kh, generator=g(2)) / (Cout * 9) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    sc = (1 + 0.2 * torch.randn(Cout, generator=g(3))).to(DEV)
    gate = torch.randn(Nb, Hin, Win, Cin, generator=g(4)).to(DEV)
    mir = ops.WeightMirror([(w, sc)], [])
    mir.refresh("bwd")
    m = mir.lookup(w, sc)
    ws_ = (w * sc.view(-1, 1, 1, 1)).permute(1, 2, 3, 0).reshape(Cin, kh * kh * Cout)          # Wt [c][tap][o]
    assert torch.equal(m[3][:ws_.numel()].view(Cin, -1), ws_.bfloat16())                        # the plain-bf16 image of cdetr_weight_mirror
    geo = ops._geom(_ffi.ROWS_CONV_DGRAD, Ho, Wo, Hin, Win, kh, kh, stride, pad, dil)
    dh, dl_ = ops.split_planes(dz)
    for tile, stages in [(1, 3), (3, 3), (0, 3)]:
        dx = torch.empty(Nb, Hin, Win, Cin, device=DEV)
        dx16 = torch.empty(Nb, Hin, Win, Cin, device=DEV, dtype=torch.bfloat16)
        ops.gemm_raw(dz, Cout, m[0], m[1], dx, Cin, Nb * Hin * Win, Cin, Cout, taps=kh * kh, gate=gate, ldg=Cin, geom=geo, B_split=m[2],
                     B16=m[3] if tile != 3 else None,           # plain bf16: the hi-only weight image, or (tile 3) the hi halves of the split image
                     gate16=gate.bfloat16() if tile != 0 else None,      # the gate's bf16 twin (same signs) or the fp32 gate itself
                     precision=precision, A16=dh, A16lo=dl_, C16=dx16, dl=(tile, stages))
        ws = (w * sc.view(-1, 1, 1, 1))
        wh, wl = _split_ref(ws)
        ct = lambda a, b: F.conv_transpose2d(a.double().permute(0, 3, 1, 2).cpu(), b.double().cpu(), stride=stride, padding=pad, dilation=dil,   # noqa: E731
                                             output_padding=(Hin - ((Ho - 1) * stride - 2 * pad + dil * (kh - 1) + 1),
                                                             Win - ((Wo - 1) * stride - 2 * pad + dil * (kh - 1) + 1))).permute(0, 2, 3, 1)
        ref = (ct(dh.double() + dl_.double(), wh.double() + wl.double()) - ct(dl_, wl)) if precision == 1 else ct(dh, wh)
        ref = torch.where(gate.double().cpu() > 0, ref, torch.zeros_like(ref))
        err = (dx.double().cpu() - ref).abs().max().item()
        assert err <= 3e-5 * (ref.abs().max().item() + 1.0), f"tile {tile}: {err:.3e}"
        assert torch.equal(dx16, dx.bfloat16())


@pytest.mark.parametrize("tile", [0, 1, 2, 3], ids=["128x128", "128x64", "64x128", "64x64"])
@pytest.mark.parametrize("precision", [1, 3], ids=["bf16x3", "bf16"])
@pytest.mark.parametrize("case", [(2, 19, 23, 192, 132, 1), (2, 19, 23, 64, 64, 2), (1, 50, 50, 128, 128, 2), (3, 9, 100, 128, 64, 1), (2, 7, 5, 64, 68, 1)],
                         ids=["19x23", "19x23d2", "50x50d2", "9x100", "7x5"])
def test_dl_halo_conv_forward(tile, precision, case):
    """The halo-resident 3x3 form (stages code 13; math: A2/models/resnet.py:146-148 conv2 + FrozenBN + ReLU, dilation as A2/models/backbone.py:153-155):
    the pixel rows around a tile are staged once per channel chunk, the nine taps read them at row offsets.  Borders in both directions, tiles
    that straddle image rows and images, M not a multiple of the tile, 1..6 channel chunks (halo double buffer reused), halo of one or two
    pieces per batch (W = 100, dilation 2), maps smaller than a tile."""
    from counting_detr_amd import ops
    Nb, H, W, Cin, Cout, dil = case
    x = torch.randn(Nb, H, W, Cin, generator=g(1)).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g(2)) / (Cin * 9) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    sc = (1 + 0.2 * torch.randn(Cout, generator=g(3))).to(DEV)
    bias = torch.randn(Cout, generator=g(4)).to(DEV)
    geo, Ho, Wo = ops.conv_geom_fwd(H, W, 3, 3, 1, dil, dil)
    assert (Ho, Wo) == (H, W)
    resid = torch.randn(Nb, H, W, Cout, generator=g(5)).to(DEV)
    mir, sp = _mirror(w, sc)
    xh, xl = ops.split_planes(x)
    y = torch.empty(Nb, H, W, Cout, device=DEV)
    y16 = torch.empty(Nb, H, W, Cout, device=DEV, dtype=torch.bfloat16)
    y16lo = torch.empty_like(y16)
    ops.gemm_raw(x, Cin, w, 9 * Cin, y, Cout, Nb * H * W, Cout, Cin, taps=9, w_scale=sc, bias=bias, relu=True, resid=resid,
                 ldr=Cout, geom=geo, B_split=sp, precision=precision, A16=xh, A16lo=xl, C16=y16, C16lo=y16lo, dl=(tile, 13))
    ws = w * sc.view(-1, 1, 1, 1)
    wh, wl = _split_ref(ws)
    conv = lambda a, b: F.conv2d(a.double().permute(0, 3, 1, 2).cpu(), b.double().cpu(), padding=dil, dilation=dil).permute(0, 2, 3, 1)   # noqa: E731
    ref = (conv(xh.double() + xl.double(), wh.double() + wl.double()) - conv(xl, wl)) if precision == 1 else conv(xh, wh)
    ref = (ref + bias.double().cpu() + resid.double().cpu()).clamp(min=0)
    err = (y.double().cpu() - ref).abs().max().item()
    assert err <= 3e-5 * (ref.abs().max().item() + 1.0), f"{err:.3e}"
    _check_planes(y, y16, y16lo)
    # and against the classic (tap-by-tap gather) form of the same kernel: same products, another summation order
    y2 = torch.empty_like(y)
    ops.gemm_raw(x, Cin, w, 9 * Cin, y2, Cout, Nb * H * W, Cout, Cin, taps=9, w_scale=sc, bias=bias, relu=True, resid=resid,
                 ldr=Cout, geom=geo, B_split=sp, precision=precision, A16=xh, A16lo=xl, C16=y16, C16lo=y16lo, dl=(3, 3))
    assert (y - y2).abs().max().item() <= 2e-5 * (ref.abs().max().item() + 1.0)


@pytest.mark.parametrize("tile", [0, 1, 2, 3], ids=["128x128", "128x64", "64x128", "64x64"])
@pytest.mark.parametrize("precision", [1, 3], ids=["bf16x3", "bf16"])
@pytest.mark.parametrize("dil", [1, 2])
def test_dl_halo_conv_dgrad_rows(tile, precision, dil):
    """Data-gradient rows of a stride-1 3x3 through the halo form (the taps mirror: pixel - (k - 1) dil), ReLU gate from the bf16 twin,
    twin-only output (C == NULL) as the backbone's inner gradients use it."""
    from counting_detr_amd import ops, _ffi
    Nb, H, W, Cin, Cout = 2, 21, 17, 128, 192
    dz = torch.randn(Nb, H, W, Cout, generator=g(1)).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g(2)) / (Cout * 9) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    sc = (1 + 0.2 * torch.randn(Cout, generator=g(3))).to(DEV)
    gate = torch.randn(Nb, H, W, Cin, generator=g(4)).to(DEV)
    mir = ops.WeightMirror([(w, sc)], [])
    mir.refresh("bwd")
    m = mir.lookup(w, sc)
    geo = ops._geom(_ffi.ROWS_CONV_DGRAD, H, W, H, W, 3, 3, 1, dil, dil)
    dh, dl_ = ops.split_planes(dz)
    dx = torch.empty(Nb, H, W, Cin, device=DEV)
    dx16 = torch.empty(Nb, H, W, Cin, device=DEV, dtype=torch.bfloat16)
    ops.gemm_raw(dz, Cout, m[0], m[1], dx, Cin, Nb * H * W, Cin, Cout, taps=9, gate=gate, ldg=Cin, geom=geo, B_split=m[2], B16=m[3],
                 gate16=gate.bfloat16(), precision=precision, A16=dh, A16lo=dl_, C16=dx16, dl=(tile, 13))
    ws = (w * sc.view(-1, 1, 1, 1))
    wh, wl = _split_ref(ws)
    ct = lambda a, b: F.conv_transpose2d(a.double().permute(0, 3, 1, 2).cpu(), b.double().cpu(), padding=dil, dilation=dil).permute(0, 2, 3, 1)   # noqa: E731
    ref = (ct(dh.double() + dl_.double(), wh.double() + wl.double()) - ct(dl_, wl)) if precision == 1 else ct(dh, wh)
    ref = torch.where(gate.double().cpu() > 0, ref, torch.zeros_like(ref))
    err = (dx.double().cpu() - ref).abs().max().item()
    assert err <= 3e-5 * (ref.abs().max().item() + 1.0), f"{err:.3e}"
    assert torch.equal(dx16, dx.bfloat16())
    if precision == 3:
        dx16b = torch.empty_like(dx16)
        ops.gemm_raw(None, Cout, m[0], m[1], None, Cin, Nb * H * W, Cin, Cout, taps=9, gate=gate, ldg=Cin, geom=geo, B_split=m[2], B16=m[3],
                     gate16=gate.bfloat16(), precision=3, A16=dh, C16=dx16b, dl=(tile, 13))
        assert torch.equal(dx16b, dx16)


def test_grouped_operands_through_conv_ops():
    """ops.Groups (interleaved split-bf16 groups, include/cdetr_hip.h CDETR_GEMM_A_GROUPS / C_GROUPS / RESID_GROUPS) through ops.conv_fwd:
    a 1x1 convolution reading grouped rows, adding a grouped residual and writing grouped output equals the fp32-tensor path on the same
    hi + lo values (A2/models/resnet.py:140-160: conv + FrozenBN + residual + ReLU)."""
    from counting_detr_amd import ops
    ops.PRECISION, old = 1, ops.PRECISION
    Nb, H, W, Cin, Cout = 2, 48, 50, 256, 128
    try:
        x = torch.randn(Nb, H, W, Cin, generator=g(1)).to(DEV)
        r = torch.randn(Nb, H, W, Cout, generator=g(2)).to(DEV)
        w = (torch.randn(Cout, Cin, 1, 1, generator=g(3)) / Cin ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
        sc = (1 + 0.2 * torch.randn(Cout, generator=g(4))).to(DEV)
        bias = torch.randn(Cout, generator=g(5)).to(DEV)
        ops.MIRROR = ops.WeightMirror([], [(w, sc)])
        ops.MIRROR.refresh("fwd")
        xg, rg = ops.Groups.of(x), ops.Groups.of(r)
        assert torch.equal(xg.float(), (x.bfloat16().float() + (x - x.bfloat16().float()).bfloat16().float()))
        yg, y16 = ops.conv_fwd(xg, w, sc, bias, relu=True, resid=rg, twin=True, out_groups=True)
        assert isinstance(yg, ops.Groups) and yg.shape == (Nb, H, W, Cout)
        # the same values as fp32 tensors through the ordinary path
        y_ref = ops.conv_fwd(xg.float(), w, sc, bias, relu=True, resid=rg.float())
        got = yg.float()
        assert (got - y_ref).abs().max().item() <= 2e-5 * (y_ref.abs().max().item() + 1.0)
        hi = y_ref.bfloat16()
        assert (y16.float() - hi.float()).abs().max().item() <= 2 ** -7 * (y_ref.abs().max().item() + 1.0)      # twin = bf16 of the same value (ulp flips at ties)
        # groups written by the epilogue ARE the split of its fp32 value: hi = bf16(v), lo = bf16(v - hi)
        gt = yg.t.view(-1, Cout // 32, 64)
        assert torch.equal(gt[..., :32].reshape(-1, Cout), y16.view(-1, Cout))
    finally:
        ops.MIRROR = None
        ops.PRECISION = old


@pytest.mark.parametrize("planes,H,W", [(64, 48, 48), (128, 50, 50)])
def test_bottleneck_chain_with_grouped_block_outputs(planes, H, W):
    """Two stride-1 bottlenecks (A2/models/resnet.py:105-160) in training mode, the first block's output handed over as ops.Groups (no fp32
    tensor: conv1 of the second block streams it on the direct-to-LDS kernel, its conv3 adds hi + lo, the backward takes ReLU mask and
    weight-gradient operand from the twin) against the same chain with an fp32 hand-over: outputs to 2e-5, input gradient and every weight
    gradient to 1e-2 in norm (a few ReLU masks flip between two forwards that differ in the fifth digit)."""
    from counting_detr_amd import ops
    from counting_detr_amd.backbone import Bottleneck
    ops.PRECISION, old = 1, ops.PRECISION
    ops.PRECISION_BWD = 3
    torch.manual_seed(0)
    try:
        blks = [Bottleneck(4 * planes, planes, 1, 1, False).to(DEV) for _ in range(2)]
        for b in blks:
            for bn in (b.bn1, b.bn2, b.bn3):
                bn.weight.data.uniform_(0.5, 1.5)
                bn.bias.data.normal_(0, 0.1)
        ent = []
        for b in blks:
            ent += [(b.conv1.weight.data, b.bn1.affine()[0]), (b.conv2.weight.data, b.bn2.affine()[0]), (b.conv3.weight.data, b.bn3.affine()[0])]
        ops.MIRROR = ops.WeightMirror(ent, ent)
        ops.MIRROR.refresh()
        x = torch.randn(2, H, W, 4 * planes, generator=g(7)).relu().to(DEV)
        x16 = x.bfloat16()
        dz = torch.randn(2, H, W, 4 * planes, generator=g(8)).to(DEV)
        res = {}
        for grouped in (False, True):
            for b in blks:
                for c in (b.conv1, b.conv2, b.conv3):
                    c.weight.grad = torch.zeros_like(c.weight)
            saved = []
            with torch.no_grad():
                h, h16 = blks[0].forward_fused(x, saved, x16=x16, twins=True, out_groups=grouped)
                assert isinstance(h, ops.Groups) == grouped
                out, out16 = blks[1].forward_fused(h, saved, x16=h16, twins=True)
                d, d16 = ops.relu_mask(out, dz, twin=True)
                d, d16 = blks[1].backward_fused(saved[1], d, need_dx=True, dz16=d16)
                dx, _ = blks[0].backward_fused(saved[0], d, need_dx=True, dz16=d16)
            torch.cuda.synchronize()
            res[grouped] = (out.clone(), dx.clone(), [c.weight.grad.clone() for b in blks for c in (b.conv1, b.conv2, b.conv3)])
        o0, dx0, g0 = res[False]
        o1, dx1, g1 = res[True]
        assert (o0 - o1).abs().max().item() <= 2e-5 * (o0.abs().max().item() + 1.0)
        # the two forwards differ by ~1e-5 of the scale, so a handful of the 10^6 outputs sit on the other side of a ReLU: their mask flips and the
        # gradient moves by a full element there (seen: max difference 5 % of the largest entry, one flip).  The norms say whether the paths agree.
        rel = lambda a, b: ((a - b).double().norm() / a.double().norm()).item()      # noqa: E731
        assert rel(dx0, dx1) <= 1e-2, rel(dx0, dx1)
        for a, b in zip(g0, g1):
            assert rel(a, b) <= 1e-2, rel(a, b)
    finally:
        ops.MIRROR = None
        ops.PRECISION = old


def test_dl_is_what_cdetr_gemm_picks_for_presplit_operands():
    """cdetr_gemm itself routes a large-enough problem with A16 + A16lo + B_split to the direct-to-LDS kernel: same result as the
    forced configuration, and within the split-product error of the register-staged kernel fed the fp32 operand."""
    from counting_detr_amd import ops
    M, N, K = 20000, 512, 128
    x = torch.randn(M, K, generator=g(1)).to(DEV)
    w = (torch.randn(N, K, generator=g(2)) / K ** 0.5).to(DEV)
    mir, sp = _mirror(w, None)
    xh, xl = ops.split_planes(x)
    y_auto, y_forced, y_old = (torch.empty(M, N, device=DEV) for _ in range(3))
    ops.gemm_raw(x, K, w, K, y_auto, N, M, N, K, B_split=sp, precision=1, A16=xh, A16lo=xl)
    ops.gemm_raw(x, K, w, K, y_forced, N, M, N, K, B_split=sp, precision=1, A16=xh, A16lo=xl, dl=(0, 3))
    ops.gemm_raw(x, K, w, K, y_old, N, M, N, K, B_split=sp, precision=1)
    assert torch.equal(y_auto, y_forced)
    np.testing.assert_allclose(y_auto.cpu().numpy(), y_old.cpu().numpy(), rtol=0, atol=2e-5 * float(y_old.abs().max()))


def test_twin_only_output_through_cdetr_gemm():
    """cdetr_gemm_desc.C == NULL (an inner gradient of a bottleneck: nothing reads its fp32 copy): the direct-to-LDS kernel writes the
    bf16 twin alone, bit-identical to the twin of the full call, whatever the problem size; without direct-to-LDS operands the call is
    refused, not run; the twin-fed weight-gradient kernel accepts dY == NULL the same way."""
    from counting_detr_amd import ops
    for (M, N, K) in [(5000, 256, 1024), (300, 64, 128)]:          # the second one is below the kernel's usual size threshold
        dz = torch.randn(M, K, generator=g(M)).to(DEV)
        w = (torch.randn(K, N, generator=g(K)) / K ** 0.5).to(DEV)          # logical [Cout = K][Cin = N]
        gate = torch.randn(M, N, generator=g(3)).to(DEV)
        mir = ops.WeightMirror([(w, None)], [])
        mir.refresh("bwd")
        m = mir.lookup(w, None)
        dz16 = dz.bfloat16()
        full, full16, only16 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV, dtype=torch.bfloat16), torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        kw = dict(gate=gate, ldg=N, B_split=m[2], B16=m[3], precision=3, A16=dz16, gate16=gate.bfloat16())
        ops.gemm_raw(dz, K, m[0], m[1], full, N, M, N, K, C16=full16, dl=(3, 3), **kw)
        ops.gemm_raw(None, K, m[0], m[1], None, N, M, N, K, C16=only16, **kw)          # neither the fp32 operand nor the fp32 result exists
        assert torch.equal(only16, full16) and torch.equal(full16, full.bfloat16())
        with pytest.raises(RuntimeError):                                       # no twin of A: not a direct-to-LDS problem -> refused
            ops.gemm_raw(dz, K, m[0], m[1], None, N, M, N, K, C16=only16, gate=gate, ldg=N, B_split=m[2], precision=3)
    # weight gradient from twins only
    P, Nout, Cin = 5000, 256, 128
    dY, X = torch.randn(P, Nout, generator=g(1)).to(DEV), torch.randn(P, Cin, generator=g(2)).to(DEV)
    dW1, dW2 = torch.zeros(Nout, Cin, device=DEV), torch.zeros(Nout, Cin, device=DEV)
    ops.wgrad_raw(dY, Nout, X, Cin, dW1, Cin, P, Nout, Cin, dY16=dY.bfloat16(), X16=X.bfloat16(), precision=3)
    ops.wgrad_raw(None, Nout, X, Cin, dW2, Cin, P, Nout, Cin, dY16=dY.bfloat16(), X16=X.bfloat16(), precision=3)
    ref = dY.bfloat16().double().t() @ X.bfloat16().double()
    assert (dW2.double() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    np.testing.assert_allclose(dW1.cpu().numpy(), dW2.cpu().numpy(), rtol=0, atol=1e-5 * float(ref.abs().max()))     # (atomic accumulation order)
    with pytest.raises(RuntimeError):
        ops.wgrad_raw(None, Nout, X, Cin, dW2, Cin, P, Nout, Cin, dY16=dY.bfloat16(), precision=3)       # no X twin
