"""Parity of the HIP kernels (through the C-ABI) against the oracle / plain fp64 CPU math.  Needs an MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(params=[0, 1], ids=["fp32mfma", "bf16x3"])
def precision(request):
    """Both matrix-core arithmetic modes of the GEMM kernels: fp32 MFMA (exact products) and split-bf16 x3."""
    from counting_detr_amd import ops
    old = ops.PRECISION
    ops.PRECISION = request.param
    yield request.param
    ops.PRECISION = old


# Kernel tests that contain a backward contraction also run in the library's DEFAULT backward arithmetic (ops.PRECISION_BWD = 3: plain
# bf16 products fed by bf16 twins, 1 MFMA per product) with its own bar: a bf16 rounding of both operands is 2^-9 relative per
# product, random in sign -- a contraction's error is ~3e-3 of its typical magnitude, 1.5e-2 of the tensor's scale is the bar (the
# forward's split-bf16 / fp32 results inside those tests are held to their strict bars by the bf16x3-backward run of the same test).
BWD_DEFAULT_TESTS = {"test_linear_forward_backward", "test_conv_forward_dgrad_wgrad", "test_rcda_core", "test_multihead_rcda_module_vs_oracle",
                     "test_encoder_layer_fused_equals_unfused", "test_decoder_stack_fused_equals_unfused", "test_mha_core",
                     "test_weight_mirror_and_dgrad", "test_exemplar_feature_and_concat_free_projection", "test_wgrad_group_matches_individual_calls",
                     "test_two_level_batch_image_by_head"}
BF16_BWD_BAR = 1.5e-2
_BWD_MODE = 1


def pytest_generate_tests(metafunc):
    if "bwd_mode" in metafunc.fixturenames:
        modes = [1, 3] if metafunc.function.__name__ in BWD_DEFAULT_TESTS else [1]
        metafunc.parametrize("bwd_mode", modes, ids=["bwd-as-fwd" if m == 1 else "bwd-bf16" for m in modes])


@pytest.fixture(autouse=True)
def _backward_arithmetic(request, bwd_mode):
    """bwd-as-fwd: the backward contractions pinned to the forward's arithmetic (bf16x3 / fp32): the strict 2e-4 bars are the bars of
    those kernels.  bwd-bf16: the shipped default (plain bf16, twins), bar BF16_BWD_BAR; only with the bf16x3 forward (the fp32-MFMA
    mode ignores PRECISION_BWD)."""
    global _BWD_MODE
    from counting_detr_amd import ops
    if bwd_mode == 3 and "precision" in request.fixturenames and request.getfixturevalue("precision") != 1:
        pytest.skip("PRECISION_BWD only applies to the split-bf16 mode")
    old = ops.PRECISION_BWD
    ops.PRECISION_BWD = _BWD_MODE = bwd_mode
    yield
    ops.PRECISION_BWD = old
    _BWD_MODE = 1


def tol(precision):
    return dict(rtol=2e-4, atol_scale=2e-5) if precision == 0 else dict(rtol=2e-4, atol_scale=6e-5)


def g(seed):
    return torch.Generator().manual_seed(seed)


def close(actual, ref, rtol=2e-4, atol_scale=2e-5, msg=""):
    """fp32-MFMA tolerance: relative to the reference's magnitude (sums of K products)."""
    a = actual.detach().double().cpu()
    r = ref.detach().double().cpu()
    assert a.shape == r.shape, (a.shape, r.shape)
    scale = r.abs().max().item() + 1e-30
    err = (a - r).abs().max().item()
    if _BWD_MODE == 3:
        rtol = max(rtol, BF16_BWD_BAR)
    assert torch.isfinite(a).all(), msg + " non-finite output"
    assert err <= rtol * scale + atol_scale * scale, f"{msg} max err {err:.3e} vs scale {scale:.3e}"


# ----------------------------------------------------------------------------------------------------- GEMM / linear
@pytest.mark.parametrize("M,N,K", [(600, 256, 256), (5000, 1024, 256), (5000, 256, 1024), (77, 70, 36), (600, 2, 256),
                                   (600, 4, 256), (1, 256, 256), (130, 64, 20), (300, 512, 256)])
@pytest.mark.parametrize("relu,use_resid", [(False, False), (True, True)])
def test_linear_forward_backward(M, N, K, relu, use_resid, precision):
    from counting_detr_amd import ops
    x = torch.randn(M, K, generator=g(1))
    w = torch.randn(N, K, generator=g(2)) / K ** 0.5
    b = torch.randn(N, generator=g(3))
    r = torch.randn(M, N, generator=g(4)) if use_resid else None
    gy = torch.randn(M, N, generator=g(5))
    xd = x.to(DEV).requires_grad_(True)
    wd = torch.nn.Parameter(w.to(DEV))
    bd = torch.nn.Parameter(b.to(DEV))
    rd = r.to(DEV).requires_grad_(True) if use_resid else None
    y = ops.linear(xd, wd, bd, relu=relu, resid=rd)
    y.backward(gy.to(DEV))
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    r64 = r.double().requires_grad_(True) if use_resid else None
    y64 = F.linear(x64, w64, b64)
    if use_resid:
        y64 = y64 + r64
    if relu:   # use the kernel's own ReLU mask in the reference backward: pre-activations within rounding of 0 may flip
        y64 = y64 * (y.detach().double().cpu() > 0)
    y64.backward(gy.double())
    t = tol(precision)
    close(y, y64, msg="y", **t)
    close(xd.grad, x64.grad, msg="dx", **t)
    close(wd.grad, w64.grad, msg="dW", **t)
    close(bd.grad, b64.grad, msg="db", **t)
    if use_resid:
        close(rd.grad, r64.grad, msg="dresid", **t)


def test_linear_row_slices_accumulate():
    """in_proj style: rows [lo:hi) of one parameter; gradients accumulate in place into .grad rows."""
    from counting_detr_amd import ops
    E = 256
    w = torch.randn(5 * E, E, generator=g(1)) / 16
    b = torch.randn(5 * E, generator=g(2))
    x = torch.randn(2, 37, E, generator=g(3))
    wd, bd = torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(b.to(DEV))
    xd = x.to(DEV).requires_grad_(True)
    y = ops.linear(xd, wd, bd, rows=(2 * E, 3 * E)) + ops.linear(xd, wd, bd, rows=(2 * E, 3 * E)) * 2
    y.sum().backward()
    w64, b64, x64 = w.double().requires_grad_(True), b.double().requires_grad_(True), x.double().requires_grad_(True)
    y64 = 3 * F.linear(x64, w64[2 * E:3 * E], b64[2 * E:3 * E])
    y64.sum().backward()
    close(y, y64)
    close(wd.grad, w64.grad, msg="dW")
    close(bd.grad, b64.grad, msg="db")
    close(xd.grad, x64.grad, msg="dx")


# ----------------------------------------------------------------------------------------------------- conv
CONV_CASES = [
    # Cin, Cout, k, stride, pad, dil, H, W
    (64, 64, 1, 1, 0, 1, 20, 24),
    (64, 128, 3, 1, 1, 1, 20, 24),
    (128, 128, 3, 2, 1, 1, 21, 25),
    (128, 64, 3, 1, 2, 2, 13, 17),
    (256, 512, 1, 2, 0, 1, 20, 24),
    (4, 64, 7, 2, 3, 1, 64, 96),
    (512, 2048, 1, 1, 0, 1, 10, 12),
]


@pytest.mark.parametrize("Cin,Cout,k,stride,pad,dil,H,W", CONV_CASES)
def test_conv_forward_dgrad_wgrad(Cin, Cout, k, stride, pad, dil, H, W, precision):
    from counting_detr_amd import ops
    B = 2
    x = torch.randn(B, Cin, H, W, generator=g(1))
    w = torch.randn(Cout, Cin, k, k, generator=g(2)) / (Cin * k * k) ** 0.5
    scale = 1 + 0.1 * torch.randn(Cout, generator=g(3))
    bias = 0.1 * torch.randn(Cout, generator=g(4))
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    conv64 = F.conv2d(x64, w64, stride=stride, padding=pad, dilation=dil)
    Ho, Wo = conv64.shape[-2:]
    resid = torch.randn(B, Cout, Ho, Wo, generator=g(5))
    pre64 = conv64 * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1) + resid.double()
    gy = torch.randn(B, Cout, Ho, Wo, generator=g(6))
    dz64 = gy.double() * (pre64 > 0)                     # gradient w.r.t. the pre-ReLU value
    y64 = F.relu(pre64)
    pre64.backward(dz64)

    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = torch.nn.Parameter(w.contiguous(memory_format=torch.channels_last).to(DEV))
    assert wd.stride() == w.contiguous(memory_format=torch.channels_last).stride()
    sd, bd = scale.to(DEV), bias.to(DEV)
    rd = resid.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = ops.conv_fwd(xd, wd, sd, bd, stride=stride, pad=pad, dil=dil, relu=True, resid=rd)
    t = tol(precision)
    close(y.permute(0, 3, 1, 2), y64, msg="conv fwd", **t)
    dz = dz64.float().permute(0, 2, 3, 1).contiguous().to(DEV)
    dx = ops.conv_dgrad(dz, wd, sd, (H, W), stride=stride, pad=pad, dil=dil)
    close(dx.permute(0, 3, 1, 2), x64.grad, msg="conv dgrad", **t)
    # gate + resid epilogue of the data-gradient
    gate = torch.randn(B, H, W, Cin, generator=g(7)).to(DEV)
    extra = torch.randn(B, H, W, Cin, generator=g(8)).to(DEV)
    dx2 = ops.conv_dgrad(dz, wd, sd, (H, W), stride=stride, pad=pad, dil=dil, gate=gate, resid=extra)
    ref2 = (x64.grad.permute(0, 2, 3, 1) + extra.double().cpu()) * (gate.cpu() > 0)
    close(dx2, ref2, msg="conv dgrad gate+resid", **t)
    ops.conv_wgrad_(dz, xd, wd, sd, stride=stride, pad=pad, dil=dil)
    ops.conv_wgrad_(dz, xd, wd, sd, stride=stride, pad=pad, dil=dil)    # accumulates
    close(wd.grad, 2 * w64.grad, msg="conv wgrad", rtol=5e-4)


def test_maxpool():
    from counting_detr_amd import ops
    x = torch.randn(2, 64, 37, 41, generator=g(1))
    y = ops.maxpool3x3s2(x.permute(0, 2, 3, 1).contiguous().to(DEV))
    ref = F.max_pool2d(x, 3, 2, 1)
    assert torch.equal(y.permute(0, 3, 1, 2).cpu(), ref)


# ----------------------------------------------------------------------------------------------------- RCDA core
def rcda_core_ref(qr, qc, kr, kc, v, mr, mc, nh):
    """fp64 restatement of A2/models/row_column_decoupled_attention.py:215-309 on projected inputs."""
    N, L, E = qr.shape
    H, W = v.shape[1:3]
    d = E // nh
    hs = lambda t: t.reshape(N, -1, nh, d).permute(0, 2, 1, 3)   # noqa: E731
    s_row = (hs(qr) * d ** -0.5) @ hs(kr).transpose(-1, -2)
    s_col = (hs(qc) * d ** -0.5) @ hs(kc).transpose(-1, -2)
    if mr is not None:
        s_row = s_row.masked_fill(mr.bool()[:, None, None, :], float("-inf"))
        s_col = s_col.masked_fill(mc.bool()[:, None, None, :], float("-inf"))
    a_row, a_col = s_row.softmax(-1), s_col.softmax(-1)
    vv = v.reshape(N, H, W, nh, d).permute(0, 3, 1, 2, 4)
    o = torch.einsum("bnqh,bnqw,bnhwc->bnqc", a_col, a_row, vv)
    return o.permute(0, 2, 1, 3).reshape(N, L, E)


@pytest.mark.parametrize("N,L,H,W,masked", [(2, 300, 50, 50, False), (1, 600, 20, 30, True), (2, 77, 7, 5, True),
                                            (1, 130, 70, 40, True), (2, 2500, 50, 50, True), (1, 33, 24, 36, False),
                                            (1, 200, 40, 84, True), (1, 100, 84, 70, False), (1, 64, 16, 64, True),   # W > 64: the wide-map kernels
                                            (2, 4200, 50, 84, True), (2, 300, 50, 84, True),      # FSCD-LVIS 800 x 1333: encoder / decoder shapes (round 6)
                                            (1, 150, 33, 65, True), (1, 96, 96, 96, True), (1, 70, 84, 50, True), (1, 60, 100, 20, False)])
def test_rcda_core(N, L, H, W, masked, precision):
    from counting_detr_amd import ops
    nh, E = 8, 256
    qr, qc = torch.randn(N, L, E, generator=g(1)), torch.randn(N, L, E, generator=g(2))
    kr, kc = torch.randn(N, W, E, generator=g(3)), torch.randn(N, H, E, generator=g(4))
    v = torch.randn(N, H, W, E, generator=g(5))
    gout = torch.randn(N, L, E, generator=g(6))
    mr = mc = None
    if masked:
        mr = torch.zeros(N, W, dtype=torch.uint8)
        mc = torch.zeros(N, H, dtype=torch.uint8)
        mr[0, W - 2:] = 1
        mc[0, H - 3:] = 1
    ins = [t.to(DEV).requires_grad_(True) for t in (qr, qc, kr, kc, v)]
    out = ops.rcda_core(*ins, mr.to(DEV) if masked else None, mc.to(DEV) if masked else None, nh)
    out.backward(gout.to(DEV))
    ins64 = [t.double().requires_grad_(True) for t in (qr, qc, kr, kc, v)]
    ref = rcda_core_ref(*ins64, mr, mc, nh)
    ref.backward(gout.double())
    close(out, ref, msg="rcda out", **tol(precision))
    for name, a, b in zip(("dq_row", "dq_col", "dk_row", "dk_col", "dv"), ins, ins64):
        close(a.grad, b.grad, msg=name, rtol=5e-4, atol_scale=tol(precision)["atol_scale"])


@pytest.mark.parametrize("N,L,H,W,masked", [(2, 300, 50, 50, False), (2, 77, 17, 5, True), (2, 2500, 50, 50, True), (1, 130, 70, 40, True)])
@pytest.mark.parametrize("slices", [1, 2, 3, 4, 6, 8])
def test_rcda_forward_key_row_slices(N, L, H, W, masked, slices, monkeypatch):
    """cdetr_rcda_fwd_desc.ws: the two-step forward with the key rows cut into `slices` workgroups per query block (partials exchanged
    through the scratch, summed in slice order by the last arrival) against the fp64 restatement; the saved softmaxes are those of the
    unsliced launch bit for bit, repeated launches agree bit for bit, and the arrival counters are left zero."""
    from counting_detr_amd import ops
    nh, E = 8, 256
    mk = lambda *s, seed: torch.randn(*s, generator=g(seed))
    qr, qc, kr, kc, v = mk(N, L, E, seed=1), mk(N, L, E, seed=2), mk(N, W, E, seed=3), mk(N, H, E, seed=4), mk(N, H, W, E, seed=5)
    mr = mc = None
    if masked:
        mr = torch.zeros(N, W, dtype=torch.uint8)
        mc = torch.zeros(N, H, dtype=torch.uint8)
        mr[0, W - 2:] = 1
        mc[0, H - 3:] = 1
    args = [t.to(DEV) for t in (qr, qc, kr, kc, v)] + [mr.to(DEV) if masked else None, mc.to(DEV) if masked else None, nh]
    monkeypatch.setattr(ops, "PRECISION", 1)
    monkeypatch.setattr(ops, "RCDA_SLICES", False)
    o1, ar1, ac1 = ops.rcda_fwd_raw(*args)
    monkeypatch.setattr(ops, "RCDA_SLICES", True)
    monkeypatch.setenv("CDETR_RCDA_HS", str(slices))
    o2, ar2, ac2 = ops.rcda_fwd_raw(*args)
    o3, _, _ = ops.rcda_fwd_raw(*args)
    torch.cuda.synchronize()
    assert torch.equal(ar1, ar2) and torch.equal(ac1, ac2)
    assert torch.equal(o2, o3)
    assert int(ops.splitk_ws()[:4096].abs().sum()) == 0
    ref = rcda_core_ref(qr.double(), qc.double(), kr.double(), kc.double(), v.double(), mr, mc, nh)
    close(o2, ref, msg="rcda out, sliced", **tol(1))
    close(o2, o1, msg="sliced vs unsliced", rtol=1e-5, atol_scale=1e-6)


@pytest.mark.parametrize("fuse_dq,fuse_dk", [(False, False), (True, False), (True, True)])
def test_rcda_backward_fusion_levels_agree(fuse_dq, fuse_dk, monkeypatch, precision):
    """The attention backward with the query / key gradients computed by separate GEMM / weight-gradient launches from the saved
    logit gradients, with dq fused into the dS kernel, and with dq and dk fused (logit gradients never written): same gradients."""
    from counting_detr_amd import ops
    N, L, H, W = 2, 300, 25, 38
    nh, E = 8, 256
    mk = lambda *s, seed: torch.randn(*s, generator=g(seed))
    q_row, q_col = mk(N, L, E, seed=1), mk(N, L, E, seed=2)
    k_row, k_col, v = mk(N, W, E, seed=3), mk(N, H, E, seed=4), mk(N, H, W, E, seed=5)
    args = [t.to(DEV) for t in (q_row, q_col, k_row, k_col, v)]
    dO = mk(N, L, E, seed=6).to(DEV)
    o, a_row, a_col = ops.rcda_fwd_raw(*args, None, None, nh)
    monkeypatch.setattr(ops, "FUSE_RCDA_DQ", True)
    monkeypatch.setattr(ops, "FUSE_RCDA_DK", True)
    ref = ops.rcda_bwd_raw(dO, *args, a_row, a_col, nh)
    monkeypatch.setattr(ops, "FUSE_RCDA_DQ", fuse_dq)
    monkeypatch.setattr(ops, "FUSE_RCDA_DK", fuse_dk)
    got = ops.rcda_bwd_raw(dO, *args, a_row, a_col, nh)
    for name, a, b in zip(("dq_row", "dq_col", "dk_row", "dk_col", "dv"), got, ref):
        close(a, b, msg=name, rtol=2e-4, atol_scale=6e-5)


def test_multihead_rcda_module_vs_oracle(precision):
    """Full module (projections + core + out_proj) against the oracle restatement, both mask branches."""
    from counting_detr_amd.transformer import MultiheadRCDA
    from oracle import model as OM
    E, nh, N, H, W, L = 256, 8, 2, 9, 14, 40
    m = MultiheadRCDA(E, nh).to(DEV)
    sd = {"a." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    v = torch.randn(N, H, W, E, generator=g(1))
    kr = v + torch.randn(N, 1, W, E, generator=g(2))
    kc = v + torch.randn(N, H, 1, E, generator=g(3))
    qr, qc = torch.randn(N, L, E, generator=g(4)), torch.randn(N, L, E, generator=g(5))
    mask = torch.zeros(N, H, W, dtype=torch.bool)
    mask[1, :, W - 3:] = True
    mask[1, H - 2:, :] = True
    ins = [t.to(DEV).requires_grad_(True) for t in (qr, qc, kr, kc, v)]
    out, _ = m(*ins, key_padding_mask=mask.to(DEV))
    ins_o = [t.clone().requires_grad_(True) for t in (qr, qc, kr, kc, v)]
    ref = OM.rcda(*ins_o, sd, "a", mask=mask, nh=nh)
    gout = torch.randn(ref.shape, generator=g(6))
    out.backward(gout.to(DEV))
    ref.backward(gout)
    close(out, ref, rtol=1e-3, msg="module out")          # north_star: within 1e-3 rel of the fp32 reference
    for a, b, nm in zip(ins, ins_o, ("qr", "qc", "kr", "kc", "v")):
        close(a.grad, b.grad, rtol=1e-3, msg="d" + nm)
    for k, p in m.named_parameters():
        close(p.grad, sd["a." + k].grad, rtol=1e-3, msg="d" + k)


# ----------------------------------------------------------------------------------------------------- matcher
G45 = ["q300_t37", "q576_t200", "q900_t56", "q300_t450", "q900_t900", "b2_q40", "b2_q300", "t0"]
# FSC-147's crowded images (up to 3731 targets, A2/data/fsc147.py:80-84 -> A2/models/matcher.py:229-247): reference vectors of
# oracle/gen_golden.py g45L; they reach the LDS-resident solver's column limit and the [Q][T] (T > Q) layout at every query count
G45L = ["q300_t3000", "q576_t3731", "q900_t3000", "b2_q300_t2100", "q300_t1100"]


def _g45_file(name):
    return "g45_large_t.npz" if name in G45L else "g45_matcher_criterion.npz"


@pytest.mark.parametrize("name", G45 + G45L)
def test_matcher_golden(golden, name):
    """Device cost + device LSAP reproduce the REFERENCE's Hungarian indices bit-exactly (golden vectors)."""
    from counting_detr_amd.matcher import OriginalHungarianMatcher
    z = golden(_g45_file(name))
    B = int(z[f"{name}/B"])
    outs = {k: torch.from_numpy(z[f"{name}/{k}"]).to(DEV) for k in ("pred_logits", "pred_boxes")}
    tg = []
    for b in range(B):
        bx = torch.from_numpy(z[f"{name}/tgt{b}"]).reshape(-1, 4).to(DEV)
        tg.append({"boxes": bx, "labels": torch.zeros(bx.shape[0], dtype=torch.int64, device=DEV)})
    idx = OriginalHungarianMatcher(2, 5, 2)(outs, tg)
    for b in range(B):
        assert idx[b][0].dtype == torch.int64 and not idx[b][0].is_cuda
        assert np.array_equal(idx[b][0].numpy(), z[f"{name}/idx_i{b}"]), f"idx_i image {b}"
        assert np.array_equal(idx[b][1].numpy(), z[f"{name}/idx_j{b}"]), f"idx_j image {b}"


@pytest.mark.parametrize("name", ["q300_t37", "q576_t200", "q900_t56", "q300_t450", "q900_t900", "b2_q40", "b2_q300", "negvar", "t0"] + G45L)
@pytest.mark.parametrize("fused", [False, True])
def test_criterion_golden(golden, name, fused):
    """SetCriterion (fused kernel and tensor-op composition) vs the REFERENCE's losses and input gradients (golden vectors)."""
    from counting_detr_amd.anchor_detr import SetCriterion
    from counting_detr_amd.matcher import OriginalHungarianMatcher
    z = golden(_g45_file(name))
    B = int(z[f"{name}/B"])
    outs = {k: torch.from_numpy(z[f"{name}/{k}"]).to(DEV).requires_grad_(True) for k in ("pred_logits", "pred_boxes", "pred_vars")}
    tg = []
    for b in range(B):
        bx = torch.from_numpy(z[f"{name}/tgt{b}"]).reshape(-1, 4).to(DEV)
        tg.append({"boxes": bx, "labels": torch.zeros(bx.shape[0], dtype=torch.int64, device=DEV)})
    wd = {"loss_ce": 2, "loss_bbox": 5, "loss_giou": 2, "loss_variance": 2}
    crit = SetCriterion(1, OriginalHungarianMatcher(2, 5, 2), wd, ["labels", "boxes", "cardinality", "vars"], focal_alpha=0.25)
    crit.fused = fused
    losses = crit(outs, tg)
    assert (crit.last_total is not None) == fused
    for k in ("loss_ce", "class_error", "cardinality_error", "loss_bbox", "loss_giou", "loss_variance"):
        np.testing.assert_allclose(float(losses[k]), float(z[f"{name}/L_{k}"]), rtol=1e-4, atol=1e-6, err_msg=k, equal_nan=True)
    if f"{name}/g_pred_logits" in z.files:
        total = sum(losses[k] * wd[k] for k in losses if k in wd)
        total.backward()
        for k in outs:
            ref = torch.from_numpy(z[f"{name}/g_{k}"])
            close(outs[k].grad, ref, rtol=2e-4, atol_scale=1e-5, msg="d" + k)


def test_match_cost_values(golden):
    from counting_detr_amd import ops
    from oracle import criterion as OC
    z = golden("g45_matcher_criterion.npz")
    name = "b2_q40"
    logits = torch.from_numpy(z[f"{name}/pred_logits"])
    boxes = torch.from_numpy(z[f"{name}/pred_boxes"])
    tgs = [torch.from_numpy(z[f"{name}/tgt{b}"]).reshape(-1, 4) for b in range(2)]
    plan = ops.MatchPlan([t.shape[0] for t in tgs], 40, DEV)
    cost = ops.match_cost(logits.to(DEV), boxes.to(DEV), torch.cat(tgs).to(DEV), plan).cpu()
    for b in range(2):
        T = tgs[b].shape[0]
        blk = cost[plan.cost_off_host[b]: plan.cost_off_host[b] + 40 * T]
        blk = blk.view(T, 40).t() if T < 40 else blk.view(40, T)
        ref = OC.match_cost(logits[b], boxes[b], tgs[b])
        np.testing.assert_allclose(blk.numpy(), ref.numpy(), rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("Q,T,kind", [(300, 37, "float"), (64, 64, "ties"), (50, 120, "ties"), (120, 50, "ints"),
                                      (900, 900, "float"), (300, 1500, "float"), (7, 1, "float"), (300, 120, "float"),
                                      (300, 120, "ties"), (300, 120, "ints"), (576, 200, "ties"), (130, 128, "ties"),
                                      (65, 64, "ints"), (320, 319, "ties"), (512, 3, "ints"), (1000, 64, "ties"),
                                      (300, 3000, "float"), (300, 3731, "ints"), (576, 3731, "float"), (900, 3000, "ties"),
                                      (300, 3800, "float"), (900, 2049, "ints")])
def test_lsap_vs_scipy(Q, T, kind):
    """The device solver returns scipy's (row_ind, col_ind), ties included, when fed the same matrix."""
    from scipy.optimize import linear_sum_assignment
    from counting_detr_amd import ops
    rng = np.random.default_rng(Q * 7 + T)
    if kind == "float":
        c = rng.standard_normal((Q, T)).astype(np.float32)
    elif kind == "ties":
        c = rng.integers(0, 2, (Q, T)).astype(np.float32)
    else:
        c = rng.integers(0, 10, (Q, T)).astype(np.float32)
    plan = ops.MatchPlan([T], Q, DEV)
    solver_layout = c.T.copy() if T < Q else c
    cost = torch.from_numpy(np.ascontiguousarray(solver_layout).reshape(-1)).to(DEV)
    idx_i, idx_j, status = ops.lsap(cost, plan)
    assert int(status[0]) == 0
    ri, ci = linear_sum_assignment(c)
    m = min(Q, T)
    assert np.array_equal(idx_i[0, :m].cpu().numpy(), ri)
    assert np.array_equal(idx_j[0, :m].cpu().numpy(), ci)


def test_lsap_invalid_cost_raises():
    from counting_detr_amd import ops
    plan = ops.MatchPlan([3], 4, DEV)
    c = torch.zeros(12)
    c[5] = float("nan")
    _, _, status = ops.lsap(c.to(DEV), plan)
    assert int(status[0]) == 2


# ----------------------------------------------------------------------------------------------------- layer-level fusion
@pytest.mark.parametrize("rows,C", [(600, 256), (5000, 256), (77, 1024), (3, 512)])
def test_layer_norm(rows, C):
    from counting_detr_amd import ops
    x = torch.randn(rows, C, generator=g(1)) * 2 + 0.5
    w, b = 1 + 0.1 * torch.randn(C, generator=g(2)), 0.1 * torch.randn(C, generator=g(3))
    gy = torch.randn(rows, C, generator=g(4))
    xd = x.to(DEV).requires_grad_(True)
    wd, bd = torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(b.to(DEV))
    y = ops.layer_norm(xd, wd, bd)
    y.backward(gy.to(DEV))
    x64, w64, b64 = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    y64 = F.layer_norm(x64, (C,), w64, b64, 1e-5)
    y64.backward(gy.double())
    close(y, y64, rtol=1e-5, msg="ln y")
    close(xd.grad, x64.grad, rtol=1e-5, msg="ln dx")
    close(wd.grad, w64.grad, rtol=1e-5, msg="ln dgamma")
    close(bd.grad, b64.grad, rtol=1e-5, msg="ln dbeta")


def test_encoder_layer_fused_equals_unfused(precision):
    """ops.EncoderLayerFn (one node, hand-scheduled backward) == the op-by-op autograd composition, incl. every gradient."""
    from counting_detr_amd.transformer import TransformerEncoderLayerSpatial
    torch.manual_seed(0)
    N, H, W, Cc = 2, 9, 14, 256
    res = []
    for fused in (False, True):
        torch.manual_seed(1)
        layer = TransformerEncoderLayerSpatial(Cc, 1024, 8).to(DEV)
        layer.fused = fused
        src = torch.randn(N, H, W, Cc, generator=g(1)).to(DEV).requires_grad_(True)
        pr = torch.randn(N, W, Cc, generator=g(2)).to(DEV).requires_grad_(True)
        pc = torch.randn(N, H, Cc, generator=g(3)).to(DEV).requires_grad_(True)
        mr = torch.zeros(N, W, dtype=torch.uint8); mr[1, W - 3:] = 1
        mc = torch.zeros(N, H, dtype=torch.uint8); mc[1, H - 2:] = 1
        y = layer(src, mr.to(DEV), mc.to(DEV), pr, pc)
        y.backward(torch.randn(y.shape, generator=g(4)).to(DEV))
        res.append((y, src.grad, pr.grad, pc.grad, {k: p.grad for k, p in layer.named_parameters()}))
    (y0, s0, r0, c0, p0), (y1, s1, r1, c1, p1) = res
    close(y1, y0, rtol=2e-5, msg="enc out")
    close(s1, s0, rtol=1e-4, msg="enc dsrc")
    close(r1, r0, rtol=1e-4, msg="enc dposrow")
    close(c1, c0, rtol=1e-4, msg="enc dposcol")
    for k in p0:
        close(p1[k], p0[k], rtol=1e-4, msg="enc d" + k)


@pytest.mark.parametrize("used", [(2,), (0, 1, 2)])
def test_decoder_stack_fused_equals_unfused(precision, used):
    """ops.DecoderStackFn (all decoder layers in one node, hand-scheduled backward, shared-input gradients accumulated inside)
    == the op-by-op autograd composition: layer outputs and every gradient, with only the last or with all layer outputs used."""
    from counting_detr_amd import ops
    from counting_detr_amd.transformer import TransformerDecoderLayer
    N, L, H, W, E = 2, 40, 9, 14, 256
    res = []
    for fused in (False, True):
        torch.manual_seed(3)
        layers = [TransformerDecoderLayer(E, 1024, 8).to(DEV) for _ in range(3)]
        mk = lambda shape, seed: torch.randn(*shape, generator=g(seed)).to(DEV).requires_grad_(True)   # noqa: E731
        tgt, qp, qx, qy = mk((N, L, E), 1), mk((N, L, E), 2), mk((N, L, E), 3), mk((N, L, E), 4)
        mem, krm, kcm = mk((N, H, W, E), 5), mk((N, W, E), 6), mk((N, H, E), 7)
        mr = torch.zeros(N, W, dtype=torch.uint8); mr[1, W - 3:] = 1
        mc = torch.zeros(N, H, dtype=torch.uint8); mc[1, H - 2:] = 1
        mr, mc = mr.to(DEV), mc.to(DEV)
        if fused:
            outs = ops.DecoderStackFn.apply(tgt, qp, qx, qy, mem, krm, kcm, mr, mc, layers, tgt)
        else:
            outs, x = [], tgt
            for layer in layers:
                x = layer(x, qp, qx, qy, mem, krm, kcm, mr, mc)
                outs.append(x)
        loss = sum((outs[i] * torch.randn(outs[i].shape, generator=g(10 + i)).to(DEV)).sum() for i in used)
        loss.backward()
        grads = {f"{i}.{k}": p.grad for i, layer in enumerate(layers) for k, p in layer.named_parameters()}
        res.append(([o.detach() for o in outs], [t.grad for t in (tgt, qp, qx, qy, mem, krm, kcm)], grads))
    (o0, i0, p0), (o1, i1, p1) = res
    for a, b in zip(o1, o0):
        close(a, b, rtol=2e-5, msg="dec out")
    for name, a, b in zip(("tgt", "qpos", "qx", "qy", "mem", "krm", "kcm"), i1, i0):
        close(a, b, rtol=1e-4, msg="dec d" + name)
    for k in p0:
        close(p1[k], p0[k], rtol=1e-4, msg="dec d" + k)


@pytest.mark.parametrize("N,L", [(2, 300), (1, 900), (2, 37), (1, 129), (1, 64), (2, 5)])
def test_mha_core(N, L, precision):
    from counting_detr_amd import ops
    nh, E = 8, 256
    qk = torch.randn(N, L, 2 * E, generator=g(1))
    v = torch.randn(N, L, E, generator=g(2))
    go = torch.randn(N, L, E, generator=g(3))
    qkd, vd = qk.to(DEV).requires_grad_(True), v.to(DEV).requires_grad_(True)
    o = ops.mha_core(qkd, vd, nh)
    old = ops.MHA_BWD_BF16
    ops.MHA_BWD_BF16 = _BWD_MODE == 3                   # (opt-in in the product: cdetr_mha_bwd precision 3 is exercised here)
    try:
        o.backward(go.to(DEV))
    finally:
        ops.MHA_BWD_BF16 = old
    qk64, v64 = qk.double().requires_grad_(True), v.double().requires_grad_(True)
    hs = lambda t: t.reshape(N, L, nh, 32).permute(0, 2, 1, 3)   # noqa: E731
    a = ((hs(qk64[..., :E]) * 32 ** -0.5) @ hs(qk64[..., E:]).transpose(-1, -2)).softmax(-1)
    ref = (a @ hs(v64)).permute(0, 2, 1, 3).reshape(N, L, E)
    ref.backward(go.double())
    close(o, ref, rtol=2e-5, msg="mha o")
    close(qkd.grad, qk64.grad, rtol=1e-4, msg="mha dqk")
    close(vd.grad, v64.grad, rtol=1e-4, msg="mha dv")


def test_weight_mirror_and_dgrad(precision):
    """cdetr_weight_mirror: Wt[c][tap][o] = W[o][tap][c] * scale[o]; data gradients through the mirror == through the weight."""
    from counting_detr_amd import ops
    torch.manual_seed(5)
    dev = "cuda"
    w3 = torch.randn(64, 32, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    s3 = torch.rand(64, device=dev) + 0.5
    wl = torch.randn(160, 96, device=dev)
    w1 = torch.randn(48, 64, 1, 1, device=dev).contiguous(memory_format=torch.channels_last)
    mir = ops.WeightMirror([(w3, s3), (wl, None), (w1, None)], [(w3, s3), (wl, None)])
    mir.refresh()
    m3, ld3, sp3, h3 = mir.lookup(w3, s3)
    assert torch.equal(h3[:32 * 9 * 64].view(32, 9 * 64), (w3 * s3.view(-1, 1, 1, 1)).permute(1, 2, 3, 0).reshape(32, 9 * 64).bfloat16())     # plain-bf16 image
    ref3 = (w3 * s3.view(-1, 1, 1, 1)).permute(1, 2, 3, 0).reshape(32, 9 * 64)        # [c][tap][o]
    assert ld3 == 9 * 64 and torch.equal(m3[:32 * 9 * 64].view(32, 9 * 64), ref3)

    def unsplit(buf, rows, klen):      # [rows][klen/32][hi 32 | lo 32] bf16 -> (hi, lo) fp32 [rows, klen]
        v = buf[:rows * klen].view(torch.bfloat16).view(rows, klen // 32, 2, 32).float()
        return v[:, :, 0].reshape(rows, klen), v[:, :, 1].reshape(rows, klen)
    hi, lo = unsplit(sp3, 32, 9 * 64)
    assert torch.equal(hi, ref3.bfloat16().float()) and torch.equal(lo, (ref3 - ref3.bfloat16().float()).bfloat16().float())
    ml, ldl, spl, _ = mir.lookup(wl[32:96])
    assert ldl == 160 and float(ml[0]) == float(wl[32, 0]) and float(ml[160]) == float(wl[32, 1]) and spl is not None
    full = mir.lookup(wl)[0][:96 * 160].view(96, 160)
    assert torch.equal(full, wl.t())
    assert mir.lookup(wl[8:40])[2] is None            # a k offset that is not a multiple of 32 has no pre-split view
    assert mir.lookup(w3, None) is None and mir.lookup(torch.randn(4, 4, device=dev)) is None
    # forward images: W * scale, k = (tap, c)
    f3 = mir.lookup_fwd(w3, s3)
    reff = (w3 * s3.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(64, 9 * 32)
    hi, lo = unsplit(f3, 64, 9 * 32)
    assert torch.equal(hi, reff.bfloat16().float()) and torch.equal(lo, (reff - reff.bfloat16().float()).bfloat16().float())
    fl = mir.lookup_fwd(wl[32:96])
    hi, _ = unsplit(fl, 64, 96)
    assert torch.equal(hi, wl[32:96].bfloat16().float()) and mir.lookup_fwd(w1) is None
    # dgrad equivalence (conv 3x3 with scale + gate/resid, linear slice)
    dz = torch.randn(2, 10, 12, 64, device=dev)
    gate = torch.randn(2, 10, 12, 32, device=dev)
    a = ops.conv_dgrad(dz, w3, s3, (10, 12), stride=1, pad=1, dil=1, gate=gate)
    dy = torch.randn(300, 64, device=dev)
    b = ops.linear_dgrad(dy, wl[32:96])
    xin = torch.randn(2, 10, 12, 32, device=dev)
    bias3 = torch.randn(64, device=dev)
    f0 = ops.conv_fwd(xin, w3, s3, bias3, stride=1, pad=1, dil=1, relu=True)
    xl = torch.randn(5000, 96, device=dev)
    l0 = ops.linear_fwd(xl, wl[32:96])
    dyb = torch.randn(5000, 64, device=dev)
    b0 = ops.linear_dgrad(dyb, wl[32:96])
    ops.MIRROR = mir
    try:
        a2 = ops.conv_dgrad(dz, w3, s3, (10, 12), stride=1, pad=1, dil=1, gate=gate)
        b2 = ops.linear_dgrad(dy, wl[32:96])
        f1 = ops.conv_fwd(xin, w3, s3, bias3, stride=1, pad=1, dil=1, relu=True)      # pre-split forward operand
        l1 = ops.linear_fwd(xl, wl[32:96])                                             # M = 5000: the tile kernel with B_split
        b1 = ops.linear_dgrad(dyb, wl[32:96])
    finally:
        ops.MIRROR = None
    if precision == 1:       # same products in the same order: the pre-split path is bit-identical to the in-kernel split
        assert torch.equal(l1, l0)
    close(f1, f0.double().cpu(), **tol(precision))
    close(l1, l0.double().cpu(), **tol(precision))
    close(b1, b0.double().cpu(), **tol(precision))
    close(a2, a.double().cpu(), **tol(precision))
    close(b2, b.double().cpu(), **tol(precision))
    close(b2, dy.double().cpu() @ wl[32:96].double().cpu(), **tol(precision))


@pytest.mark.parametrize("variant", [0, 3, 4, 9, 10, 13, 14, 16])
def test_gemm_variants_ragged_shapes(variant, monkeypatch):
    """Every tile variant of the implicit-GEMM kernel (CDETR_GEMM_VARIANT), bf16x3, on shapes whose M / N / K are NOT multiples
    of the tile (clamped-load tails), k-contiguous and n-contiguous weight operands, dense and 3x3 (strided / dilated) rows,
    with the fused epilogue -- against fp64.  Guards the unconditional-load / clamping logic of the prefetch pipeline."""
    from counting_detr_amd import ops, _ffi
    monkeypatch.setenv("CDETR_GEMM_VARIANT", str(variant))
    old = ops.PRECISION
    ops.PRECISION = 1
    try:
        rng = np.random.default_rng(100 + variant)
        cases = [(1, 32, 32), (63, 36, 64), (65, 132, 96), (130, 260, 160), (257, 64, 32), (600, 256, 256), (1000, 516, 128), (3, 4, 64),
                 (1237, 132, 192)]
        for M, N, K in cases:
            for bl in (0, 1):
                if variant in (3, 14) and K % 64:
                    continue
                A = torch.randn(M, K, generator=g(M + N))
                Bm = torch.randn(N, K, generator=g(M + K)) if bl == 0 else torch.randn(K, N, generator=g(M + K))
                bias, resid, gate = torch.randn(N, generator=g(1)), torch.randn(M, N, generator=g(2)), torch.randn(M, N, generator=g(3))
                ws = 1 + 0.1 * torch.randn(N if bl == 0 else K, generator=g(4))
                ref = (A.double() @ (Bm.double() * ws.double()[:, None]).t()) if bl == 0 else (A.double() @ (Bm.double() * ws.double()[:, None]))
                ref = torch.relu((ref + bias.double()) * 0.5 + resid.double()) * (gate > 0)
                # kernel epilogue order: gate is applied before relu; both commute here since relu(x)*[g>0] == relu(x*[g>0])
                out = torch.empty(M, N, device=DEV)
                ops.gemm_raw(A.to(DEV), K, Bm.to(DEV), K if bl == 0 else N, out, N, M, N, K, b_layout=bl, bias=bias.to(DEV),
                             w_scale=ws.to(DEV), resid=resid.to(DEV), ldr=N, gate=gate.to(DEV), ldg=N, relu=True, out_scale=0.5)
                close(out, ref, msg=f"variant {variant} M{M} N{N} K{K} bl{bl}", **tol(1))
        # 3x3 conv rows: stride 2 and dilation 2, odd spatial sizes, channel counts off the tile
        for (Cin, Cout, st, pd, dl, H, W) in [(32, 48, 1, 1, 1, 9, 11), (64, 80, 2, 1, 1, 13, 10), (32, 144, 1, 2, 2, 8, 15), (128, 64, 1, 1, 1, 12, 7)]:      # dgrad with taps needs Cout % 16 == 0
            x = torch.randn(2, Cin, H, W, generator=g(Cin + H))
            w = torch.randn(Cout, Cin, 3, 3, generator=g(Cout)) / (Cin * 9) ** 0.5
            y64 = F.conv2d(x.double(), w.double(), stride=st, padding=pd, dilation=dl)
            xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
            wd = w.contiguous(memory_format=torch.channels_last).to(DEV)
            y = ops.conv_fwd(xd, wd, None, None, stride=st, pad=pd, dil=dl)
            close(y.permute(0, 3, 1, 2), y64, msg=f"variant {variant} conv {Cin}->{Cout} s{st} d{dl}", **tol(1))
            gy = torch.randn(y64.shape, generator=g(5))
            x64 = x.double().requires_grad_(True)
            F.conv2d(x64, w.double(), stride=st, padding=pd, dilation=dl).backward(gy.double())
            dx = ops.conv_dgrad(gy.permute(0, 2, 3, 1).contiguous().to(DEV), wd, None, (H, W), stride=st, pad=pd, dil=dl)
            close(dx.permute(0, 3, 1, 2), x64.grad, msg=f"variant {variant} dgrad {Cin}->{Cout}", **tol(1))
    finally:
        ops.PRECISION = old


@pytest.mark.parametrize("wvariant", [0, 1, 2, 3, 4])
def test_wgrad_variants_ragged_shapes(wvariant, monkeypatch, precision):
    """Weight-gradient tile variants on pixel counts / channel counts off the tile, with the fused bias gradient, 1x1 and 3x3."""
    from counting_detr_amd import ops
    monkeypatch.setenv("CDETR_WGRAD_VARIANT", str(wvariant))
    for (P, Nout, Cin) in [(1100, 64, 64), (1500, 132, 68), (2049, 256, 36), (5000, 40, 260)]:
        dY = torch.randn(P, Nout, generator=g(P))
        X = torch.randn(P, Cin, generator=g(P + 1))
        dW = torch.zeros(Nout, Cin, device=DEV)
        db = torch.zeros(Nout, device=DEV)
        for _ in range(2):
            ops.wgrad_raw(dY.to(DEV), Nout, X.to(DEV), Cin, dW, Cin, P, Nout, Cin, dbias=db)
        close(dW, 2 * dY.double().t() @ X.double(), msg=f"wgrad {P}x{Nout}x{Cin}", rtol=5e-4, atol_scale=6e-5)
        close(db, 2 * dY.double().sum(0), msg="dbias", rtol=5e-4, atol_scale=6e-5)
    for (Cin, Cout, st, pd, dl, H, W) in [(36, 40, 1, 1, 1, 19, 21), (64, 136, 2, 1, 1, 33, 30), (32, 64, 1, 2, 2, 28, 25)]:
        x = torch.randn(2, Cin, H, W, generator=g(Cin))
        w64 = (torch.randn(Cout, Cin, 3, 3, generator=g(Cout)).double() / (Cin * 9) ** 0.5).requires_grad_(True)
        y64 = F.conv2d(x.double(), w64, stride=st, padding=pd, dilation=dl)
        gy = torch.randn(y64.shape, generator=g(6))
        y64.backward(gy.double())
        wd = torch.nn.Parameter(w64.detach().float().contiguous(memory_format=torch.channels_last).to(DEV))
        ops.conv_wgrad_(gy.permute(0, 2, 3, 1).contiguous().to(DEV), x.permute(0, 2, 3, 1).contiguous().to(DEV), wd, None, stride=st, pad=pd, dil=dl)
        close(wd.grad, w64.grad, msg=f"conv wgrad {Cin}->{Cout} s{st} d{dl}", rtol=5e-4, atol_scale=6e-5)


@pytest.mark.parametrize("N,nh,L,Wk", [(2, 8, 300, 50), (3, 4, 77, 37), (2, 8, 2500, 50)])
def test_two_level_batch_image_by_head(N, nh, L, Wk, precision):
    """batch_inner: one launch runs the per-head contractions of every image (inner stride = head, outer stride = image):
    dq[n,l,h,:] = sum_w S[n,h,l,w] k[n,w,h,:]  (cdetr_gemm, n-contiguous B)  and  dk[n,w,h,:] += sum_l S[n,h,l,w] q[n,l,h,:]
    (cdetr_wgrad) -- the logits -> q / k gradient step of the RCDA backward."""
    from counting_detr_amd import ops
    E = nh * 32
    Wp = (Wk + 3) & ~3
    S = torch.randn(N, nh, L, Wp, generator=g(1))
    S[..., Wk:] = 0
    k = torch.randn(N, Wk, E, generator=g(2))
    q = torch.randn(N, L, E, generator=g(3))
    Sd, kd, qd = S.to(DEV), k.to(DEV), q.to(DEV)
    dq = torch.empty(N, L, E, device=DEV)
    dk = torch.zeros(N, Wk, E, device=DEV)
    ops.gemm_raw(Sd, Wp, kd, E, dq, E, L, 32, Wk, b_layout=1, batch=N * nh, sA=L * Wp, sB=32, sC=32,
                 batch_inner=nh, sA2=nh * L * Wp, sB2=Wk * E, sC2=L * E)
    ops.wgrad_raw(Sd, Wp, qd, E, dk, E, L, Wk, 32, batch=N * nh, sY=L * Wp, sX=32, sW=32,
                  batch_inner=nh, sY2=nh * L * Wp, sX2=L * E, sW2=Wk * E)
    S64 = S[..., :Wk].double()
    k64 = k.double().view(N, Wk, nh, 32)
    q64 = q.double().view(N, L, nh, 32)
    close(dq.view(N, L, nh, 32), torch.einsum("nhlw,nwhd->nlhd", S64, k64), msg="dq", **tol(precision))
    close(dk.view(N, Wk, nh, 32), torch.einsum("nhlw,nlhd->nwhd", S64, q64), msg="dk", **tol(precision))


@pytest.mark.parametrize("H,W", [(64, 96), (75, 53), (800, 800)])
def test_stem_row_packed_equals_per_tap_form(H, W, precision):
    """The 7x7 / stride-2 stem as 7 row taps of 32 floats on a zero-padded image == the per-element-tap form == F.conv2d (fp64)."""
    from counting_detr_amd.backbone import ResNetBody
    body = ResNetBody(True).to(DEV)
    with torch.no_grad():
        body.conv1.weight.copy_(torch.randn(body.conv1.weight.shape, generator=g(1)) * 0.05)
        body.bn1.weight.copy_(torch.rand(64, generator=g(2)) + 0.5)
        body.bn1.bias.copy_(torch.randn(64, generator=g(3)) * 0.1)
        body.bn1.running_mean.copy_(torch.randn(64, generator=g(4)) * 0.1)
        body.bn1.running_var.copy_(torch.rand(64, generator=g(5)) + 0.5)
    img = torch.randn(2, 3, H, W, generator=g(6))
    y_rows = body.stem_rows(img.to(DEV))
    s, b = body.bn1.affine()
    ref = F.conv2d(img.double(), body.conv1.weight.detach().double().cpu(), stride=2, padding=3)
    ref = (ref * s.double().cpu().view(1, -1, 1, 1) + b.double().cpu().view(1, -1, 1, 1)).clamp_min(0)
    close(y_rows.permute(0, 3, 1, 2), ref, msg="row-packed stem", **tol(precision))


@pytest.mark.parametrize("split", [True, False], ids=["split", "per-group"])
@pytest.mark.parametrize("B,h,w,C,G", [(2, 50, 50, 256, 32), (1, 7, 9, 64, 16), (3, 13, 5, 256, 4), (2, 20, 30, 96, 8), (3, 37, 29, 256, 64), (1, 16, 8, 256, 32),
                                       (2, 24, 36, 256, 32)])
def test_groupnorm_nhwc(B, h, w, C, G, split, monkeypatch):
    """cdetr_groupnorm_fwd_ws / bwd_ws on NHWC == F.group_norm on the channels-first view (fp64), incl. dgamma / dbeta accumulation: the
    split form (C == 256, G = 32 / 64, >= 128 pixels: pixels spread over workgroups, ragged last chunk, a mean far from zero for the merged
    centred moments) and the one-workgroup-per-(image, group) kernels every other shape -- and `split = False` -- runs."""
    from counting_detr_amd import ops
    monkeypatch.setattr(ops, "GN_SPLIT", split)
    x = torch.randn(B, h, w, C, generator=g(1)) * 2 + (30.0 if (h, w) == (24, 36) else 0.5)
    gm, bt = torch.randn(C, generator=g(2)), torch.randn(C, generator=g(3))
    dy = torch.randn(B, h, w, C, generator=g(4))
    xd = x.to(DEV).requires_grad_(True)
    wp, bp = torch.nn.Parameter(gm.to(DEV)), torch.nn.Parameter(bt.to(DEV))
    y = ops.GroupNormNHWCFn.apply(xd, wp, bp, G, 1e-5)
    y.backward(dy.to(DEV))
    x64 = x.double().requires_grad_(True)
    g64, b64 = gm.double().requires_grad_(True), bt.double().requires_grad_(True)
    r = F.group_norm(x64.permute(0, 3, 1, 2), G, g64, b64, 1e-5).permute(0, 2, 3, 1)
    r.backward(dy.double())
    close(y, r, rtol=1e-4, msg="gn y")
    close(xd.grad, x64.grad, rtol=2e-4, msg="gn dx")
    close(wp.grad, g64.grad, rtol=2e-4, msg="gn dgamma")
    close(bp.grad, b64.grad, rtol=2e-4, msg="gn dbeta")


def test_gemm_group_matches_individual_calls(precision, monkeypatch):
    """cdetr_gemm_group (ops.gemm_queue): few-row problems of three k-lengths, the 64x128 class with a pre-split weight image,
    a data-gradient operand (n-contiguous weight), epilogues (bias, residual, ReLU, gate) and 14 problems of one class (two
    grouped launches) == the same calls issued one by one (bit-identical) == fp64 within tolerance.  (Bit identity holds per kernel
    class: the single-call route of few-row long reductions through the split 64x64 tile is switched off here.)"""
    from counting_detr_amd import ops
    monkeypatch.setenv("CDETR_GEMM_FEWROW_SPLIT", "0")
    shapes = [(600, 256, 256), (100, 256, 256), (600, 256, 512), (600, 256, 1024), (37, 40, 36), (5000, 256, 256), (5000, 256, 256),
              (5000, 256, 256), (4000, 128, 64)] + [(50 + 13 * k, 256, 256) for k in range(14)]
    Ws = [torch.randn(N, K, generator=g(11 * i)) / K ** 0.5 for i, (M, N, K) in enumerate(shapes)]
    Wd = [w.to(DEV) for w in Ws]
    entries = [(w, None) for w in Wd if w.shape[1] % 32 == 0 and w.shape[0] % 32 == 0]
    mirror = ops.WeightMirror([], entries)
    mirror.refresh()
    old_m = ops.MIRROR
    ops.MIRROR = mirror
    try:
        xs, outs_q, outs_1 = [], [], []
        for i, (M, N, K) in enumerate(shapes):
            x = torch.randn(M, K, generator=g(7 * i + 1))
            b = torch.randn(N, generator=g(7 * i + 2))
            r = torch.randn(M, N, generator=g(7 * i + 3))
            xs.append((x, b, r, x.to(DEV), b.to(DEV), r.to(DEV)))
        with ops.gemm_queue():
            for i, (M, N, K) in enumerate(shapes):
                x, b, r, xd, bd, rd = xs[i]
                outs_q.append(ops.linear_fwd(xd, Wd[i], bd, relu=(i % 2 == 0), resid=rd if i % 3 == 0 else None))
            gq = ops.linear_dgrad(xs[0][5], Wd[0].t().contiguous().t(), gate=xs[0][5])      # not in the mirror: n-contiguous operand
        for i, (M, N, K) in enumerate(shapes):
            x, b, r, xd, bd, rd = xs[i]
            outs_1.append(ops.linear_fwd(xd, Wd[i], bd, relu=(i % 2 == 0), resid=rd if i % 3 == 0 else None))
        g1 = ops.linear_dgrad(xs[0][5], Wd[0].t().contiguous().t(), gate=xs[0][5])
    finally:
        ops.MIRROR = old_m
    assert torch.equal(gq, g1)
    for i, (M, N, K) in enumerate(shapes):
        x, b, r = xs[i][:3]
        assert torch.equal(outs_q[i], outs_1[i]), f"grouped != single for {shapes[i]}"
        ref = x.double() @ Ws[i].double().t() + b.double()
        if i % 3 == 0:
            ref = ref + r.double()
        if i % 2 == 0:
            ref = ref.clamp_min(0)
        close(outs_q[i], ref, msg=f"gemm group {shapes[i]}", **tol(precision))


def test_wgrad_group_matches_individual_calls(precision):
    """cdetr_wgrad_group: a queue of independent parameter gradients (few-pixel, 64x64 transpose-read and other kernel classes,
    two of them accumulating into the SAME dW / dbias, 19 problems = two grouped launches of one class) == the sum of the
    individual contractions (fp64 reference)."""
    from counting_detr_amd import ops
    shapes = [(600, 256, 256), (100, 256, 256), (5000, 256, 256), (5000, 1024, 256), (5000, 256, 1024), (1500, 132, 68),
              (5000, 2048, 512), (77, 4, 256), (600, 2, 256)] + [(300 + 17 * k, 64, 96) for k in range(10)]
    refs, bufs, keep = [], [], []
    with ops.wgrad_queue():
        for k, (P, Nout, Cin) in enumerate(shapes):
            dY = torch.randn(P, Nout, generator=g(3 * k))
            X = torch.randn(P, Cin, generator=g(3 * k + 1))
            dW = torch.zeros(Nout, Cin, device=DEV)
            db = torch.zeros(Nout, device=DEV)
            dYd, Xd = dY.to(DEV), X.to(DEV)
            ops.wgrad_raw(dYd, Nout, Xd, Cin, dW, Cin, P, Nout, Cin, dbias=db, may_defer=True)
            ref_w, ref_b = dY.double().t() @ X.double(), dY.double().sum(0)
            if k in (0, 2):       # a second problem accumulating into the same buffers
                dY2 = torch.randn(P, Nout, generator=g(1000 + k))
                X2 = torch.randn(P, Cin, generator=g(2000 + k))
                dY2d, X2d = dY2.to(DEV), X2.to(DEV)
                ops.wgrad_raw(dY2d, Nout, X2d, Cin, dW, Cin, P, Nout, Cin, dbias=db, may_defer=True)
                ref_w = ref_w + dY2.double().t() @ X2.double()
                ref_b = ref_b + dY2.double().sum(0)
                keep += [dY2d, X2d]
            refs.append((ref_w, ref_b)); bufs.append((dW, db)); keep += [dYd, Xd]
        assert float(bufs[0][0].abs().max()) == 0.0          # nothing ran yet
    for (P, Nout, Cin), (rw, rb), (dW, db) in zip(shapes, refs, bufs):
        close(dW, rw, msg=f"group wgrad {P}x{Nout}x{Cin}", rtol=5e-4, atol_scale=6e-5)
        close(db, rb, msg=f"group dbias {P}x{Nout}x{Cin}", rtol=5e-4, atol_scale=6e-5)


@pytest.mark.parametrize("target", [0, 1536, 6144, 50000])
def test_wgrad_workgroup_target(target):
    """cdetr_wgrad_desc.wg_target (the step's last weight-gradient launch asks for many short pixel slices: it has the chip to itself):
    single launches and a grouped launch, fp32 operands and bf16 twins, 1x1 and 3x3 -- the same sums for every target (fp64 reference)."""
    from counting_detr_amd import ops
    cases = [(20000, 512, 128, 1), (20000, 128, 512, 1), (5000, 256, 1024, 1), (1500, 132, 68, 1), (3200, 128, 128, 9)]
    for grouped in (False, True):
        refs, outs, keep = [], [], []
        with ops.wgrad_queue():
            for k, (P, Nout, Cin, taps) in enumerate(cases):
                geom = None
                if taps == 9:
                    geom, Ho, Wo = ops.conv_geom_fwd(40, 40, 3, 3, 1, 1, 1)
                    x4 = torch.randn(2, 40, 40, Cin, generator=g(5 * k))
                    dY = torch.randn(2 * Ho * Wo, Nout, generator=g(5 * k + 1))
                    X = x4.reshape(-1, Cin)
                    w = torch.zeros(Nout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
                    y = torch.nn.functional.conv2d(x4.bfloat16().double().permute(0, 3, 1, 2), w, padding=1)
                    (y * dY.bfloat16().double().view(2, Ho, Wo, Nout).permute(0, 3, 1, 2)).sum().backward()
                    ref = w.grad.permute(0, 2, 3, 1).reshape(Nout, 9 * Cin)
                else:
                    dY = torch.randn(P, Nout, generator=g(5 * k + 1))
                    X = torch.randn(P, Cin, generator=g(5 * k))
                    ref = dY.bfloat16().double().t() @ X.bfloat16().double()
                dYd, Xd = dY.to(DEV), X.to(DEV)
                dW = torch.zeros(Nout, taps * Cin, device=DEV)
                tw = k % 2 == 0 and Nout % 8 == 0 and Cin % 8 == 0
                d16, x16 = (dYd.bfloat16(), Xd.bfloat16()) if tw else (None, None)
                ops.wgrad_raw(dYd, Nout, Xd, Cin, dW, taps * Cin, dYd.shape[0], Nout, Cin, taps=taps, geom=geom, may_defer=grouped, dY16=d16, X16=x16,
                              precision=3, wg_target=target)
                refs.append(ref); outs.append(dW); keep += [dYd, Xd, d16, x16]
        for (P, Nout, Cin, taps), rw, dW in zip(cases, refs, outs):
            close(dW, rw, msg=f"wg_target {target} grouped {grouped}: {P}x{Nout}x{Cin}x{taps}", rtol=5e-4, atol_scale=6e-5)


def test_sine_embed_matches_reference_formula():
    """cdetr_sine_embed fwd / bwd == the tensor-op formula of A2/models/transformer.py:474-494 (fp64)."""
    import math
    from counting_detr_amd.transformer import pos2posemb1d, pos2posemb2d

    def sine64(pos, nfeat, T=10000):
        dim_t = torch.arange(nfeat, dtype=torch.float64)
        dim_t = T ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / nfeat)
        px = (pos * (2 * math.pi))[..., None] / dim_t
        return torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=-1).flatten(-2)
    p1 = torch.rand(2, 37, generator=g(1))
    p2 = torch.rand(2, 300, 2, generator=g(2))
    a1 = p1.to(DEV).requires_grad_(True); a2 = p2.to(DEV).requires_grad_(True)
    e1, e2 = pos2posemb1d(a1), pos2posemb2d(a2)
    g1, g2 = torch.randn(e1.shape, generator=g(3)), torch.randn(e2.shape, generator=g(4))
    (e1 * g1.to(DEV)).sum().backward(); (e2 * g2.to(DEV)).sum().backward()
    r1 = p1.double().requires_grad_(True); r2 = p2.double().requires_grad_(True)
    f1 = sine64(r1, 256); f2 = torch.cat((sine64(r2[..., 1], 128), sine64(r2[..., 0], 128)), dim=-1)
    (f1 * g1.double()).sum().backward(); (f2 * g2.double()).sum().backward()
    assert e1.shape == (2, 37, 256) and e2.shape == (2, 300, 256)
    np.testing.assert_allclose(e1.detach().cpu().numpy(), f1.detach().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(e2.detach().cpu().numpy(), f2.detach().numpy(), rtol=0, atol=2e-5)
    close(a1.grad, r1.grad, rtol=1e-4, msg="dpos 1d")
    close(a2.grad, r2.grad, rtol=1e-4, msg="dpos 2d")


# ------------------------------------------------------------------------------------------------------- optimizer tail
@pytest.mark.parametrize("n", [1, 7, 1024, 262147, 37449312])
def test_sumsq_value_and_bit_reproducibility(n):
    """cdetr_sumsq (global gradient norm of clip_grad_norm_, A2/engine.py:54-57): value against a float64 sum, and the
    SAME bits on every call -- data-parallel ranks holding the same reduced gradient must clip identically."""
    from counting_detr_amd import _ffi
    gvec = (torch.randn(n, generator=g(11)) * 0.3).to(DEV)
    out = torch.empty(1, device=DEV)
    ws = torch.empty(2048, device=DEV)
    seen = set()
    for _ in range(6):
        _ffi.check(_ffi.lib().cdetr_sumsq(gvec.data_ptr(), n, out.data_ptr(), ws.data_ptr(), _ffi.stream_ptr()), "cdetr_sumsq")
        seen.add(out.cpu().numpy().tobytes())
    assert len(seen) == 1, "sum of squares changes from call to call"
    ref = float((gvec.double() ** 2).sum())
    assert abs(float(out[0]) - ref) <= 2e-6 * ref + 1e-30


@pytest.mark.parametrize("bwd,lim,twins", [(2, 3e-3, False), (3, 5e-3, False), (3, 5e-3, True)])
@pytest.mark.parametrize("Cin,Cout,k,stride,pad,dil,H,W", [(256, 256, 3, 1, 1, 1, 50, 50), (512, 2048, 1, 1, 0, 1, 50, 50), (128, 128, 3, 2, 1, 1, 100, 100),
                                                        (512, 512, 3, 1, 2, 2, 50, 50), (72, 40, 3, 1, 1, 1, 21, 19)])
def test_reduced_term_backward_kernels(Cin, Cout, k, stride, pad, dil, H, W, bwd, lim, twins):
    """ops.PRECISION_BWD 2 ("bf16x2": weights / activations rounded to bf16, the incoming gradient split) and 3 (plain bf16): data
    and weight gradients of a convolution through the weight mirror vs fp64.  A product then carries a 2^-9 rounding; over a
    K-term sum the error stays below ~1e-3 of the result's scale (bar 2.5e-3); the forward is untouched (bf16x3, 2e-4)."""
    from counting_detr_amd import ops
    ops.PRECISION, old = 1, ops.PRECISION
    ops.PRECISION_BWD = bwd
    try:
        B = 2
        x = torch.randn(B, Cin, H, W, generator=g(11))
        w = torch.randn(Cout, Cin, k, k, generator=g(12)) / (Cin * k * k) ** 0.5
        sc = torch.rand(Cout, generator=g(13)) + 0.5
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        wd = torch.nn.Parameter(w.contiguous(memory_format=torch.channels_last).to(DEV))
        wd.grad = torch.zeros_like(wd)
        scd, bd = sc.to(DEV), torch.zeros(Cout, device=DEV)
        mirror = ops.WeightMirror([(wd.data, scd)], [(wd.data, scd)] if Cin % 32 == 0 else [])
        mirror.refresh()
        ops.MIRROR = mirror
        y = ops.conv_fwd(xd, wd.data, scd, bd, stride=stride, pad=pad, dil=dil)
        gy = torch.randn(y.shape, generator=g(14)).to(DEV)
        dx = ops.conv_dgrad(gy, wd.data, scd, (H, W), stride=stride, pad=pad, dil=dil, dz16=gy.to(torch.bfloat16) if twins else None)
        if twins:       # the kernel fed from bf16 twins of both operands (what the producers' epilogues write next to the fp32 tensors)
            ops.conv_wgrad_(gy, xd, wd, scd, stride=stride, pad=pad, dil=dil, dz16=gy.to(torch.bfloat16), x16=xd.to(torch.bfloat16))
        else:
            ops.conv_wgrad_(gy, xd, wd, scd, stride=stride, pad=pad, dil=dil)
        x64 = x.double().requires_grad_(True)
        w64 = w.double().requires_grad_(True)
        y64 = F.conv2d(x64, w64 * sc.double().view(-1, 1, 1, 1), None, stride, pad, dil)
        y64.backward(gy.double().cpu().permute(0, 3, 1, 2))
        close(y.permute(0, 3, 1, 2), y64, msg="y (forward stays bf16x3)", rtol=2e-4, atol_scale=6e-5)
        close(dx.permute(0, 3, 1, 2), x64.grad, msg=f"dx bwd={bwd}", rtol=lim, atol_scale=0.0)
        close(wd.grad, w64.grad, msg=f"dW bwd={bwd}", rtol=lim, atol_scale=0.0)
    finally:
        ops.MIRROR = None
        ops.PRECISION = old


@pytest.mark.parametrize("kp", [1, 2, 4])
@pytest.mark.parametrize("Cin,Cout,k,stride,pad,dil,H,W", [(256, 256, 3, 1, 1, 1, 50, 50), (512, 2048, 1, 1, 0, 1, 50, 50), (128, 128, 3, 2, 1, 1, 100, 100),
                                                        (512, 512, 3, 1, 2, 2, 50, 50), (72, 40, 3, 1, 1, 1, 21, 19), (64, 256, 1, 1, 0, 1, 37, 41)])
def test_twin_fed_weight_gradient_pixel_blocks(Cin, Cout, k, stride, pad, dil, H, W, kp, monkeypatch):
    """The twin-fed weight-gradient kernels stage KP 32-pixel blocks per barrier (csrc/igemm.hip wgrad_tr16_body, CDETR_WGRAD_KP): every KP
    must give the fp64 result of the SAME bf16 operands (gradient of A2/models/resnet.py:140-160's convolutions), single launch and grouped
    launch, slices that end inside a staged tile, pixel counts that are no multiple of 32 KP."""
    from counting_detr_amd import ops
    monkeypatch.setenv("CDETR_WGRAD_KP", str(kp))
    ops.PRECISION, old = 1, ops.PRECISION
    ops.PRECISION_BWD = 3
    try:
        B = 2
        Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        x = torch.randn(B, H, W, Cin, generator=g(21)).to(DEV)
        gy = torch.randn(B, Ho, Wo, Cout, generator=g(22)).to(DEV)
        sc = (torch.rand(Cout, generator=g(23)) + 0.5).to(DEV)
        x16, gy16 = x.bfloat16(), gy.bfloat16()
        ws = [torch.nn.Parameter(torch.zeros(Cout, Cin, k, k, device=DEV).contiguous(memory_format=torch.channels_last)) for _ in range(3)]
        for w in ws:
            w.grad = torch.zeros_like(w)
        ops.conv_wgrad_(gy, x, ws[0], sc, stride=stride, pad=pad, dil=dil, dz16=gy16, x16=x16)
        with ops.wgrad_queue():                        # the grouped launch (two problems of the 64x64 class ride together when eligible)
            ops.conv_wgrad_(gy, x, ws[1], sc, stride=stride, pad=pad, dil=dil, dz16=gy16, x16=x16)
            ops.conv_wgrad_(gy, x, ws[2], sc, stride=stride, pad=pad, dil=dil, dz16=gy16, x16=x16)
        x64 = x16.double().cpu().permute(0, 3, 1, 2)
        w64 = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
        y64 = F.conv2d(x64, w64 * sc.double().cpu().view(-1, 1, 1, 1), None, stride, pad, dil)
        y64.backward(gy16.double().cpu().permute(0, 3, 1, 2))
        scale = w64.grad.abs().max().item()
        for i, w in enumerate(ws):          # exact bf16 products, fp32 accumulation over <= 20000 pixels (+ fp32 atomics across the slices)
            err = (w.grad.double().cpu() - w64.grad).abs().max().item()
            assert err <= 3e-5 * scale, f"dW kp={kp} launch {i}: {err:.3e} vs scale {scale:.3e}"
    finally:
        ops.PRECISION = old


# ----------------------------------------------------------------------------------------------------- glue kernels (csrc/glue.hip)
@pytest.mark.parametrize("B,H,W,h,w", [(2, 800, 800, 50, 50), (3, 128, 160, 8, 10), (2, 96, 75, 6, 5), (1, 37, 53, 3, 4)])
def test_mask_prep_matches_the_tensor_composition(B, H, W, h, w):
    """cdetr_mask_prep == F.interpolate(nearest) + first row / column + mask2pos (A2/models/backbone.py:143, transformer.py:497-503)."""
    from counting_detr_amd import ops
    from counting_detr_amd.transformer import mask2pos
    mask = torch.ones(B, H, W, dtype=torch.bool)
    gg = g(5)
    for b in range(B):
        hh = int(torch.randint(H // 2, H + 1, (1,), generator=gg)) if b else H
        ww = int(torch.randint(W // 2, W + 1, (1,), generator=gg)) if b else W
        mask[b, :hh, :ww] = False
    mi = ops.mask_prep(mask.to(DEV), h, w)
    m = F.interpolate(mask[None].float(), size=(h, w)).to(torch.bool)[0]
    assert torch.equal(mi.m.cpu(), m)
    assert torch.equal(mi.mask_row.cpu().bool(), m[:, 0, :]) and torch.equal(mi.mask_col.cpu().bool(), m[:, :, 0])
    pc, pr = mask2pos(m)
    np.testing.assert_allclose(mi.pos_col.cpu().numpy(), pc.numpy(), rtol=1e-6)
    np.testing.assert_allclose(mi.pos_row.cpu().numpy(), pr.numpy(), rtol=1e-6)
    ext = torch.stack([(~m[:, :, 0]).sum(1), (~m[:, 0, :]).sum(1)], 1).float()
    assert torch.equal(mi.extent.cpu(), ext)


@pytest.mark.parametrize("per_image", [True, False])
def test_exemplar_feature_and_concat_free_projection(per_image, precision):
    """ops.ExemplarFeatureFn / ops.AggrProjFn vs the reference composition: pf = mean of x at the truncated box centres
    (A2/models/backbone.py:122-131), y = conv1x1(cat([x, x * pf])) (:132-136 + anchor_detr.py:119): values and every gradient."""
    from counting_detr_amd import ops
    B, h, w, Cc, d, K = 2, 9, 7, 64, 32, 3
    x = torch.randn(B, h, w, Cc, generator=g(1))
    W = torch.randn(d, 2 * Cc, 1, 1, generator=g(2)) / (2 * Cc) ** 0.5
    bias = torch.randn(d, generator=g(3))
    rects = torch.tensor([[[.10, .10, .20, .20], [.40, .40, .50, .55], [.70, .20, .999, .999]],
                          [[.55, .60, .70, .80], [.05, .30, .20, .45], [-1., -1., -1., -1.]]])
    extent = torch.tensor([[float(h), float(w)], [6.0, 5.0]])
    gy = torch.randn(B, h, w, d, generator=g(4))
    # reference composition in fp64
    x64, W64, b64 = x.double().requires_grad_(True), W.double().requires_grad_(True), bias.double().requires_grad_(True)
    pfs = []
    for b in range(B):
        rb = rects[b if per_image else 0]
        hv, wv = (extent[b] if per_image else torch.tensor([float(h), float(w)]))
        rows = []
        for r in rb:
            if r[2] < 0:
                continue
            xc = int((r[0] * wv + r[2] * wv) / 2)
            yc = int((r[1] * hv + r[3] * hv) / 2)
            rows.append(x64[b, min(yc, h - 1), min(xc, w - 1)])
        pfs.append(torch.stack(rows).mean(0))
    pf64 = torch.stack(pfs)
    feat = torch.cat([x64, x64 * pf64[:, None, None, :]], -1)
    y64 = feat @ W64.reshape(d, -1).t() + b64
    y64.backward(gy.double())
    xd = x.to(DEV).requires_grad_(True)
    Wd, bd = torch.nn.Parameter(W.to(DEV)), torch.nn.Parameter(bias.to(DEV))
    y = ops.AggrProjFn.apply(xd, rects.to(DEV), extent.to(DEV), per_image, Wd, bd)
    y.backward(gy.to(DEV))
    t = tol(precision)
    close(y, y64, msg="y", **t)
    close(xd.grad, x64.grad, msg="dx", **t)
    close(Wd.grad.reshape(d, -1), W64.grad.reshape(d, -1), msg="dW", **t)
    close(bd.grad, b64.grad, msg="db", **t)
    xe = x.to(DEV).requires_grad_(True)
    pf = ops.ExemplarFeatureFn.apply(xe, rects.to(DEV), extent.to(DEV), per_image)
    close(pf, pf64, rtol=1e-6, msg="pf")
    gp = torch.randn(B, Cc, generator=g(6))
    pf.backward(gp.to(DEV))
    (gx64,) = torch.autograd.grad(pf64, x64, gp.double())
    close(xe.grad, gx64, rtol=1e-6, msg="d pf / dx")


@pytest.mark.parametrize("L", [1, 3])
def test_box_head_tail_matches_torch(L):
    """ops.BoxHeadFn == sigmoid(cat([tmp[..., :2] + inverse_sigmoid(ref), tmp[..., 2:]])) incl. torch's clamp-backward conventions at
    the edges (reference points exactly 0, 1, 1e-5, below / above the clamps)  (A2/models/transformer.py:193-203, util/misc.py:475-479)."""
    from counting_detr_amd import ops
    from counting_detr_amd.transformer import inverse_sigmoid
    B, Q = 2, 37
    ref = torch.rand(B, Q, 2, generator=g(1))
    ref[0, :8, 0] = torch.tensor([0.0, 1.0, 1e-5, 1 - 1e-5, 5e-6, 1 - 5e-6, -0.1, 1.2])
    tmp = torch.randn(*((L,) if L > 1 else ()), B, Q, 4, generator=g(2))
    gb = torch.randn(tmp.shape, generator=g(3))
    r0, t0 = ref.clone().requires_grad_(True), tmp.clone().requires_grad_(True)
    want = torch.cat([t0[..., :2] + inverse_sigmoid(r0), t0[..., 2:]], -1).sigmoid()
    want.backward(gb)
    r1, t1 = ref.to(DEV).requires_grad_(True), tmp.to(DEV).requires_grad_(True)
    got = ops.BoxHeadFn.apply(t1, r1)
    got.backward(gb.to(DEV))
    close(got, want, rtol=1e-6, msg="boxes")
    close(t1.grad, t0.grad, rtol=1e-6, msg="d tmp")
    close(r1.grad, r0.grad, rtol=1e-5, msg="d ref")


def test_sine_embed_xy_matches_the_one_coordinate_form():
    from counting_detr_amd import ops
    p = torch.rand(2, 300, 2, generator=g(1))
    ge = (torch.randn(2, 300, 256, generator=g(2)).to(DEV), torch.randn(2, 300, 256, generator=g(3)).to(DEV))
    a = p.to(DEV).requires_grad_(True)
    ex, ey = ops.sine_embed_xy(a, 256)
    (ex * ge[0]).sum().backward(retain_graph=True)
    (ey * ge[1]).sum().backward()
    b = p.to(DEV).requires_grad_(True)
    fx, fy = ops.sine_embed(b[..., 0], 256), ops.sine_embed(b[..., 1], 256)
    ((fx * ge[0]).sum() + (fy * ge[1]).sum()).backward()
    assert torch.equal(ex, fx) and torch.equal(ey, fy)
    close(a.grad, b.grad, rtol=1e-6, msg="d points")


@pytest.mark.parametrize("slices", [2, 3, 4])
@pytest.mark.parametrize("prec,lim", [(0, 2e-5), (1, 5e-5), (3, 2e-2)])
def test_gemm_split_reduction(slices, prec, lim, monkeypatch):
    """cdetr_gemm_desc.splitk_ws: the reduction of a tile cut into 2-4 slices that meet in the scratch (the last workgroup to arrive
    adds them in slice order and runs the epilogue).  Against fp64 and against the unsplit launch: dense rows and 3x3 taps (a slice
    boundary inside a tap and on one), ragged M / N, bias + residual + gate + relu + bf16 twin, every tile class, called twice in a
    row (the arrival counters must come back to zero) -- and bit-identical across repeats (ordered sum)."""
    from counting_detr_amd import ops
    cases = [(5000, 256, 1024, 0), (5000, 256, 256, 0), (1237, 132, 512, 0), (600, 512, 2048, 0), (130, 260, 160, 0), (2000, 1024, 256, 4),
             (5000, 256, 1024, 14), (333, 70, 640, 14), (1237, 260, 1024, 16)]
    for M, N, K, variant in cases:
        monkeypatch.setenv("CDETR_GEMM_VARIANT", str(variant))
        A = torch.randn(M, K, generator=g(M + N)).to(DEV)
        Bm = (torch.randn(N, K, generator=g(M + K)) / K ** 0.5).to(DEV)
        bias, resid, gate = torch.randn(N, generator=g(1)).to(DEV), torch.randn(M, N, generator=g(2)).to(DEV), torch.randn(M, N, generator=g(3)).to(DEV)
        ref = torch.relu((A.double() @ Bm.double().t() + bias.double()) * 0.5 + resid.double()) * (gate > 0)
        outs = []
        for s in (1, slices, slices):
            monkeypatch.setenv("CDETR_GEMM_SPLITK", "0" if s == 1 else str(s))
            out, out16 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            ops.gemm_raw(A, K, Bm, K, out, N, M, N, K, bias=bias, resid=resid, ldr=N, gate=gate, ldg=N, relu=True, out_scale=0.5,
                         precision=prec, C16=out16)
            outs.append(out)
            close(out, ref, rtol=lim, atol_scale=lim, msg=f"M{M} N{N} K{K} slices {s}")
            assert torch.equal(out16, out.to(torch.bfloat16))
        close(outs[1], outs[0].double(), rtol=1e-5, atol_scale=1e-5, msg="split vs unsplit")
        assert torch.equal(outs[1], outs[2]), "the ordered sum of the slices must not depend on arrival order"
    monkeypatch.setenv("CDETR_GEMM_VARIANT", "0")
    for (Cin, Cout, st, pd, dl, H, W) in [(256, 256, 1, 1, 1, 50, 50), (64, 80, 2, 1, 1, 13, 10), (512, 512, 1, 2, 2, 50, 50)]:
        x = torch.randn(2, Cin, H, W, generator=g(Cin + H))
        w = torch.randn(Cout, Cin, 3, 3, generator=g(Cout)) / (Cin * 9) ** 0.5
        y64 = F.conv2d(x.double(), w.double(), stride=st, padding=pd, dilation=dl)
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        wd = w.contiguous(memory_format=torch.channels_last).to(DEV)
        old = ops.PRECISION
        ops.PRECISION = prec if prec != 3 else 1
        try:
            monkeypatch.setenv("CDETR_GEMM_SPLITK", str(slices))
            y = ops.conv_fwd(xd, wd, None, None, stride=st, pad=pd, dil=dl)
            l2 = lim if prec != 3 else 5e-5
            close(y.permute(0, 3, 1, 2), y64, rtol=l2, atol_scale=l2, msg=f"conv {Cin}->{Cout} slices {slices}")
        finally:
            ops.PRECISION = old
    ws = ops.splitk_ws()
    assert int(ws[:4096].abs().sum()) == 0, "arrival counters left non-zero"


@pytest.mark.parametrize("N,H,W", [(2, 50, 50), (1, 7, 5), (3, 16, 20)])
def test_layernorm_backward_with_merged_addends(N, H, W):
    """cdetr_layernorm_bwd_merge == cdetr_grad_merge / cdetr_bcast_add2_sum followed by cdetr_layernorm_bwd: the incoming gradient as a sum of
    up to three tensors (+ the accumulator side effects) and, optionally, two addends broadcast over the map; rows not a multiple of the
    kernel's row batch; against the composition of the separate launches (bit-identical sums are not required: the order of additions differs)."""
    from counting_detr_amd import ops
    C = 256
    R = N * H * W
    gen = g(R)
    x = torch.randn(R, C, generator=gen).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=gen)).to(DEV), torch.randn(C, generator=gen).to(DEV)
    y, mu, rs = ops.ln_fwd_raw(x, gamma, beta, 1e-5) if hasattr(ops, "ln_fwd_raw") else (None, None, None)
    if y is None:
        mu = x.mean(1)
        rs = 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)
    dy, g1, g2 = [torch.randn(R, C, generator=gen).to(DEV) for _ in range(3)]
    Br, Bc = torch.randn(N * W, C, generator=gen).to(DEV), torch.randn(N * H, C, generator=gen).to(DEV)
    add = torch.randn(R, C, generator=gen).to(DEV)

    def run(merged, bcast, with_acc):
        gw, gb = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        a1 = torch.ones(R, C, device=DEV) if with_acc else None
        a2 = torch.full((R, C), 2.0, device=DEV) if with_acc else None
        if merged:
            dx = ops.ln_bwd_raw(dy, x, mu, rs, gamma, gw, gb, add=add, merge=(g1, g2, a1, a2),
                                bcast=(Br, Bc, 1.0 / H, 1.0 / W, H, W) if bcast else None)
        else:
            if bcast:
                t = ops.bcast_add2_sum(dy.view(N, H, W, C), g1.view(N, H, W, C), g2.view(N, H, W, C), Br, Bc, 1.0 / H, 1.0 / W).view(R, C)
                if with_acc:
                    a1 += g1; a2 += g2
            else:
                t = ops.grad_merge(dy, g1, g2, a1, a2)
            dx = ops.ln_bwd_raw(t, x, mu, rs, gamma, gw, gb, add=add)
        return dx, gw, gb, a1, a2

    for bcast in (False, True):
        for with_acc in (False, True):
            ref, out = run(False, bcast, with_acc), run(True, bcast, with_acc)
            for a, b, name in zip(out, ref, ("dx", "dgamma", "dbeta", "acc1", "acc2")):
                if a is not None:
                    close(a, b, rtol=2e-5, atol_scale=2e-5, msg=f"{name} bcast={bcast} acc={with_acc}")
    # fp64 check of the merged form itself
    xe = x.double().cpu().requires_grad_(True)
    ye = F.layer_norm(xe, (C,), gamma.double().cpu(), beta.double().cpu(), 1e-5)
    tot = (dy + g1 + g2).double().cpu() + (Br.double().cpu().view(N, 1, W, C) / H + Bc.double().cpu().view(N, H, 1, C) / W).expand(N, H, W, C).reshape(R, C)
    ye.backward(tot)
    dx = run(True, True, False)[0]
    close(dx, xe.grad + add.double().cpu(), rtol=5e-5, atol_scale=5e-5, msg="merged vs fp64")


def test_fused_heads_node_equals_per_layer_nodes():
    """ops.HeadsFn (the class / box / variance heads of A2/models/transformer.py:79-107 as one autograd node, hand-scheduled backward
    with the ReLU masks in the data-gradient epilogues) == the per-layer LinearFn nodes: outputs bit-identical (the same launches),
    input gradient and every parameter gradient equal to the reduced-term arithmetic's noise."""
    from counting_detr_amd import ops
    from counting_detr_amd.transformer import MLP, Linear, mlps_levelwise
    torch.manual_seed(0)
    ce, be, ve = Linear(256, 2).to(DEV), MLP(256, 256, 4, 3).to(DEV), MLP(256, 256, 2, 3).to(DEV)
    x = torch.randn(2, 300, 256, device=DEV)
    ups = [torch.randn(2, 300, n, device=DEV) for n in (2, 4, 2)]
    res = []
    for fused in (False, True):
        for p in list(ce.parameters()) + list(be.parameters()) + list(ve.parameters()):
            p.grad = torch.zeros_like(p)
        xi = x.clone().requires_grad_(True)
        if fused:
            hp = [ce.weight, ce.bias] + [t for l in be.layers for t in (l.weight, l.bias)] + [t for l in ve.layers for t in (l.weight, l.bias)]
            outs = ops.HeadsFn.apply(xi, *hp)
        else:
            outs = mlps_levelwise([ce, be, ve], xi)
        with ops.wgrad_queue():
            torch.autograd.backward(list(outs), ups)
        res.append(([o.detach().clone() for o in outs], xi.grad.clone(),
                    [p.grad.clone() for p in list(ce.parameters()) + list(be.parameters()) + list(ve.parameters())]))
    (o0, dx0, g0), (o1, dx1, g1) = res
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    close(dx1, dx0, rtol=1e-5, msg="dx")
    for i, (a, b) in enumerate(zip(g0, g1)):
        close(b, a, rtol=1e-5, msg=f"param grad {i}")


def test_encoder_prologue_single_launch():
    """cdetr_posadd2_hw_reduce == cdetr_posadd2 + cdetr_hw_reduce (bit for bit: the same device bodies), also on a non-square map and a
    channel count that takes the generic path."""
    from counting_detr_amd import ops
    for (N, H, W, Cc) in [(2, 50, 50, 256), (1, 24, 36, 256), (2, 7, 5, 64)]:
        g = torch.Generator().manual_seed(H * W)
        X = torch.randn(N, H, W, Cc, generator=g).to(DEV)
        Pr, Pc = torch.randn(N, W, Cc, generator=g).to(DEV), torch.randn(N, H, Cc, generator=g).to(DEV)
        Qr, Qc = ops.posadd2(X, Pr, Pc)
        Kr, Kc = ops.hw_reduce(X, X, Pr, Pc, 1.0 / H, 1.0 / W)
        q1, q2, k1, k2 = ops.posadd2_hw_reduce(X, Pr, Pc)
        assert torch.equal(q1, Qr) and torch.equal(q2, Qc) and torch.equal(k1, Kr) and torch.equal(k2, Kc)
        close(k1, X.mean(1) + Pr, rtol=1e-5)
        close(k2, X.mean(2) + Pc, rtol=1e-5)


def test_fused_pos_mlp_node_equals_per_layer_nodes():
    """ops.PosMlpFn (Linear -> ReLU -> Linear over several inputs, one autograd node) == the per-layer LinearFn nodes: same outputs, input
    gradients (only where one is needed) and parameter gradients."""
    from counting_detr_amd import ops
    from counting_detr_amd.transformer import PosMLP
    torch.manual_seed(1)
    mlp = PosMLP(256).to(DEV)
    xs0 = [torch.randn(2, 50, 256, device=DEV), torch.randn(2, 37, 256, device=DEV), torch.randn(2, 300, 256, device=DEV)]
    ups = [torch.randn_like(x) for x in xs0]
    res = []
    for fused in (False, True):
        for p in mlp.parameters():
            p.grad = torch.zeros_like(p)
        xs = [xs0[0].clone(), xs0[1].clone(), xs0[2].clone().requires_grad_(True)]
        old, ops.FUSED_HEADS = ops.FUSED_HEADS, fused
        try:
            from counting_detr_amd.transformer import pos_mlp_many
            ys = pos_mlp_many(mlp, xs)
            with ops.wgrad_queue():
                torch.autograd.backward(ys, ups)
        finally:
            ops.FUSED_HEADS = old
        res.append(([y.detach().clone() for y in ys], xs[2].grad.clone(), [p.grad.clone() for p in mlp.parameters()]))
    (y0, dx0, g0), (y1, dx1, g1) = res
    for a, b in zip(y0, y1):
        assert torch.equal(a, b)
    close(dx1, dx0, rtol=1e-5, msg="dx")
    for i, (a, b) in enumerate(zip(g0, g1)):
        close(b, a, rtol=1e-5, msg=f"param grad {i}")


def test_forward_weight_images_streaming_kernel():
    """cdetr_weight_images (the forward operands' pre-split images W * scale -> [hi 32 | lo 32] groups as one streaming pass): bit-exact against
    torch for 16-byte aligned sources and for a parameter view at an odd offset of its arena (4-byte aligned only), with and without a row scale,
    sizes that are not a multiple of the 4096-weight block."""
    from counting_detr_amd import ops
    torch.manual_seed(3)
    arena = torch.randn(2 + 96 * 160 + 6 + 64 * 9 * 32, device=DEV)
    wl = arena[2:2 + 96 * 160].view(96, 160)                                  # 8 bytes into the arena
    w3 = arena[2 + 96 * 160 + 6:].view(64, 3, 3, 32).permute(0, 3, 1, 2)       # channels_last [64, 32, 3, 3], 32 bytes in
    assert wl.data_ptr() % 16 == 8 and w3.is_contiguous(memory_format=torch.channels_last)
    s3 = torch.rand(64, device=DEV) + 0.5
    big = torch.randn(512, 9 * 512, device=DEV)[:, :9 * 512]
    mir = ops.WeightMirror([], [(wl, None), (w3, s3), (big, None)])
    mir.refresh("fwd")
    torch.cuda.synchronize()

    def unsplit(buf, rows, klen):
        v = buf[:rows * klen].view(torch.bfloat16).view(rows, klen // 32, 2, 32).float()
        return v[:, :, 0].reshape(rows, klen), v[:, :, 1].reshape(rows, klen)
    for w, sc, rows, klen, ref in ((wl, None, 96, 160, wl), (w3, s3, 64, 288, (w3 * s3.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(64, 288)),
                                   (big, None, 512, 4608, big)):
        hi, lo = unsplit(mir.lookup_fwd(w, sc), rows, klen)
        assert torch.equal(hi, ref.bfloat16().float()), "hi image"
        assert torch.equal(lo, (ref - ref.bfloat16().float()).bfloat16().float()), "lo image"
