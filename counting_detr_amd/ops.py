"""Host-side operators over the C-ABI (include/cdetr_hip.h): thin launch wrappers + torch.autograd.Functions.

Layout conventions of the product path: activations are fp32 NHWC (`[N,H,W,C]` contiguous) or `[rows, C]`;
conv weights keep the reference's logical shape `[Cout,Cin,kh,kw]` (state-dict compatible) but live in
channels_last memory (`[Cout][kh][kw][Cin]` physically) so a filter tap is a contiguous K-run;
parameter gradients are ACCUMULATED in place into `param.grad` (the trainer's flat gradient arena) by the
weight-gradient kernels -- autograd only carries activation gradients.
"""
import torch

from . import _ffi
from ._ffi import ConvGeom, GemmDesc, RcdaBwdDesc, RcdaFwdDesc, WgradDesc, check, lib, ptr, stream_ptr

import ctypes as C


# Optional per-launch instrumentation (bench.py's roofline leg): a list collecting (family, algorithmic FLOPs,
# start event, end event) for every matrix-core launch; HIP events are recorded on the launch stream itself.
PROFILE = None

# Matrix-core arithmetic of the implicit-GEMM / weight-gradient kernels: 1 (default) = split-bf16 x3: every fp32 operand is
# split hi+lo and a product costs 3 bf16 MFMAs with fp32 accumulation (~5e-6 relative, ~5x less matrix-pipe time);
# 0 = fp32 MFMA (exact fp32 products).  Both modes pass the same parity suite (1e-3 rel, bit-exact Hungarian indices).
import os as _os
PRECISION = int(_os.environ.get("CDETR_PRECISION", "1"))


class _Timed:
    def __init__(self, family, flops, tag=None):
        self.family, self.flops, self.tag = family, flops, tag

    def __enter__(self):
        if PROFILE is not None:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
            PROFILE.append((self.family, self.flops, self.t0, t1, self.tag))
        return False


def _geom(mode=_ffi.ROWS_DENSE, Ha=0, Wa=0, Hc=0, Wc=0, kh=1, kw=1, stride=1, pad=0, dil=1):
    return ConvGeom(mode, Ha, Wa, Hc, Wc, kh, kw, stride, pad, dil)


def gemm_raw(A, lda, B, ldb, Cout, ldc, M, N, K, taps=1, b_layout=0, bias=None, w_scale=None, resid=None, ldr=0,
             gate=None, ldg=0, relu=False, out_scale=1.0, geom=None, batch=1, sA=0, sB=0, sC=0):
    d = GemmDesc()
    d.M, d.N, d.K, d.taps, d.batch, d.b_layout, d.relu, d.out_scale = M, N, K, taps, batch, b_layout, int(relu), out_scale
    d.precision = PRECISION
    d.A, d.lda, d.sA = ptr(A), lda, sA
    d.B, d.ldb, d.sB = ptr(B), ldb, sB
    d.C, d.ldc, d.sC = ptr(Cout), ldc, sC
    d.w_scale, d.bias = ptr(w_scale), ptr(bias)
    d.resid, d.ldr = ptr(resid), ldr
    d.gate, d.ldg = ptr(gate), ldg
    d.g = geom if geom is not None else _geom()
    with _Timed("igemm", 2.0 * M * N * K * taps * batch, (M, N, K, taps, b_layout, batch)):
        check(lib().cdetr_gemm(C.byref(d), stream_ptr()), "cdetr_gemm")


def wgrad_raw(dY, ldy, X, ldx, dW, ldw, P, Nout, Cin, taps=1, w_scale=None, geom=None, batch=1, sY=0, sX=0, sW=0,
              dbias=None):
    d = WgradDesc()
    d.P, d.Nout, d.Cin, d.taps, d.batch = P, Nout, Cin, taps, batch
    d.precision = PRECISION
    d.dY, d.ldy, d.sY = ptr(dY), ldy, sY
    d.X, d.ldx, d.sX = ptr(X), ldx, sX
    d.dW, d.ldw, d.sW = ptr(dW), ldw, sW
    d.w_scale = ptr(w_scale)
    d.dbias = ptr(dbias)
    d.g = geom if geom is not None else _geom()
    with _Timed("wgrad", 2.0 * P * Nout * Cin * taps * batch, (P, Nout, Cin, taps, -1, batch)):
        check(lib().cdetr_wgrad(C.byref(d), stream_ptr()), "cdetr_wgrad")


def colsum_(X2d, out):
    """out[n] += sum_m X2d[m][n]"""
    M, N = X2d.shape
    check(lib().cdetr_colsum(ptr(X2d), X2d.stride(0), M, N, ptr(out), stream_ptr()), "cdetr_colsum")


def relu_mask(y, dy, scale=1.0):
    """dz = (y > 0) ? dy * scale : 0 in one pass."""
    dy = dy.contiguous()
    dz = torch.empty_like(dy)
    check(lib().cdetr_relu_mask(ptr(y), ptr(dy), ptr(dz), dy.numel(), scale, stream_ptr()), "cdetr_relu_mask")
    return dz


def grad_buffer(p):
    """The in-place gradient accumulator of a parameter (the trainer pre-binds views of its flat arena)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


# ----------------------------------------------------------------------------------------------------- linear
def linear_fwd(x2d, weight, bias=None, relu=False, resid=None, out_scale=1.0, out=None):
    M, K = x2d.shape
    N = weight.shape[0]
    y = out if out is not None else torch.empty((M, N), device=x2d.device, dtype=torch.float32)
    gemm_raw(x2d, x2d.stride(0), weight, weight.stride(0), y, y.stride(0), M, N, K, bias=bias, relu=relu,
             resid=resid, ldr=(resid.stride(0) if resid is not None else 0), out_scale=out_scale)
    return y


def linear_dgrad(dy2d, weight, gate=None, resid=None):
    """dx = dy . W   (W [N_out][K_in] read as the n-contiguous operand)"""
    M, N = dy2d.shape
    K = weight.shape[1]
    dx = torch.empty((M, K), device=dy2d.device, dtype=torch.float32)
    gemm_raw(dy2d, dy2d.stride(0), weight, weight.stride(0), dx, K, M, K, N, b_layout=1,
             gate=gate, ldg=(gate.stride(0) if gate is not None else 0),
             resid=resid, ldr=(resid.stride(0) if resid is not None else 0))
    return dx


class LinearFn(torch.autograd.Function):
    """y = act((x W[lo:hi]^T + b[lo:hi]) * out_scale + resid) -- the F.linear sites of the reference
    (A2/models/transformer.py:412-439, row_column_decoupled_attention.py:165-208,311).
    `wparam` / `bparam` are the leaf parameters (rows lo:hi are used: the 5-way in_proj of RCDA is one parameter);
    their gradients are accumulated IN PLACE into `.grad` rows lo:hi by the weight-gradient kernel."""

    @staticmethod
    def forward(ctx, x, wparam, bparam, lo, hi, relu, resid, out_scale):
        shp = x.shape
        w = wparam.detach()[lo:hi]
        b = bparam.detach()[lo:hi] if bparam is not None else None
        x2d = x.reshape(-1, shp[-1])
        if x2d.stride(-1) != 1 or (x2d.stride(0) & 3) or (x2d.data_ptr() & 15):
            x2d = x2d.contiguous()
        r2d = None
        if resid is not None:
            r2d = resid.reshape(-1, w.shape[0])
            if r2d.stride(-1) != 1:
                r2d = r2d.contiguous()
        y = linear_fwd(x2d, w, b, relu, r2d, out_scale)
        ctx.relu, ctx.out_scale, ctx.has_resid, ctx.lo, ctx.hi = relu, out_scale, resid is not None, lo, hi
        ctx.wparam, ctx.bparam = wparam, bparam
        ctx.save_for_backward(x2d, y if relu else None)
        ctx.xshape = shp
        return y.reshape(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2d, y = ctx.saved_tensors
        wparam, bparam, lo, hi = ctx.wparam, ctx.bparam, ctx.lo, ctx.hi
        w = wparam.detach()[lo:hi]
        dy2d = dy.reshape(-1, w.shape[0])
        if not dy2d.is_contiguous():
            dy2d = dy2d.contiguous()
        if ctx.relu:   # y = relu((x W^T + b) * out_scale + resid): the mask applies to the residual branch too
            if ctx.has_resid:
                dy2d = relu_mask(y, dy2d)
                d_resid = dy2d.reshape(dy.shape)
                if ctx.out_scale != 1.0:
                    dy2d = dy2d * ctx.out_scale
            else:
                dy2d = relu_mask(y, dy2d, ctx.out_scale)
                d_resid = None
        else:
            d_resid = dy2d.reshape(dy.shape) if ctx.has_resid else None
            if ctx.out_scale != 1.0:
                dy2d = dy2d * ctx.out_scale
        dx = None
        if ctx.needs_input_grad[0]:
            dx = linear_dgrad(dy2d, w).reshape(ctx.xshape)
        if wparam.requires_grad:
            gw = grad_buffer(wparam)[lo:hi]
            gb = grad_buffer(bparam)[lo:hi] if (bparam is not None and bparam.requires_grad) else None
            wgrad_raw(dy2d, dy2d.stride(0), x2d, x2d.stride(0), gw, gw.stride(0), dy2d.shape[0], w.shape[0], x2d.shape[1],
                      dbias=gb)
        return dx, None, None, None, None, None, d_resid, None


def linear(x, weight, bias=None, relu=False, resid=None, out_scale=1.0, rows=None):
    lo, hi = rows if rows is not None else (0, weight.shape[0])
    return LinearFn.apply(x, weight, bias, lo, hi, relu, resid, out_scale)


# ----------------------------------------------------------------------------------------------------- conv
def conv_geom_fwd(Hin, Win, kh, kw, stride, pad, dil):
    Hout = (Hin + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    Wout = (Win + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    dense = kh == 1 and kw == 1 and stride == 1 and pad == 0
    g = _geom() if dense else _geom(_ffi.ROWS_CONV_FWD, Hin, Win, Hout, Wout, kh, kw, stride, pad, dil)
    return g, Hout, Wout


def conv_fwd(x, weight, scale, bias, stride=1, pad=0, dil=1, relu=False, resid=None):
    """x [N,H,W,Cin] NHWC -> [N,Ho,Wo,Cout]; weight logical [Cout,Cin,kh,kw] in channels_last memory.
    y = relu?( conv(x, W) * scale[c] + bias[c] + resid )   (A2/models/resnet.py:140-160 + backbone.py:50-60)."""
    Nb, H, W, Cin = x.shape
    Cout, Cin_w, kh, kw = weight.shape
    assert Cin_w == Cin and x.is_contiguous()
    g, Ho, Wo = conv_geom_fwd(H, W, kh, kw, stride, pad, dil)
    y = torch.empty((Nb, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    gemm_raw(x, Cin, weight, kh * kw * Cin, y, Cout, Nb * Ho * Wo, Cout, Cin, taps=kh * kw, w_scale=scale, bias=bias,
             relu=relu, resid=resid, ldr=Cout, geom=g)
    return y


def conv_dgrad(dz, weight, scale, in_hw, stride=1, pad=0, dil=1, gate=None, resid=None):
    """dx [N,Hin,Win,Cin] = conv_transpose(dz * scale, W) (+ resid), zeroed where gate <= 0."""
    Nb, Ho, Wo, Cout = dz.shape
    Cout_w, Cin, kh, kw = weight.shape
    Hin, Win = in_hw
    dense = kh == 1 and kw == 1 and stride == 1 and pad == 0
    g = _geom() if dense else _geom(_ffi.ROWS_CONV_DGRAD, Ho, Wo, Hin, Win, kh, kw, stride, pad, dil)
    dx = torch.empty((Nb, Hin, Win, Cin), device=dz.device, dtype=torch.float32)
    gemm_raw(dz, Cout, weight, Cin, dx, Cin, Nb * Hin * Win, Cin, Cout, taps=kh * kw, b_layout=1, w_scale=scale,
             gate=gate, ldg=Cin, resid=resid, ldr=Cin, geom=g)
    return dx


def conv_wgrad_(dz, x, weight, scale, stride=1, pad=0, dil=1):
    Nb, Ho, Wo, Cout = dz.shape
    _, H, W, Cin = x.shape
    kh, kw = weight.shape[2:]
    g, Ho2, Wo2 = conv_geom_fwd(H, W, kh, kw, stride, pad, dil)
    assert (Ho2, Wo2) == (Ho, Wo)
    gw = grad_buffer(weight)
    assert gw.is_contiguous(memory_format=torch.channels_last) or (kh == 1 and kw == 1)
    wgrad_raw(dz, Cout, x, Cin, gw, kh * kw * Cin, Nb * Ho * Wo, Cout, Cin, taps=kh * kw, w_scale=scale, geom=g)


def maxpool3x3s2(x):
    Nb, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((Nb, Ho, Wo, Cc), device=x.device, dtype=torch.float32)
    check(lib().cdetr_maxpool3x3s2(ptr(x), ptr(y), Nb, H, W, Cc, stream_ptr()), "cdetr_maxpool3x3s2")
    return y


# ----------------------------------------------------------------------------------------------------- RCDA core
def rcda_pads(H, W):
    return (H + 7) & ~7, (W + 3) & ~3


class RcdaCoreFn(torch.autograd.Function):
    """Fused two-softmax + double contraction of A2/models/row_column_decoupled_attention.py:215-309.
    q_row,q_col [N,L,E]; k_row [N,W,E]; k_col [N,H,E]; v [N,H,W,E]; masks uint8 [N,W] / [N,H] or None -> out [N,L,E]."""

    @staticmethod
    def forward(ctx, q_row, q_col, k_row, k_col, v, mask_row, mask_col, nh):
        N, L, E = q_row.shape
        H, W = v.shape[1:3]
        assert E == nh * 32, "the RCDA kernels are specialised for head_dim 32"
        q_row, q_col, k_row, k_col, v = [t.contiguous() for t in (q_row, q_col, k_row, k_col, v)]
        Hp, Wp = rcda_pads(H, W)
        out = torch.empty((N, L, E), device=v.device, dtype=torch.float32)
        a_row = torch.empty((N, nh, L, Wp), device=v.device, dtype=torch.float32)
        a_col = torch.empty((N, nh, L, Hp), device=v.device, dtype=torch.float32)
        d = RcdaFwdDesc()
        d.N, d.L, d.H, d.W, d.nh, d.scale = N, L, H, W, nh, 32 ** -0.5
        d.precision = PRECISION
        d.q_row, d.q_col, d.k_row, d.k_col, d.v = ptr(q_row), ptr(q_col), ptr(k_row), ptr(k_col), ptr(v)
        d.mask_row, d.mask_col = ptr(mask_row), ptr(mask_col)
        d.out, d.a_row, d.a_col = ptr(out), ptr(a_row), ptr(a_col)
        with _Timed("rcda_fwd", 2.0 * N * nh * L * (H * W * 32 + (H + W) * 32)):
            check(lib().cdetr_rcda_fwd(C.byref(d), stream_ptr()), "cdetr_rcda_fwd")
        ctx.save_for_backward(q_row, q_col, k_row, k_col, v, a_row, a_col)
        ctx.nh = nh
        return out

    @staticmethod
    def backward(ctx, d_out):
        q_row, q_col, k_row, k_col, v, a_row, a_col = ctx.saved_tensors
        nh = ctx.nh
        N, L, E = q_row.shape
        H, W = v.shape[1:3]
        Hp, Wp = rcda_pads(H, W)
        d_out = d_out.contiguous()
        ds_row = torch.empty_like(a_row)
        ds_col = torch.empty_like(a_col)
        d_v = torch.zeros_like(v)
        d = RcdaBwdDesc()
        d.N, d.L, d.H, d.W, d.nh, d.scale = N, L, H, W, nh, 32 ** -0.5
        d.precision = PRECISION
        d.d_out, d.a_row, d.a_col, d.v = ptr(d_out), ptr(a_row), ptr(a_col), ptr(v)
        d.ds_row, d.ds_col, d.d_v = ptr(ds_row), ptr(ds_col), ptr(d_v)
        with _Timed("rcda_bwd", 2.0 * N * nh * L * (2 * H * W * 32)):
            check(lib().cdetr_rcda_bwd(C.byref(d), stream_ptr()), "cdetr_rcda_bwd")
        # logits -> projected q/k gradients: four small batched GEMMs per (n, head) on the same MFMA kernels
        dq_row = torch.empty_like(q_row)
        dq_col = torch.empty_like(q_col)
        dk_row = torch.zeros_like(k_row)
        dk_col = torch.zeros_like(k_col)
        for n in range(N):
            # dq[n, :, head*32:(head+1)*32] = dS[n, head] @ k[n, :, head*32:...]   (batch over heads)
            gemm_raw(ds_row[n], Wp, k_row[n], E, dq_row[n], E, L, 32, W, b_layout=1, batch=nh, sA=L * Wp, sB=32, sC=32)
            gemm_raw(ds_col[n], Hp, k_col[n], E, dq_col[n], E, L, 32, H, b_layout=1, batch=nh, sA=L * Hp, sB=32, sC=32)
            # dk[n, w, head*32+c] += sum_q dS[n, head, q, w] q[n, q, head*32+c]
            wgrad_raw(ds_row[n], Wp, q_row[n], E, dk_row[n], E, L, W, 32, batch=nh, sY=L * Wp, sX=32, sW=32)
            wgrad_raw(ds_col[n], Hp, q_col[n], E, dk_col[n], E, L, H, 32, batch=nh, sY=L * Hp, sX=32, sW=32)
        return dq_row, dq_col, dk_row, dk_col, d_v, None, None, None


def rcda_core(q_row, q_col, k_row, k_col, v, mask_row, mask_col, nh):
    return RcdaCoreFn.apply(q_row, q_col, k_row, k_col, v, mask_row, mask_col, nh)


# ----------------------------------------------------------------------------------------------------- matcher
class MatchPlan:
    """Host-side (static) description of one batch of targets: sizes and device offset tables."""

    def __init__(self, sizes, Q, device):
        self.sizes = [int(s) for s in sizes]
        self.Q = Q
        self.B = len(self.sizes)
        off = [0]
        for s in self.sizes:
            off.append(off[-1] + s)
        coff = [0]
        for s in self.sizes:
            coff.append(coff[-1] + Q * s)
        self.tgt_off = torch.tensor(off, dtype=torch.int32, device=device)
        self.cost_off = torch.tensor(coff[:-1], dtype=torch.int64, device=device)
        self.cost_numel = max(coff[-1], 1)
        self.cost_off_host = coff
        self.tgt_off_host = off
        self.M = [min(Q, s) for s in self.sizes]
        self.Mmax = max(max(self.M), 1)
        self.nc_max = max([Q] + self.sizes)
        self.sizes_f = torch.tensor([float(s) for s in self.sizes], dtype=torch.float32, device=device)


def match_cost(logits, boxes, tgt_boxes, plan, w_class=2.0, w_bbox=5.0, w_giou=2.0):
    cost = torch.empty(plan.cost_numel, device=logits.device, dtype=torch.float32)
    logits = logits.contiguous()
    boxes = boxes.contiguous()
    tgt_boxes = tgt_boxes.contiguous()
    check(lib().cdetr_match_cost(ptr(logits), logits.shape[-1], ptr(boxes), ptr(tgt_boxes) if tgt_boxes.numel() else ptr(cost),
                                 ptr(plan.tgt_off), ptr(plan.cost_off), plan.B, plan.Q, w_class, w_bbox, w_giou, ptr(cost),
                                 stream_ptr()), "cdetr_match_cost")
    return cost


def lsap(cost, plan):
    dev = cost.device
    idx_i = torch.zeros((plan.B, plan.Mmax), dtype=torch.int64, device=dev)
    idx_j = torch.zeros((plan.B, plan.Mmax), dtype=torch.int64, device=dev)
    status = torch.zeros(plan.B, dtype=torch.int32, device=dev)
    check(lib().cdetr_lsap(ptr(cost), ptr(plan.cost_off), ptr(plan.tgt_off), plan.B, plan.Q, plan.nc_max, plan.Mmax,
                           ptr(idx_i), ptr(idx_j), ptr(status), stream_ptr()), "cdetr_lsap")
    return idx_i, idx_j, status
