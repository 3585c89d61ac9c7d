"""Host-side operators over the C-ABI (include/cdetr_hip.h): thin launch wrappers + torch.autograd.Functions.

Layout conventions of the product path: activations are fp32 NHWC (`[N,H,W,C]` contiguous) or `[rows, C]`;
conv weights keep the reference's logical shape `[Cout,Cin,kh,kw]` (state-dict compatible) but live in
channels_last memory (`[Cout][kh][kw][Cin]` physically) so a filter tap is a contiguous K-run;
parameter gradients are ACCUMULATED in place into `param.grad` (the trainer's flat gradient arena) by the
weight-gradient kernels -- autograd only carries activation gradients.
"""
import os

import torch

from . import _ffi
from ._ffi import CriterionDesc, MirrorItem, ConvGeom, GemmDesc, RcdaBwdDesc, RcdaFwdDesc, WgradDesc, check, lib, ptr, stream_ptr

import ctypes as C


# Optional per-launch instrumentation (bench.py's roofline leg): a list collecting (family, algorithmic FLOPs,
# start event, end event) for every matrix-core launch; HIP events are recorded on the launch stream itself.
PROFILE = None

# Matrix-core arithmetic of the implicit-GEMM / weight-gradient kernels: 1 (default) = split-bf16 x3: every fp32 operand is
# split hi+lo and a product costs 3 bf16 MFMAs with fp32 accumulation (~5e-6 relative, ~5x less matrix-pipe time);
# 0 = fp32 MFMA (exact fp32 products).  Both modes pass the same parity suite (1e-3 rel, bit-exact Hungarian indices).
import os as _os
PRECISION = int(_os.environ.get("CDETR_PRECISION", "1"))
# Arithmetic of the BACKWARD contractions (data gradients and weight gradients) when PRECISION == 1: 1 = bf16x3 like the forward,
# 2 = "bf16x2" (the weight / activation operand rounded to bf16, the incoming gradient split hi+lo: 2 MFMAs per product),
# 3 = plain bf16 (1 MFMA).  The forward always runs bf16x3: its 1e-3 / bit-exact-assignment contract has no room for a bf16
# rounding per product (~1e-2 end to end), gradients have (per-parameter norms within 1e-2 of the reference).  DESIGN.md section 3.
PRECISION_BWD = int(_os.environ.get("CDETR_PRECISION_BWD", "3"))
# The decoder self-attention's backward in the backward's default arithmetic (scores recomputed in split-bf16, the four gradient contractions in
# plain bf16: cdetr_mha_bwd precision 3).  Measured -0.02 ms per step (21 -> 17 us per call) -- and OFF: the gradient of `adapt_pos2d.2.bias` reaches
# it only through this kernel's d_q / d_k, is ~3e-6 of the step's norm (a key-side bias cancels in the softmax) and its post-AdamW SIGNS are then
# decided by the bf16 noise: the full-size step's parameter-sum bar on it fails (3.5 sign flips of 256 where 2.5 are allowed).  profiles/r6_ab_mha_bwd.txt
MHA_BWD_BF16 = _os.environ.get("CDETR_MHA_BWD_BF16", "0") == "1"


def bwd_precision():
    return PRECISION_BWD if PRECISION == 1 else PRECISION


class arithmetic:
    """`with ops.arithmetic(precision, precision_bwd):` -- the arithmetic an ENGINE owns, in force for the calls it makes.  engine.Trainer and
    engine.InferenceEngine take theirs at construction and enter this scope around every forward / backward / capture they run, so two
    engines of different arithmetic (a split-bf16 trainer and an fp32-MFMA evaluator) live in one process without leaking into each other;
    the module-level PRECISION / PRECISION_BWD remain the defaults of code that runs outside any engine (kernel tests, tools).
    Scopes nest; a backward runs on the autograd engine's thread while the calling thread holds the scope (module globals: visible there)."""

    def __init__(self, precision, precision_bwd=None):
        self.want = (int(precision), int(PRECISION_BWD if precision_bwd is None else precision_bwd))

    def __enter__(self):
        global PRECISION, PRECISION_BWD
        self.saved = (PRECISION, PRECISION_BWD)
        PRECISION, PRECISION_BWD = self.want
        return self

    def __exit__(self, *exc):
        global PRECISION, PRECISION_BWD
        PRECISION, PRECISION_BWD = self.saved
        return False


class scope:
    """`with ops.scope(MIRROR=m, BRANCH_BESIDE=0, WGRAD_EVERY=0, AFTER_BACKBONE=fn, WG_DEFER_NESTED=True):` -- the launch-shaping switches an
    ENGINE owns while it captures or runs a piece of a step (which weight images the GEMMs read, whether shortcut convolutions / weight
    gradients fork to side streams, the release signal behind the backbone), set for the block and restored on exit whatever happens inside.
    The same contract as `arithmetic`: engines never leave a module-level switch changed behind their back; the module values stay the
    defaults of code that runs outside any engine.  Scopes nest."""
    _KEYS = ("MIRROR", "BRANCH_BESIDE", "WGRAD_EVERY", "AFTER_BACKBONE", "WG_DEFER_NESTED", "ZERO_ARENA", "RCDA_SAVE", "PROFILE")

    def __init__(self, **kw):
        assert all(k in self._KEYS for k in kw), kw
        self.want = kw

    def __enter__(self):
        g = globals()
        self.saved = {k: g[k] for k in self.want}
        g.update(self.want)
        return self

    def __exit__(self, *exc):
        globals().update(self.saved)
        return False


class ZeroArena:
    """The backward's zeroed accumulators (the RCDA key / value gradients of both stacks, the decoder's query-position sums, the projection's
    per-image weight gradient) as slices of ONE buffer an engine owns and zero-fills on its side stream, beside the forward, together with the
    gradient arena -- instead of three tensor-library fills on the backward's critical path (7-17 + 7 + 5 us, tools/step_listing.py).
    `buf is None`: a measuring pass -- requests are served by torch.zeros and their total recorded in `need`.  `reset()` opens a backward."""

    def __init__(self):
        self.buf, self.need, self.off = None, 0, 0

    def reset(self):
        self.off = 0

    def take(self, n, device):
        n_al = (n + 63) & ~63                      # 256-byte slices
        if self.buf is None:
            self.off += n_al
            self.need = max(self.need, self.off)
            return torch.zeros(n, device=device, dtype=torch.float32)
        if self.off + n_al > self.buf.numel():     # (a backward that asks for more than the measured one did: never silently alias)
            raise RuntimeError(f"ZeroArena: {self.off + n_al} elements requested, {self.buf.numel()} measured")
        t = self.buf[self.off:self.off + n]
        self.off += n_al
        return t


ZERO_ARENA = None


def zeros_flat(n, device):
    """n zeroed fp32 elements: a slice of the engine's pre-zeroed arena (ops.scope(ZERO_ARENA=...)) or a fresh torch.zeros."""
    return ZERO_ARENA.take(n, device) if ZERO_ARENA is not None else torch.zeros(n, device=device, dtype=torch.float32)


class _Timed:
    def __init__(self, family, flops, tag=None, nbytes=0.0, issued=None):
        self.family, self.flops, self.tag, self.nbytes = family, flops, tag, nbytes     # nbytes: compulsory (algorithmic) HBM bytes of the launch
        self.issued = flops if issued is None else issued     # matrix-core FLOPs actually issued (algorithmic x MFMAs per product)

    def __enter__(self):
        if PROFILE is not None:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
            PROFILE.append((self.family, self.flops, self.t0, t1, self.tag, self.nbytes, self.issued))
        return False


_TERMS = {0: 1, 1: 3, 2: 2, 3: 1}      # bf16 MFMAs per algorithmic product of each arithmetic mode (0 = fp32 MFMA, priced on its own peak)


def _geom(mode=_ffi.ROWS_DENSE, Ha=0, Wa=0, Hc=0, Wc=0, kh=1, kw=1, stride=1, pad=0, dil=1):
    return ConvGeom(mode, Ha, Wa, Hc, Wc, kh, kw, stride, pad, dil)


# Scratch of the split-reduction GEMMs (cdetr_gemm_desc.splitk_ws): one per stream -- launches of one stream are ordered, launches of
# two streams may overlap and must not share the arrival counters.  Zeroed once; every launch leaves the counters zero.
SPLITK = int(_os.environ.get("CDETR_SPLITK", "1"))
SPLITK_BYTES = 64 << 20
_SPLITK_WS = {}


def splitk_ws():
    s = stream_ptr().value
    t = _SPLITK_WS.get(s)
    if t is None:
        t = _SPLITK_WS[s] = torch.empty(SPLITK_BYTES // 4, dtype=torch.int32, device="cuda")
        t[:4096].zero_()        # only the arrival counters need a defined start (a scratch first met inside a graph capture re-runs this fill on replay)
    return t


def gemm_raw(A, lda, B, ldb, Cout, ldc, M, N, K, taps=1, b_layout=0, bias=None, w_scale=None, resid=None, ldr=0,
             gate=None, ldg=0, relu=False, out_scale=1.0, geom=None, batch=1, sA=0, sB=0, sC=0, B_split=None,
             batch_inner=0, sA2=0, sB2=0, sC2=0, precision=None, C16=None, A16=None, A16lo=None, C16lo=None, dl=None, probe_ws=None, B16=None, gate16=None,
             A_split=None, C_split=None, prio=False, resid_groups=False, gate16_only=False):
    """dl = (tile, stages): run the direct-to-LDS tile kernel with that configuration (cdetr_gemm_dl; tests / sweeps)."""
    d = GemmDesc()
    d.C16, d.A16, d.A16lo, d.C16lo, d.B16, d.gate16 = ptr(C16), ptr(A16), ptr(A16lo), ptr(C16lo), ptr(B16), ptr(gate16)
    if A_split is not None:        # interleaved groups [hi 32 | lo 32] travel in the A16 / C16lo fields under a flag (include/cdetr_hip.h)
        d.A16, d.flags = ptr(A_split), d.flags | 1
    if C_split is not None:
        d.C16lo, d.flags = ptr(C_split), d.flags | 2
    if prio:
        d.flags |= 4                # CDETR_GEMM_PRIO
    if resid_groups:
        d.flags |= 8                # CDETR_GEMM_RESID_GROUPS: `resid` is an interleaved-groups tensor
    if gate16_only:
        d.flags |= 16               # CDETR_GEMM_GATE16_ONLY: `gate` is no fp32 tensor, only its twin may be read
    d.batch_inner, d.sA2, d.sB2, d.sC2 = batch_inner, sA2, sB2, sC2
    d.M, d.N, d.K, d.taps, d.batch, d.b_layout, d.relu, d.out_scale = M, N, K, taps, batch, b_layout, int(relu), out_scale
    d.precision = PRECISION if precision is None else precision
    d.A, d.lda, d.sA = ptr(A), lda, sA
    d.B, d.ldb, d.sB = ptr(B), ldb, sB
    d.B_split = ptr(B_split)
    d.C, d.ldc, d.sC = ptr(Cout), ldc, sC
    d.w_scale, d.bias = ptr(w_scale), ptr(bias)
    d.resid, d.ldr = ptr(resid), ldr
    d.gate, d.ldg = ptr(gate), ldg
    d.g = geom if geom is not None else _geom()
    if probe_ws is not None:
        d.splitk_ws, d.splitk_ws_bytes = ptr(probe_ws), probe_ws.numel() * probe_ws.element_size()
    elif SPLITK and batch == 1 and _GEMM_QUEUE is None and M * N <= (1 << 22) and K * taps >= 256:
        ws = splitk_ws()
        d.splitk_ws, d.splitk_ws_bytes = ptr(ws), SPLITK_BYTES
    if _GEMM_QUEUE is not None:       # inside gemm_queue(): submitted together by its exit (cdetr_gemm_group)
        _GEMM_QUEUE.append((d, 2.0 * M * N * K * taps * batch, 4.0 * (M * K + N * K * taps + M * N) * max(batch, 1),
                            (A, B, Cout, bias, w_scale, resid, gate, B_split, C16, A16, A16lo, C16lo, B16, gate16, A_split, C_split), _TERMS[d.precision]))
        return
    # compulsory fp32 bytes: input rows once (a strided / dilated conv reads <= M*K of them), weights, output
    fl = 2.0 * M * N * K * taps * batch
    with _Timed(_gemm_family(M, N, batch, geom), fl, (M, N, K, taps, b_layout, batch), 4.0 * (M * K + N * K * taps + M * N) * max(batch, 1),
                fl * _TERMS[d.precision]):
        if dl is not None:
            check(lib().cdetr_gemm_dl(C.byref(d), int(dl[0]), int(dl[1]), stream_ptr()), "cdetr_gemm_dl")
        else:
            check(lib().cdetr_gemm(C.byref(d), stream_ptr()), "cdetr_gemm")


_GEMM_QUEUE = None


def _gemm_family(M, N, batch, geom):
    """Profile family of a GEMM launch: the tile kernels (`igemm`: igemm_fast_kernel and its generic sibling) or the few-row class
    (`igemm_fewrow`: dense problems of <= 48 tiles of 64x64 -- the decoder's 600-row GEMMs, positional MLPs, heads; csrc/igemm.hip
    gemm_is_direct) whose time is launch latency, not matrix work."""
    dense = geom is None or geom.mode == _ffi.ROWS_DENSE
    return "igemm_fewrow" if dense and ((M + 63) // 64) * ((N + 63) // 64) * max(batch, 1) <= 48 else "igemm"


class gemm_queue:
    """Context manager: every GEMM issued inside (linear_fwd / linear_dgrad / gemm_raw) is queued and the whole set is submitted as
    ONE cdetr_gemm_group call on exit (grouped launches per kernel class).  The caller guarantees that the queued problems are
    independent -- none reads another's output -- and that nothing inside the block reads an output tensor: they are filled at
    exit.  Operand tensors are kept alive until then."""

    def __enter__(self):
        global _GEMM_QUEUE
        assert _GEMM_QUEUE is None, "gemm_queue does not nest"
        _GEMM_QUEUE = []
        return self

    def __exit__(self, et, ev, tb):
        global _GEMM_QUEUE
        q, _GEMM_QUEUE = _GEMM_QUEUE, None
        if et is None and q:
            arr = (GemmDesc * len(q))(*[e[0] for e in q])
            d0 = q[0][0]
            with _Timed(_gemm_family(d0.M, d0.N, d0.batch, d0.g), sum(e[1] for e in q), ("group", sum(e[2] for e in q), len(q), 0, -1, 0),
                        sum(e[2] for e in q), sum(e[1] * e[4] for e in q)):
                check(lib().cdetr_gemm_group(arr, len(q), stream_ptr()), "cdetr_gemm_group")
        return False


def wgrad_raw(dY, ldy, X, ldx, dW, ldw, P, Nout, Cin, taps=1, w_scale=None, geom=None, batch=1, sY=0, sX=0, sW=0,
              dbias=None, batch_inner=0, sY2=0, sX2=0, sW2=0, may_defer=False, dY16=None, X16=None, precision=None, wg_target=0):
    """dY16 / X16: optional bf16 twins of dY / X (same shape and strides in elements): the plain-bf16 kernel reads them instead.
    wg_target: cdetr_wgrad_desc.wg_target (0 = the library's default; a launch that has the chip to itself asks for more workgroups)."""
    d = WgradDesc()
    d.wg_target = wg_target
    d.dY16, d.X16 = ptr(dY16), ptr(X16)
    d.batch_inner, d.sY2, d.sX2, d.sW2 = batch_inner, sY2, sX2, sW2
    d.P, d.Nout, d.Cin, d.taps, d.batch = P, Nout, Cin, taps, batch
    d.precision = bwd_precision() if precision is None else precision
    d.dY, d.ldy, d.sY = ptr(dY), ldy, sY
    d.X, d.ldx, d.sX = ptr(X), ldx, sX
    d.dW, d.ldw, d.sW = ptr(dW), ldw, sW
    d.w_scale = ptr(w_scale)
    d.dbias = ptr(dbias)
    d.g = geom if geom is not None else _geom()
    if may_defer and _WG_QUEUE is not None:       # inside wgrad_queue(): submitted together by its exit (cdetr_wgrad_group)
        _WG_QUEUE.append((d, 2.0 * P * Nout * Cin * taps * batch, (dY, X, dW, w_scale, dbias, dY16, X16), 4.0 * (P * Nout + P * Cin + 2 * Nout * Cin * taps) * max(batch, 1),
                          _TERMS[d.precision]))
        return
    # compulsory bytes: dY and X once, dW read + written (accumulation into the gradient arena)
    fl = 2.0 * P * Nout * Cin * taps * batch
    with _Timed("wgrad", fl, (P, Nout, Cin, taps, -1, batch), 4.0 * (P * Nout + P * Cin + 2 * Nout * Cin * taps) * max(batch, 1), fl * _TERMS[d.precision]):
        check(lib().cdetr_wgrad(C.byref(d), stream_ptr()), "cdetr_wgrad")


_WG_QUEUE = None


WG_DEFER_NESTED = False      # a nested wgrad_queue() hands its problems to the enclosing one instead of submitting them (see below)


class wgrad_queue:
    """Context manager: PARAMETER-gradient calls (`_wg`) issued inside are queued and submitted as ONE cdetr_wgrad_group call on
    exit (grouped launches).  Legal because a layer's parameter gradients are independent of each other and nothing inside a
    backward reads the gradient buffers -- only the optimizer / gradient exchange do, after the block.  The operand tensors
    are kept alive until the flush.  Weight-gradient-shaped contractions whose result the backward itself consumes (the RCDA
    key gradients) are never deferred.
    With `WG_DEFER_NESTED` set, a queue that closes INSIDE another one passes its problems up instead of submitting them: the trainer's
    chain layout collects every parameter gradient of the backward above the backbone in its own outer queue and submits them as a
    separate graph on the weight-gradient stream (engine.Trainer._capture_chain)."""

    def __enter__(self):
        global _WG_QUEUE
        self.prev = _WG_QUEUE
        _WG_QUEUE = []
        return self

    def __exit__(self, et, ev, tb):
        global _WG_QUEUE
        if et is None:
            if WG_DEFER_NESTED and self.prev is not None:
                self.prev.extend(_WG_QUEUE)
                del _WG_QUEUE[:]
            else:
                wgrad_flush()
        _WG_QUEUE = self.prev
        return False


# Parameter gradients OFF the critical path: the backward's data-gradient chain is a string of dependent launches that leave most CUs
# idle at two images per GPU; the weight gradients of a finished layer depend on nothing downstream, so their grouped launch goes to a
# side stream (fork after the layer, join before the optimizer / the bucket's all-reduce) and fills those CUs while the chain runs on.
# The operand tensors of launches in flight are kept referenced until the join: the allocator must not hand their memory to the chain.
ENC_DEFER = int(_os.environ.get("CDETR_ENC_DEFER", "1"))      # encoder backward: a layer's d(src) parts are summed by the LayerNorm backward below it
WGRAD_ASYNC = int(_os.environ.get("CDETR_WGRAD_ASYNC", "1"))
# Measured (one MI355X, B=2 800x800, same box, ms/step): everything at the end 10.95 | one overlapped submission per backbone segment
# (layer4 / layer3 / layer2) 10.87 | every 3 blocks 10.95 | every block 11.02 (small groups lose the grouped launch) | encoder / decoder
# layers overlapped as well 11.13-11.25 (their chains are latency-bound: a concurrent kernel slows every link).
# Round 2 left this off (0.07 ms then).  With the round-3 kernels (twin-fed weight gradients, direct-to-LDS data gradients) one overlapped
# submission per backbone segment is worth 0.10-0.20 ms (same-lease A/B: 9.79 -> 9.69 and 9.70 -> 9.50 ms; every 2 blocks 9.73; encoder /
# decoder stacks as well 9.81): ON.  The kernels that run beside each other stretch (tile GEMM family 4.66 -> 4.97 ms of summed launch
# time, weight gradients 1.43 -> 1.55), so bench.py reports the family figures as they run AND with the overlap off (roofline.unoverlapped).
WGRAD_EVERY = int(_os.environ.get("CDETR_WGRAD_EVERY", "100"))     # backbone: blocks per overlapped submission (100: one per segment; 0 = at the end)
_WG_SIDE = {}
_WG_INFLIGHT = []


def wgrad_side_stream():
    dev = torch.cuda.current_device()
    s = _WG_SIDE.get(dev)
    if s is None:
        s = _WG_SIDE[dev] = torch.cuda.Stream()
    return s


WGRAD_BESIDE_TARGET = int(_os.environ.get("CDETR_WGRAD_BESIDE_TARGET", "384"))


def wgrad_flush(overlap=False, wg_target=None):
    """Submit whatever the innermost wgrad_queue() holds (the gradient exchange calls this before it ships a bucket).
    overlap: on the side stream, ordered after everything issued so far on the current one; wgrad_join() is the matching join.
    wg_target: cdetr_wgrad_desc.wg_target for the queued problems that carry none -- launches that run BESIDE the data-gradient chain
    (overlap=True, or a side-stream graph of the chain layout) ask for WGRAD_BESIDE_TARGET = 384 workgroups: fewer, longer workgroups
    disturb the chain less (profiles/r5_ab_wgrad.txt); the library's own default (768) is for a launch that has the chip to itself."""
    q = _WG_QUEUE
    if q:
        if wg_target is None and overlap and WGRAD_ASYNC:
            wg_target = WGRAD_BESIDE_TARGET
        if wg_target:
            for e in q:
                if e[0].wg_target == 0:
                    e[0].wg_target = int(wg_target)
        arr = (WgradDesc * len(q))(*[e[0] for e in q])
        if overlap and WGRAD_ASYNC:
            side = wgrad_side_stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with _Timed("wgrad", sum(e[1] for e in q), (-len(q), 0, 0, 0, -1, 0), sum(e[3] for e in q), sum(e[1] * e[4] for e in q)):
                    check(lib().cdetr_wgrad_group(arr, len(q), stream_ptr()), "cdetr_wgrad_group")
            _WG_INFLIGHT.append([e[2] for e in q])
        else:
            with _Timed("wgrad", sum(e[1] for e in q), (-len(q), 0, 0, 0, -1, 0), sum(e[3] for e in q), sum(e[1] * e[4] for e in q)):
                check(lib().cdetr_wgrad_group(arr, len(q), stream_ptr()), "cdetr_wgrad_group")
        del q[:]


def wgrad_join(stream=None):
    """Make `stream` (default: the current one) wait for the parameter-gradient launches that were sent to the side stream."""
    if _WG_INFLIGHT:
        (stream or torch.cuda.current_stream()).wait_stream(wgrad_side_stream())
        if stream is None:
            del _WG_INFLIGHT[:]


# A bottleneck's shortcut convolution (and its data gradient) depends on nothing in the block's main branch until the residual add: it runs
# on a side stream beside conv1 -> conv2 (a parallel branch of the captured graph).  The output buffer is allocated by the caller on ITS
# stream before the fork, so no tensor changes allocator streams.
BRANCH_BESIDE = int(_os.environ.get("CDETR_BRANCH_BESIDE", "1"))
_BR_SIDE = {}


class fork_branch:
    """with fork_branch() as br: <launches on the side stream, ordered after everything issued so far> ... br.join() before the first consumer."""

    def __enter__(self):
        main = torch.cuda.current_stream()
        key = (main.device, main.cuda_stream)     # one side stream per parent stream: its split-reduction scratch (splitk_ws) is not shared
        side = _BR_SIDE.get(key)                  # with a step that runs concurrently on another stream (the frozen-stage prefetch)
        if side is None:
            side = _BR_SIDE[key] = torch.cuda.Stream(device=main.device)
        side.wait_stream(main)
        self.side, self.ctx = side, torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self.event = self.side.record_event()
        return self.ctx.__exit__(*exc)

    def join(self):
        torch.cuda.current_stream().wait_event(self.event)


def colsum_(X2d, out):
    """out[n] += sum_m X2d[m][n]"""
    M, N = X2d.shape
    check(lib().cdetr_colsum(ptr(X2d), X2d.stride(0), M, N, ptr(out), stream_ptr()), "cdetr_colsum")


def relu_mask(y, dy, scale=1.0, twin=False):
    """dz = (y > 0) ? dy * scale : 0 in one pass; twin: -> (dz, bf16 copy of dz) from the same pass."""
    dy = dy.contiguous()
    dz = torch.empty_like(dy)
    if twin:
        dz16 = torch.empty(dy.shape, device=dy.device, dtype=torch.bfloat16)
        check(lib().cdetr_relu_mask2(ptr(y), ptr(dy), ptr(dz), ptr(dz16), dy.numel(), scale, stream_ptr()), "cdetr_relu_mask2")
        return dz, dz16
    check(lib().cdetr_relu_mask(ptr(y), ptr(dy), ptr(dz), dy.numel(), scale, stream_ptr()), "cdetr_relu_mask")
    return dz


def split_groups(x):
    """fp32 [..., K] (K % 32 == 0) -> its interleaved split-bf16 form [..., K/32, 64] bf16: per group of 32 values [hi 32 | lo 32]
    (cdetr_gemm_desc.A_split / B_split; tensor ops: tests and sweeps -- the product's producers write it from their epilogues)."""
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    sh = x.shape[:-1] + (x.shape[-1] // 32, 32)
    return torch.cat([hi.view(sh), lo.view(sh)], dim=-1).contiguous()


class Groups:
    """An NHWC activation kept ONCE as interleaved split-bf16 groups -- per 32 channels [hi 32 | lo 32] bf16, 4 bytes per element, the byte layout
    of cdetr_gemm_desc.B_split -- instead of an fp32 tensor: the direct-to-LDS tile kernel reads it as its A operand in full 128-byte lines
    (CDETR_GEMM_A_GROUPS), the epilogue that produces it writes it in place of the fp32 output (CDETR_GEMM_C_GROUPS), a later residual add reads
    hi + lo (CDETR_GEMM_RESID_GROUPS).  hi + lo carries 16 mantissa bits of the fp32 value (the split-bf16 x3 product reads no more of it).
    `.t` = the bf16 buffer [N, H, W, C / 32, 64]; `.shape` = the logical (N, H, W, C)."""
    __slots__ = ("t", "shape")

    def __init__(self, t, shape):
        self.t, self.shape = t, tuple(shape)

    @classmethod
    def empty(cls, shape, device):
        assert shape[-1] % 32 == 0
        return cls(torch.empty(tuple(shape[:-1]) + (shape[-1] // 32, 64), device=device, dtype=torch.bfloat16), shape)

    @classmethod
    def of(cls, x):
        return cls(split_groups(x), x.shape)

    @property
    def device(self):
        return self.t.device

    @property
    def is_cuda(self):
        return self.t.is_cuda

    def float(self):
        """fp32 value hi + lo (tests, fallbacks)."""
        g = self.t.float()
        return (g[..., :32] + g[..., 32:]).reshape(self.shape)


# Bottleneck outputs inside a backbone stage as interleaved groups.  Built and parity-green in round 5 (g10 goldens, timed-path tests, the chain test in
# tests/test_gemm_dl.py), and NEUTRAL in the step: 8.88-8.91 ms against 8.91-8.93 ms without (two same-lease pairs; under the tracer +0.05 ms) --
# the cold-operand sweep's 1.12-1.29x on the 1x1 shapes does not appear when the operand was just written (profiles/r5_ab_groups.txt).  Off by default.
GROUPS = os.environ.get("CDETR_GROUPS", "0") != "0"
GROUPS_MIN_ROWS = 4096      # every consumer must run on the direct-to-LDS tile kernel: pixel rows from which cdetr_gemm's rules send it there


def use_groups(rows, channels):
    """Whether a bottleneck output of `rows` pixels x `channels` may leave its epilogue as interleaved groups only (split-bf16 forward with the
    pre-split weight images at hand; big enough that the next 1x1 convolutions are direct-to-LDS problems)."""
    return GROUPS and PRECISION == 1 and MIRROR is not None and rows >= GROUPS_MIN_ROWS and channels % 64 == 0


def split_planes(x):
    """fp32 tensor -> (hi, lo) bf16 planes of the split-bf16 form: hi = bf16(x), lo = bf16(x - hi) -- what the producing epilogues
    write as C16 / C16lo (tests and tools; the product path never converts with tensor ops)."""
    hi = x.to(torch.bfloat16)
    return hi, (x - hi.float()).to(torch.bfloat16)


def bf16_twins():
    """Whether the producers of the backbone's backward should write bf16 twins of their outputs: the plain-bf16 weight gradients read
    them instead of the fp32 tensors (half the operand bytes, no conversion at staging; `wgrad_tr16_kernel`)."""
    return TWINS and bwd_precision() == 3


def grad_buffer(p):
    """The in-place gradient accumulator of a parameter (the trainer pre-binds views of its flat arena)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


# ----------------------------------------------------------------------------------------------------- weight mirrors
class WeightMirror:
    """Device-side images of the weights, rewritten by ONE launch per step (cdetr_weight_mirror):
      * `entries` = [(weight, scale or None)] used as data-gradient operands: the k-contiguous transpose
        Wt[c][tap][o] = W[o][tap][c] * scale[o] in fp32 and (when R % 32 == 0) pre-split into bf16 hi / lo;
      * `fwd_entries` = [(weight, scale or None)] used as forward operands: W * scale pre-split (needs C % 32 == 0).
    Weights are [R, C] or channels_last [R, C, kh, kw].  The pre-split images let the bf16x3 GEMM kernels stage the weight
    operand with a plain copy (cdetr_gemm_desc.B_split); `lookup*` also resolve row slices of registered 2-D weights."""

    def __init__(self, entries, fwd_entries=()):
        import bisect
        import numpy as np
        self._bisect = bisect
        dev = (entries or fwd_entries)[0][0].device
        tot_t = sum(w.numel() for w, _ in entries)
        tot_f = sum(w.numel() for w, _ in fwd_entries)
        self.flat = torch.zeros(max(tot_t, 1), device=dev, dtype=torch.float32)          # fp32 transposes
        self.flat_ts = torch.zeros(max(tot_t, 1), device=dev, dtype=torch.float32)       # their bf16 hi/lo images (same byte size)
        self.flat_t16 = torch.zeros(max(tot_t, 1), device=dev, dtype=torch.bfloat16)     # their plain-bf16 images (cdetr_gemm_desc.B16)
        self.flat_fs = torch.zeros(max(tot_f, 1), device=dev, dtype=torch.float32)       # forward bf16 hi/lo images
        self._keep = []
        items = (MirrorItem * (len(entries) + len(fwd_entries)))()
        self.t_entries, self.f_entries = [], []
        tile0 = 0
        self.nT = len(entries)

        def geom(w):
            R, Cc = w.shape[0], w.shape[1]
            taps = w.shape[2] * w.shape[3] if w.dim() == 4 else 1
            if w.dim() == 4:
                assert taps == 1 or w.is_contiguous(memory_format=torch.channels_last)
            else:
                assert w.is_contiguous()
            return R, Cc, taps

        off = 0
        for i, (w, sc) in enumerate(entries):
            R, Cc, taps = geom(w)
            it = items[i]
            it.src, it.dst = w.data_ptr(), self.flat.data_ptr() + 4 * off
            has_split = R % 32 == 0
            it.dst_split = (self.flat_ts.data_ptr() + 4 * off) if has_split else None
            it.dst_hi = self.flat_t16.data_ptr() + 2 * off
            it.scale = sc.data_ptr() if sc is not None else None
            it.R, it.C, it.taps, it.tile0, it.transpose = R, Cc, taps, tile0, 1
            self.t_entries.append((w.data_ptr(), w.numel() * 4, off, R, Cc, taps, sc.data_ptr() if sc is not None else 0, has_split))
            self._keep.append((w, sc))
            tile0 += ((R + 31) // 32) * ((Cc + 31) // 32) * taps
            off += w.numel()
        self.tilesT = tile0
        tile0 = 0            # the forward images are a table (and a launch) of their own: see refresh()
        off = 0
        for j, (w, sc) in enumerate(fwd_entries):
            R, Cc, taps = geom(w)
            assert Cc % 32 == 0
            it = items[len(entries) + j]
            it.src, it.dst, it.dst_split = w.data_ptr(), None, self.flat_fs.data_ptr() + 4 * off
            it.scale = sc.data_ptr() if sc is not None else None
            it.R, it.C, it.taps, it.tile0, it.transpose = R, Cc, taps, tile0, 0
            self.f_entries.append((w.data_ptr(), w.numel() * 4, off, R, Cc, taps, sc.data_ptr() if sc is not None else 0, True))
            self._keep.append((w, sc))
            tile0 += (w.numel() + 4095) // 4096          # cdetr_weight_images: blocks of 4096 weights
            off += w.numel()
        self.tilesF, self.nF = tile0, len(fwd_entries)
        raw = np.frombuffer(bytes(items), dtype=np.uint8).copy()
        self.items_dev = torch.from_numpy(raw).to(dev)
        self.item_bytes = C.sizeof(MirrorItem)
        self.t_entries.sort()
        self.f_entries.sort()
        self._tb = [e[0] for e in self.t_entries]
        self._fb = [e[0] for e in self.f_entries]
        self._memo_t, self._memo_f = {}, {}

    def refresh(self, part="all"):
        """Rewrite the images from the current weights.  part "fwd": the forward operands (needed before the forward pass);
        "bwd": the transposed data-gradient operands (needed only when the backward starts -- the trainer rewrites them on a side
        stream under the Hungarian solve); "all": both."""
        if part in ("all", "fwd") and self.nF:
            check(lib().cdetr_weight_images(self.items_dev.data_ptr() + self.nT * self.item_bytes, self.nF, self.tilesF, stream_ptr()),
                  "cdetr_weight_images")
        if part in ("all", "bwd") and self.nT:
            check(lib().cdetr_weight_mirror(self.items_dev.data_ptr(), self.nT, self.tilesT, stream_ptr()), "cdetr_weight_mirror")

    def _find(self, table, bases, w, scale):
        p = w.data_ptr()
        i = self._bisect.bisect_right(bases, p) - 1
        if i < 0:
            return None
        e = table[i]
        if p >= e[0] + e[1] or e[6] != (scale.data_ptr() if scale is not None else 0):
            return None
        row0 = (p - e[0]) // (4 * e[4] * e[5])
        if (p - e[0]) != row0 * 4 * e[4] * e[5] or (row0 and e[5] != 1):
            return None
        return e, row0

    def lookup(self, w, scale=None):
        """data-gradient operand of `w` (or of a row slice of it) -> (fp32 mirror view, ldb, pre-split view or None, bf16 view or None) or None"""
        key = (w.data_ptr(), scale.data_ptr() if scale is not None else 0)
        hit = self._memo_t.get(key, False)
        if hit is not False:
            return hit
        r = self._find(self.t_entries, self._tb, w, scale)
        if r is None:
            out = None
        else:
            (base, nbytes, off, R, Cc, taps, sptr, has_split), row0 = r
            sp = self.flat_ts[off + row0:] if (has_split and row0 % 32 == 0) else None
            out = (self.flat[off + row0:], taps * R, sp, self.flat_t16[off + row0:] if row0 % 8 == 0 else None)
        self._memo_t[key] = out        # the images never move: the answer for a given operand address is fixed
        return out

    def lookup_fwd(self, w, scale=None):
        """pre-split forward operand of `w` (or of a row slice of it) or None"""
        key = (w.data_ptr(), scale.data_ptr() if scale is not None else 0)
        hit = self._memo_f.get(key, False)
        if hit is not False:
            return hit
        r = self._find(self.f_entries, self._fb, w, scale)
        if r is None:
            out = None
        else:
            (base, nbytes, off, R, Cc, taps, sptr, _), row0 = r
            out = self.flat_fs[off + row0 * Cc * taps:]
        self._memo_f[key] = out
        return out


AFTER_BACKBONE = None      # optional callable: AnchorDETR.forward calls it right after the backbone (engine.InferenceEngine's release signal)
MIRROR = None      # set by engine.Trainer; None -> data gradients read the weight itself as the n-contiguous operand


# ----------------------------------------------------------------------------------------------------- linear
def linear_fwd(x2d, weight, bias=None, relu=False, resid=None, out_scale=1.0, out=None):
    M, K = x2d.shape
    N = weight.shape[0]
    y = out if out is not None else torch.empty((M, N), device=x2d.device, dtype=torch.float32)
    sp = MIRROR.lookup_fwd(weight) if MIRROR is not None else None
    gemm_raw(x2d, x2d.stride(0), weight, weight.stride(0), y, y.stride(0), M, N, K, bias=bias, relu=relu,
             resid=resid, ldr=(resid.stride(0) if resid is not None else 0), out_scale=out_scale, B_split=sp)
    return y


def linear_dgrad(dy2d, weight, gate=None, resid=None, dy16=None, twin=False):
    """dx = dy . W   (W [N_out][K_in] read as the n-contiguous operand).  dy16: bf16 twin of dy (plain-bf16 backward: the direct-to-LDS
    kernel reads it instead of dy); twin: -> (dx, bf16 twin of dx from the same epilogue)."""
    M, N = dy2d.shape
    K = weight.shape[1]
    dx = torch.empty((M, K), device=dy2d.device, dtype=torch.float32)
    m = MIRROR.lookup(weight) if MIRROR is not None else None
    if m is not None:
        dx16 = torch.empty((M, K), device=dy2d.device, dtype=torch.bfloat16) if twin else None
        gemm_raw(dy2d, dy2d.stride(0), m[0], m[1], dx, K, M, K, N, b_layout=0,
                 gate=gate, ldg=(gate.stride(0) if gate is not None else 0),
                 resid=resid, ldr=(resid.stride(0) if resid is not None else 0), B_split=m[2], B16=m[3], precision=bwd_precision(),
                 A16=dy16, C16=dx16)
        return (dx, dx16) if twin else dx
    gemm_raw(dy2d, dy2d.stride(0), weight, weight.stride(0), dx, K, M, K, N, b_layout=1,
             gate=gate, ldg=(gate.stride(0) if gate is not None else 0),
             resid=resid, ldr=(resid.stride(0) if resid is not None else 0), precision=bwd_precision())
    return (dx, None) if twin else dx


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
                      dbias=gb, may_defer=True)       # parameter gradient: queued when the trainer runs backward under wgrad_queue()
        return dx, None, None, None, None, None, d_resid, None


class PosMlpFn(torch.autograd.Function):
    """Linear -> ReLU -> Linear (adapt_pos1d / adapt_pos2d, A2/models/transformer.py:73-74) applied to SEVERAL inputs at once, as one
    autograd node: forward = two grouped launches (first layers, second layers), backward = one grouped launch for the data gradients through
    the second layer with the ReLU mask as the epilogue's gate, one for the first layer of the inputs that need a gradient (the sine
    embeddings of the anchor points; mask positions need none) -- instead of (data gradient, mask pass, data gradient) per input.
    args = (w0, b0, w2, b2, *xs) -> tuple of outputs."""

    @staticmethod
    def forward(ctx, w0, b0, w2, b2, *xs):
        W0, B0, W2, B2 = w0.detach(), b0.detach(), w2.detach(), b2.detach()
        x2 = []
        for x in xs:
            t = x.reshape(-1, x.shape[-1])
            if t.stride(-1) != 1 or (t.stride(0) & 3) or (t.data_ptr() & 15):
                t = t.contiguous()
            x2.append(t)
        with gemm_queue():
            hs = [linear_fwd(t, W0, B0, relu=True) for t in x2]
        with gemm_queue():
            ys = [linear_fwd(h, W2, B2) for h in hs]
        ctx.params = (w0, b0, w2, b2)
        ctx.n = len(xs)
        ctx.shapes = [x.shape for x in xs]
        ctx.save_for_backward(*x2, *hs)
        return tuple(y.reshape(*x.shape[:-1], y.shape[-1]) for y, x in zip(ys, xs))

    @staticmethod
    def backward(ctx, *dys):
        n = ctx.n
        saved = ctx.saved_tensors
        x2, hs = saved[:n], saved[n:]
        w0, b0, w2, b2 = ctx.params
        W0, W2 = w0.detach(), w2.detach()
        live = [i for i in range(n) if dys[i] is not None]
        d2 = {i: dys[i].reshape(-1, dys[i].shape[-1]).contiguous() for i in live}
        with gemm_queue():
            dh = {i: linear_dgrad(d2[i], W2, gate=hs[i]) for i in live}          # masked by relu(h): the gradient w.r.t. the first layer's pre-activation
        need = [i for i in live if ctx.needs_input_grad[4 + i]]
        dx = {}
        if need:
            with gemm_queue():
                dx = {i: linear_dgrad(dh[i], W0) for i in need}
        for i in live:
            for (dy, xin, w, b) in ((d2[i], hs[i], w2, b2), (dh[i], x2[i], w0, b0)):
                if w.requires_grad:
                    gw = grad_buffer(w)
                    gb = grad_buffer(b) if b.requires_grad else None
                    wgrad_raw(dy, dy.stride(0), xin, xin.stride(0), gw, gw.stride(0), dy.shape[0], gw.shape[0], xin.shape[1], dbias=gb, may_defer=True)
        return (None, None, None, None) + tuple(dx[i].reshape(ctx.shapes[i]) if i in dx else None for i in range(n))


FUSED_HEADS = os.environ.get("CDETR_FUSED_HEADS", "1") != "0"      # the three heads as one autograd node (A/B)


class HeadsFn(torch.autograd.Function):
    """The three heads on one decoder state (A2/models/transformer.py:79-107,193-213): class Linear(256, ncls) and the two 3-layer MLPs
    (boxes: 4 outputs, variances: 2) as ONE autograd node with a hand-scheduled backward.
    Forward: three grouped launches (the layers of one depth run together).  Backward: depth by depth, the two MLPs' data gradients of a
    depth in one grouped launch with the ReLU mask of the layer below as the epilogue's gate (no mask pass), the three input gradients
    chained through residual epilogues and ONE add -- 6 launches where the per-layer autograd nodes took 15; parameter gradients are
    queued (ops.wgrad_queue) like every other one of the step.
    params = (cls_w, cls_b, [w, b] x 3 of the box MLP, [w, b] x 3 of the variance MLP)."""

    @staticmethod
    def forward(ctx, x, *params):
        shp = x.shape
        x2d = x.reshape(-1, shp[-1])
        if x2d.stride(-1) != 1 or (x2d.stride(0) & 3) or (x2d.data_ptr() & 15):
            x2d = x2d.contiguous()
        P = [p.detach() for p in params]
        cw, cb, box, var = P[0], P[1], P[2:8], P[8:14]
        with gemm_queue():
            cls = linear_fwd(x2d, cw, cb)
            h1b = linear_fwd(x2d, box[0], box[1], relu=True)
            h1v = linear_fwd(x2d, var[0], var[1], relu=True)
        with gemm_queue():
            h2b = linear_fwd(h1b, box[2], box[3], relu=True)
            h2v = linear_fwd(h1v, var[2], var[3], relu=True)
        with gemm_queue():
            ob = linear_fwd(h2b, box[4], box[5])
            ov = linear_fwd(h2v, var[4], var[5])
        ctx.params = params
        ctx.save_for_backward(x2d, h1b, h2b, h1v, h2v)
        ctx.xshape = shp
        lead = shp[:-1]
        return cls.reshape(*lead, -1), ob.reshape(*lead, -1), ov.reshape(*lead, -1)

    @staticmethod
    def backward(ctx, d_cls, d_box, d_var):
        x2d, h1b, h2b, h1v, h2v = ctx.saved_tensors
        params = ctx.params
        P = [p.detach() for p in params]
        cw, box, var = P[0], P[2:8], P[8:14]
        d_cls, d_box, d_var = (t.reshape(-1, t.shape[-1]).contiguous() for t in (d_cls, d_box, d_var))
        with gemm_queue():
            g2b = linear_dgrad(d_box, box[4], gate=h2b)             # masked by relu(h2): the gradient w.r.t. the second layer's pre-activation
            g2v = linear_dgrad(d_var, var[4], gate=h2v)
        with gemm_queue():
            g1b = linear_dgrad(g2b, box[2], gate=h1b)
            g1v = linear_dgrad(g2v, var[2], gate=h1v)
        dxc = linear_dgrad(d_cls, cw)
        with gemm_queue():
            dxb = linear_dgrad(g1b, box[0], resid=dxc)
            dxv = linear_dgrad(g1v, var[0])
        dx = add2(dxb, dxv)[0] if ctx.needs_input_grad[0] else None

        def wg(dy, xin, w, b):
            if w.requires_grad:
                gw = grad_buffer(w)
                gb = grad_buffer(b) if (b is not None and b.requires_grad) else None
                wgrad_raw(dy, dy.stride(0), xin, xin.stride(0), gw, gw.stride(0), dy.shape[0], gw.shape[0], xin.shape[1], dbias=gb, may_defer=True)
        wg(d_cls, x2d, params[0], params[1])
        for (dy3, dy2, dy1, h2, h1, off) in ((d_box, g2b, g1b, h2b, h1b, 2), (d_var, g2v, g1v, h2v, h1v, 8)):
            wg(dy3, h2, params[off + 4], params[off + 5])
            wg(dy2, h1, params[off + 2], params[off + 3])
            wg(dy1, x2d, params[off], params[off + 1])
        return (dx.reshape(ctx.xshape) if dx is not None else None,) + (None,) * len(params)


def linear(x, weight, bias=None, relu=False, resid=None, out_scale=1.0, rows=None):
    lo, hi = rows if rows is not None else (0, weight.shape[0])
    return LinearFn.apply(x, weight, bias, lo, hi, relu, resid, out_scale)


class MaskInfo:
    """Everything the step derives from the padding mask, produced by ONE launch (cdetr_mask_prep): `m` bool [B,h,w] (nearest
    down-sampling, A2/models/backbone.py:143), `mask_row` [B,w] / `mask_col` [B,h] uint8 (first row / column: the RCDA key masks),
    `pos_row` [B,w] / `pos_col` [B,h] (mask2pos, A2/models/transformer.py:497-503), `extent` [B,2] = un-padded (rows, columns)."""
    __slots__ = ("m", "mask_row", "mask_col", "pos_row", "pos_col", "extent")


def mask_prep(mask, h, w):
    B, H, W = mask.shape
    mask = mask.contiguous()
    dev = mask.device
    u8 = torch.empty(B * h * w + B * w + B * h, device=dev, dtype=torch.uint8)
    f = torch.empty(B * (w + h + 2), device=dev, dtype=torch.float32)
    mi = MaskInfo()
    m8 = u8[:B * h * w].view(B, h, w)
    mi.mask_row = u8[B * h * w:B * h * w + B * w].view(B, w)
    mi.mask_col = u8[B * h * w + B * w:].view(B, h)
    mi.pos_row, mi.pos_col, mi.extent = f[:B * w].view(B, w), f[B * w:B * (w + h)].view(B, h), f[B * (w + h):].view(B, 2)
    check(lib().cdetr_mask_prep(mask.data_ptr(), B, H, W, h, w, ptr(m8), ptr(mi.mask_row), ptr(mi.mask_col), ptr(mi.pos_row),
                                ptr(mi.pos_col), ptr(mi.extent), stream_ptr()), "cdetr_mask_prep")
    mi.m = m8.view(torch.bool)
    return mi


def exemplar_fwd_raw(x, rects, extent, per_image):
    B, h, w, Cc = x.shape
    K = rects.shape[1]
    idx = torch.empty((B, K), device=x.device, dtype=torch.int32)
    inv_cnt = torch.empty(B, device=x.device, dtype=torch.float32)
    pf = torch.empty((B, Cc), device=x.device, dtype=torch.float32)
    check(lib().cdetr_exemplar_fwd(ptr(x), ptr(rects), ptr(extent), int(per_image), B, h, w, Cc, K, ptr(idx), ptr(inv_cnt), ptr(pf),
                                   stream_ptr()), "cdetr_exemplar_fwd")
    return pf, idx, inv_cnt


class ExemplarFeatureFn(torch.autograd.Function):
    """pf[b] = mean_k x[b, cell(b, k)]: the exemplar feature of A2/models/backbone.py:116-131 (`per_image`: image b is conditioned on
    rects[b] scaled by its un-padded extent; else the reference's rects[0]-for-the-whole-batch rule).  x NHWC [B,h,w,C] -> [B,C]."""

    @staticmethod
    def forward(ctx, x, rects, extent, per_image):
        x = x.contiguous()
        pf, idx, inv_cnt = exemplar_fwd_raw(x, rects.contiguous().to(torch.float32), extent, per_image)
        ctx.save_for_backward(idx, inv_cnt)
        ctx.shape = x.shape
        return pf

    @staticmethod
    def backward(ctx, dpf):
        idx, inv_cnt = ctx.saved_tensors
        B, h, w, Cc = ctx.shape
        dx = torch.zeros(ctx.shape, device=dpf.device, dtype=torch.float32)
        check(lib().cdetr_exemplar_bwd(ptr(dpf.contiguous()), ptr(idx), ptr(inv_cnt), ptr(dx), B, h * w, Cc, idx.shape[1], stream_ptr()),
              "cdetr_exemplar_bwd")
        return dx, None, None, None


class AggrProjFn(torch.autograd.Function):
    """y[b] = [x[b], x[b] * pf[b]] W^T + bias with pf[b] = the exemplar feature of image b -- exemplar aggregation + 1x1 projection
    (A2/models/backbone.py:116-136, anchor_detr.py:119) WITHOUT the concatenated [.., 2C] feature map: the channel-wise product
    folds into a per-image effective weight  W_eff[b] = W[:, :C] + W[:, C:] * pf[b]  (2 MB), so the projection is one batched GEMM
    with K = C instead of 2C (half the FLOPs, no 82 MB concat, no x * pf pass).  Five launches forward (exemplar gather, effective
    weight + its transpose, GEMM), six backward (data gradient, zero-fill, weight gradient, weight / exemplar-feature gradients,
    exemplar scatter into dx): dW[:, :C] = sum_b dW_eff[b], dW[:, C:] = sum_b dW_eff[b] * pf[b], dpf[b] = sum_o dW_eff[b] * W[:, C:],
    dx[b, cell] += dpf[b] / K."""

    @staticmethod
    def forward(ctx, x, rects, extent, per_image, wparam, bparam):
        B, h, w, Cc = x.shape
        W2d = wparam.detach().reshape(wparam.shape[0], -1)                 # [d, 2C]
        assert W2d.is_contiguous() and W2d.shape[1] == 2 * Cc
        d = W2d.shape[0]
        x = x.contiguous()
        pf, idx, inv_cnt = exemplar_fwd_raw(x, rects.contiguous().to(torch.float32), extent, per_image)
        Weff = torch.empty((B, d, Cc), device=x.device, dtype=torch.float32)
        WeffT = torch.empty((B, Cc, d), device=x.device, dtype=torch.float32)
        check(lib().cdetr_aggr_weight_fwd(ptr(W2d), ptr(pf), ptr(Weff), ptr(WeffT), B, d, Cc, stream_ptr()), "cdetr_aggr_weight_fwd")
        y = torch.empty((B, h, w, d), device=x.device, dtype=torch.float32)
        # the bias pointer is shared by the batch items
        gemm_raw(x, Cc, Weff, Cc, y, d, h * w, d, Cc, bias=bparam.detach(), batch=B, sA=h * w * Cc, sB=d * Cc, sC=h * w * d)
        ctx.save_for_backward(x, pf, WeffT, idx, inv_cnt)
        ctx.wparam, ctx.bparam, ctx.d = wparam, bparam, d
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pf, WeffT, idx, inv_cnt = ctx.saved_tensors
        wparam, bparam, d = ctx.wparam, ctx.bparam, ctx.d
        B, h, w, Cc = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        gemm_raw(dy, d, WeffT, d, dx, Cc, h * w, Cc, d, batch=B, sA=h * w * d, sB=Cc * d, sC=h * w * Cc, precision=bwd_precision())
        z = zeros_flat(B * d * Cc + B * Cc, x.device)      # dW_eff and dpf: one fill (or none: ops.ZeroArena)
        dWeff, dpf = z[:B * d * Cc].view(B, d, Cc), z[B * d * Cc:].view(B, Cc)
        gb = grad_buffer(bparam) if bparam.requires_grad else None
        # (in line on the main chain: the exemplar gradient below needs it.  More, shorter pixel slices -- cdetr_wgrad_desc.wg_target -- make it
        # SLOWER: 33.9 -> 42.6 us at 1536 workgroups, the 4 MB of dW_eff are written once per slice)
        wgrad_raw(dy, d, x, Cc, dWeff, Cc, h * w, d, Cc, batch=B, sY=h * w * d, sX=h * w * Cc, sW=d * Cc, dbias=gb, wg_target=WGRAD_BESIDE_TARGET)
        W2d = wparam.detach().reshape(d, -1)
        gw = grad_buffer(wparam).reshape(d, -1) if wparam.requires_grad else None
        check(lib().cdetr_aggr_weight_bwd(ptr(dWeff), ptr(pf), ptr(W2d), ptr(gw), ptr(dpf), B, d, Cc, stream_ptr()), "cdetr_aggr_weight_bwd")
        check(lib().cdetr_exemplar_bwd(ptr(dpf), ptr(idx), ptr(inv_cnt), ptr(dx), B, h * w, Cc, idx.shape[1], stream_ptr()), "cdetr_exemplar_bwd")
        return dx, None, None, None, None, None


class TileParamFn(torch.autograd.Function):
    """A parameter laid out over the batch / query axes, with its gradient accumulated IN PLACE into `.grad` like every other parameter of
    the path (no autograd AccumulateGrad node: those pin the stream they were first run on, and a later graph capture on another stream
    would record a cross-stream edge):
      mode "tile":   w [Q, c] -> out[b, p*Q + q] = w[q]        (reference points: position.weight over batch and patterns, :114-116)
      mode "repeat": w [P, c] -> out[b, p*Q + q] = w[p]        (tgt: pattern.weight over batch and positions, :137-139)"""

    @staticmethod
    def forward(ctx, wparam, B, P, Q, mode):
        w = wparam.detach()
        if mode == "tile":
            out = w.unsqueeze(0).unsqueeze(0).expand(B, P, Q, w.shape[-1]).reshape(B, P * Q, w.shape[-1])
        else:
            out = w.unsqueeze(0).unsqueeze(2).expand(B, P, Q, w.shape[-1]).reshape(B, P * Q, w.shape[-1])
        ctx.wparam, ctx.cfg = wparam, (B, P, Q, mode)
        return out.contiguous()

    @staticmethod
    def backward(ctx, d):
        B, P, Q, mode = ctx.cfg
        wparam = ctx.wparam
        if wparam.requires_grad:
            g = grad_buffer(wparam)
            d = d.contiguous()
            c = d.shape[-1]
            if mode == "tile":                      # sum over the B*P slabs of [Q, c]
                colsum_(d.view(B * P, Q * c), g.view(-1))
            elif P == 1:
                colsum_(d.view(B * Q, c), g.view(-1))
            else:
                for b in range(B):
                    for p_ in range(P):
                        colsum_(d[b, p_ * Q:(p_ + 1) * Q], g[p_])
        return None, None, None, None, None


class BoxHeadFn(torch.autograd.Function):
    """boxes = sigmoid(tmp + [inverse_sigmoid(ref), 0, 0]) -- the tail of the box head (A2/models/transformer.py:193-203,
    A2/util/misc.py:475-479) as one launch forward and one backward instead of ~12 + ~25 tensor launches (clamp x3, rsub, div, log,
    add, cat, sigmoid and their autograd mirrors).  tmp [..., 4] with leading dims (L,) B, Q; ref [B, Q, 2]."""

    @staticmethod
    def forward(ctx, tmp, ref):
        tmp, ref = tmp.contiguous(), ref.contiguous()
        M, R = tmp.numel() // 4, ref.numel() // 2
        boxes = torch.empty_like(tmp)
        check(lib().cdetr_box_head_fwd(ptr(tmp), ptr(ref), ptr(boxes), M, R, stream_ptr()), "cdetr_box_head_fwd")
        ctx.save_for_backward(boxes, ref)
        return boxes

    @staticmethod
    def backward(ctx, d_boxes):
        boxes, ref = ctx.saved_tensors
        M, R = boxes.numel() // 4, ref.numel() // 2
        d_boxes = d_boxes.contiguous()
        d_tmp = torch.empty_like(boxes)
        d_ref = None
        if ctx.needs_input_grad[1]:
            d_ref = torch.empty_like(ref) if M == R else torch.zeros_like(ref)
        check(lib().cdetr_box_head_bwd(ptr(d_boxes), ptr(boxes), ptr(ref), ptr(d_tmp), ptr(d_ref), M, R, stream_ptr()), "cdetr_box_head_bwd")
        return d_tmp, d_ref


# ----------------------------------------------------------------------------------------------------- conv
def conv_geom_fwd(Hin, Win, kh, kw, stride, pad, dil):
    Hout = (Hin + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    Wout = (Win + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    dense = kh == 1 and kw == 1 and stride == 1 and pad == 0
    g = _geom() if dense else _geom(_ffi.ROWS_CONV_FWD, Hin, Win, Hout, Wout, kh, kw, stride, pad, dil)
    return g, Hout, Wout


TWINS = os.environ.get("CDETR_TWINS", "1") != "0"


EXPAND_PLANES = os.environ.get("CDETR_EXPAND_PLANES", "1") != "0"
ENC_TWINS = os.environ.get("CDETR_ENC_TWINS", "1") != "0"      # encoder backward: bf16 twins through LayerNorm backward / data-gradient epilogues (A/B)


def expand_planes():
    """Whether a bottleneck's 3x3 convolution also writes the lo plane of its output, so that the expanding 1x1 convolution behind it runs on
    the direct-to-LDS kernel (split-bf16 forward only; needs the pre-split weight images, i.e. a trainer / inference engine around the model)."""
    return EXPAND_PLANES and PRECISION == 1 and MIRROR is not None


def conv_fwd(x, weight, scale, bias, stride=1, pad=0, dil=1, relu=False, resid=None, twin=False, xs=None, split=False, out=None, out16=None,
             out_groups=False):
    """x [N,H,W,Cin] NHWC -> [N,Ho,Wo,Cout]; weight logical [Cout,Cin,kh,kw] in channels_last memory.
    y = relu?( conv(x, W) * scale[c] + bias[c] + resid )   (A2/models/resnet.py:140-160 + backbone.py:50-60).
    twin: -> (y, bf16 copy of y written by the same epilogue).  split: -> (y, hi, lo) = y with its split-bf16 planes.
    xs = (hi, lo) planes of x (from the producer's epilogue): the tile kernels read them instead of x.
    x / resid may be `Groups` (interleaved split-bf16 form, no fp32 tensor); out_groups: y is returned as `Groups` (no fp32 output).  Any of
    the three sends the problem to the direct-to-LDS tile kernel, which needs the pre-split weight image (ops.MIRROR)."""
    Nb, H, W, Cin = x.shape
    Cout, Cin_w, kh, kw = weight.shape
    xg = x if isinstance(x, Groups) else None
    rg = resid if isinstance(resid, Groups) else None
    assert Cin_w == Cin and (xg is not None or x.is_contiguous())
    g, Ho, Wo = conv_geom_fwd(H, W, kh, kw, stride, pad, dil)
    dev = x.device
    yg = Groups.empty((Nb, Ho, Wo, Cout), dev) if out_groups else None
    y = None if out_groups else (out if out is not None else torch.empty((Nb, Ho, Wo, Cout), device=dev, dtype=torch.float32))
    y16 = out16 if out16 is not None else (torch.empty((Nb, Ho, Wo, Cout), device=dev, dtype=torch.bfloat16) if (twin or split) else None)
    ylo = torch.empty((Nb, Ho, Wo, Cout), device=dev, dtype=torch.bfloat16) if split else None
    sp = MIRROR.lookup_fwd(weight, scale) if MIRROR is not None else None
    assert sp is not None or (xg is None and rg is None and not out_groups), "interleaved-group operands need the pre-split weight images (a trainer / inference engine around the model)"
    xh, xl = xs if xs is not None else (None, None)
    gemm_raw(None if xg is not None else x, Cin, weight, kh * kw * Cin, y, Cout, Nb * Ho * Wo, Cout, Cin, taps=kh * kw, w_scale=scale, bias=bias,
             relu=relu, resid=rg.t if rg is not None else resid, ldr=Cout, geom=g, B_split=sp, C16=y16, C16lo=ylo, A16=xh if xl is not None else None, A16lo=xl,
             A_split=xg.t if xg is not None else None, C_split=yg.t if yg is not None else None, resid_groups=rg is not None)
    if out_groups:
        y = yg
    if split:
        return y, y16, ylo
    return (y, y16) if twin else y


DGRAD_PRIO = os.environ.get("CDETR_DGRAD_PRIO", "1") != "0"     # the backbone's data gradients at raised wave priority (they share the chip with the weight gradients)
TWIN_ONLY = os.environ.get("CDETR_TWIN_ONLY", "1") != "0"      # inner gradients of a bottleneck leave their kernel as the bf16 twin alone (A/B)


def conv_dgrad(dz, weight, scale, in_hw, stride=1, pad=0, dil=1, gate=None, resid=None, twin=False, dz16=None, gate16=None, out=None,
               twin_only=False):
    """dx [N,Hin,Win,Cin] = conv_transpose(dz * scale, W) (+ resid), zeroed where gate <= 0.  twin: -> (dx, bf16 copy of dx);
    dz16: the bf16 twin of dz (read instead of dz by the plain-bf16 tile kernels).
    twin_only (with twin): nothing reads the fp32 gradient (every consumer is a plain-bf16 contraction fed by the twin) -> (None, twin)
    when the problem runs on the direct-to-LDS kernel (cdetr_gemm_desc.C == NULL); otherwise both are written as usual.
    `dz` may then be None as well (its shape comes from dz16)."""
    Nb, Ho, Wo, Cout = (dz if dz is not None else dz16).shape
    Cout_w, Cin, kh, kw = weight.shape
    Hin, Win = in_hw
    dense = kh == 1 and kw == 1 and stride == 1 and pad == 0
    g = _geom() if dense else _geom(_ffi.ROWS_CONV_DGRAD, Ho, Wo, Hin, Win, kh, kw, stride, pad, dil)
    m = MIRROR.lookup(weight, scale) if MIRROR is not None else None
    g16_only = False
    if isinstance(gate, Groups):
        # the activation exists as interleaved groups only: its ReLU mask comes from the bf16 twin (same signs), which only the direct-to-LDS
        # kernel reads -- ops.use_groups made sure this problem is one of its (>= 4096 rows, twins, weight images); `gate` itself is just the flag
        assert gate16 is not None and m is not None, "a grouped activation gates a data gradient through its bf16 twin (direct-to-LDS kernel)"
        gate, g16_only = gate.t, True
    # (the consumers must be twin-fed too: the direct-to-LDS data gradient and the tile-class weight gradient -- more than 1024 rows,
    # 8-channel granularity: csrc/igemm.hip wgrad_has_twins / wgrad_is_direct; a few-row problem keeps its fp32 gradient)
    no_fp32 = (twin_only and twin and TWIN_ONLY and out is None and m is not None and m[3] is not None and dz16 is not None
               and bwd_precision() == 3 and Cout % 64 == 0 and Cin % 64 == 0 and Nb * Hin * Win > 1024
               and os.environ.get("CDETR_WGRAD_TWINS", "1") != "0")
    assert dz is not None or (m is not None and dz16 is not None), "a gradient that exists as a twin only needs the weight images"
    dx = out if out is not None else (None if no_fp32 else torch.empty((Nb, Hin, Win, Cin), device=dz16.device if dz is None else dz.device, dtype=torch.float32))
    dx16 = torch.empty((Nb, Hin, Win, Cin), device=(dz16 if dz is None else dz).device, dtype=torch.bfloat16) if twin else None
    if m is not None:     # FrozenBN scale is folded into the mirror
        gemm_raw(dz, Cout, m[0], m[1], dx, Cin, Nb * Hin * Win, Cin, Cout, taps=kh * kw, b_layout=0,
                 gate=gate, ldg=Cin, resid=resid, ldr=Cin, geom=g, B_split=m[2], B16=m[3], precision=bwd_precision(), C16=dx16, A16=dz16,
                 gate16=gate16 if gate is not None else None, prio=DGRAD_PRIO, gate16_only=g16_only)
    else:
        gemm_raw(dz, Cout, weight, Cin, dx, Cin, Nb * Hin * Win, Cin, Cout, taps=kh * kw, b_layout=1, w_scale=scale,
                 gate=gate, ldg=Cin, resid=resid, ldr=Cin, geom=g, precision=bwd_precision(), C16=dx16)
    return (dx, dx16) if twin else dx


def conv_wgrad_(dz, x, weight, scale, stride=1, pad=0, dil=1, dz16=None, x16=None):
    """dz may be None when the gradient exists as its twin only (conv_dgrad(twin_only=True))."""
    Nb, Ho, Wo, Cout = (dz if dz is not None else dz16).shape
    _, H, W, Cin = x.shape
    kh, kw = weight.shape[2:]
    g, Ho2, Wo2 = conv_geom_fwd(H, W, kh, kw, stride, pad, dil)
    assert (Ho2, Wo2) == (Ho, Wo)
    if isinstance(x, Groups):      # no fp32 tensor: only the twin-fed tile kernel can run it (csrc/igemm.hip wgrad_has_twins / wgrad_is_direct)
        assert x16 is not None and dz16 is not None and bwd_precision() == 3 and Cout % 8 == 0 and Cin % 8 == 0 and Nb * Ho * Wo > 1024 \
            and os.environ.get("CDETR_WGRAD_TWINS", "1") != "0", "a grouped activation feeds a weight gradient through its bf16 twin"
        x = None                   # cdetr_wgrad_desc.X == NULL: the C side refuses any kernel class but the twin-fed one
    gw = grad_buffer(weight)
    assert gw.is_contiguous(memory_format=torch.channels_last) or (kh == 1 and kw == 1)
    wgrad_raw(dz, Cout, x, Cin, gw, kh * kw * Cin, Nb * Ho * Wo, Cout, Cin, taps=kh * kw, w_scale=scale, geom=g, may_defer=True,
              dY16=dz16, X16=x16)


def maxpool3x3s2(x, split=False):
    """split: -> (y, hi, lo) with the split-bf16 planes of y written by the same pass."""
    Nb, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((Nb, Ho, Wo, Cc), device=x.device, dtype=torch.float32)
    if split:
        yh = torch.empty((Nb, Ho, Wo, Cc), device=x.device, dtype=torch.bfloat16)
        yl = torch.empty((Nb, Ho, Wo, Cc), device=x.device, dtype=torch.bfloat16)
        check(lib().cdetr_maxpool3x3s2_split(ptr(x), ptr(y), ptr(yh), ptr(yl), Nb, H, W, Cc, stream_ptr()), "cdetr_maxpool3x3s2_split")
        return y, yh, yl
    check(lib().cdetr_maxpool3x3s2(ptr(x), ptr(y), Nb, H, W, Cc, stream_ptr()), "cdetr_maxpool3x3s2")
    return y


# ----------------------------------------------------------------------------------------------------- RCDA core
RCDA_SKIP_SAVE = os.environ.get("CDETR_RCDA_SKIP_SAVE", "1") != "0"    # inference: no attention maps written (A/B knob)
RCDA_SAVE = True                                                        # ops.scope(RCDA_SAVE=False): the forward bodies run for inference
RCDA_SLICES = os.environ.get("CDETR_RCDA_SLICES", "1") != "0"        # key-row slices of the two-step forward (cdetr_rcda_fwd_desc.ws; A/B knob)
FUSE_RCDA_DQ = os.environ.get("CDETR_RCDA_FUSE_DQ", "1") != "0"     # query gradients inside the dS launch (A/B knob)
FUSE_RCDA_DK = os.environ.get("CDETR_RCDA_FUSE_DK", "1") != "0"     # key gradients too (split-bf16 mode)


def rcda_pads(H, W):
    return (H + 7) & ~7, (W + 3) & ~3


def rcda_fwd_raw(q_row, q_col, k_row, k_col, v, mask_row, mask_col, nh, save=None):
    """-> (out [N,L,E], a_row [N,nh,L,Wp], a_col [N,nh,L,Hp]); inputs contiguous fp32.
    save: keep the two attention maps for the backward pass (default: ops.RCDA_SAVE -- True unless the caller runs the forward bodies for
    inference, `with ops.scope(RCDA_SAVE=False)`: transformer.py's no-grad branch; torch.is_grad_enabled() cannot tell, it is False inside
    every autograd.Function.forward).  Without it the kernels get NULL and skip the 2 x 8 MB of stores per encoder call (round 6)
    -> (out, None, None)."""
    N, L, E = q_row.shape
    H, W = v.shape[1:3]
    assert E == nh * 32, "the RCDA kernels are specialised for head_dim 32"
    Hp, Wp = rcda_pads(H, W)
    if save is None:
        save = RCDA_SAVE or not RCDA_SKIP_SAVE
    out = torch.empty((N, L, E), device=v.device, dtype=torch.float32)
    a_row = torch.empty((N, nh, L, Wp), device=v.device, dtype=torch.float32) if save else None
    a_col = torch.empty((N, nh, L, Hp), device=v.device, dtype=torch.float32) if save else None
    d = RcdaFwdDesc()
    d.N, d.L, d.H, d.W, d.nh, d.scale = N, L, H, W, nh, 32 ** -0.5
    d.precision = PRECISION
    d.q_row, d.q_col, d.k_row, d.k_col, d.v = ptr(q_row), ptr(q_col), ptr(k_row), ptr(k_col), ptr(v)
    d.mask_row, d.mask_col = ptr(mask_row), ptr(mask_col)
    d.out, d.a_row, d.a_col = ptr(out), ptr(a_row), ptr(a_col)
    if RCDA_SLICES:
        d.ws, d.ws_bytes = ptr(splitk_ws()), SPLITK_BYTES
    # compulsory bytes: both query sets, both key sets, V, the output and the two saved attention maps
    fl = 2.0 * N * nh * L * (H * W * 32 + (H + W) * 32)
    with _Timed("rcda_fwd", fl, None,
                4.0 * N * (3 * L * E + (H + W) * E + H * W * E + nh * L * (Hp + Wp)), fl * _TERMS[d.precision]):
        check(lib().cdetr_rcda_fwd(C.byref(d), stream_ptr()), "cdetr_rcda_fwd")
    return out, a_row, a_col


def rcda_zero_numel(v, k_row, k_col):
    return v.numel() + k_row.numel() + k_col.numel()


def rcda_bwd_raw(d_out, q_row, q_col, k_row, k_col, v, a_row, a_col, nh, zbuf=None):
    """-> (dq_row, dq_col, dk_row, dk_col, dv).  `zbuf`: a ZEROED fp32 buffer of rcda_zero_numel(...) elements for everything the
    backward accumulates into atomically (dV and both key gradients) -- a stack of layers zero-fills one arena for all its calls."""
    N, L, E = q_row.shape
    H, W = v.shape[1:3]
    Hp, Wp = rcda_pads(H, W)
    d_out = d_out.contiguous()
    if zbuf is None:
        zbuf = torch.zeros(rcda_zero_numel(v, k_row, k_col), device=v.device, dtype=torch.float32)
    d_v = zbuf[:v.numel()].view(v.shape)
    d = RcdaBwdDesc()
    d.N, d.L, d.H, d.W, d.nh, d.scale = N, L, H, W, nh, 32 ** -0.5
    d.precision = (3 if bwd_precision() == 3 else 1) if PRECISION == 1 else PRECISION      # plain-bf16 products with the bf16 backward
    d.d_out, d.a_row, d.a_col, d.v = ptr(d_out), ptr(a_row), ptr(a_col), ptr(v)
    d.d_v = ptr(d_v)
    # logits -> projected query gradients: fused into the dS launch (the kernel still holds dS_row / dS_col in LDS)
    dq_row = torch.empty_like(q_row)
    dq_col = torch.empty_like(q_col)
    fuse_dq = FUSE_RCDA_DQ
    fuse_dk = fuse_dq and FUSE_RCDA_DK and PRECISION == 1
    dk = zbuf[v.numel():]
    dk_row, dk_col = dk[:k_row.numel()].view(k_row.shape), dk[k_row.numel():].view(k_col.shape)
    if fuse_dq:
        d.k_row, d.k_col, d.dq_row, d.dq_col = ptr(k_row), ptr(k_col), ptr(dq_row), ptr(dq_col)
    if fuse_dk:
        d.q_row, d.q_col, d.dk_row, d.dk_col = ptr(q_row), ptr(q_col), ptr(dk_row), ptr(dk_col)
        ds_row = ds_col = None                     # the logit gradients never leave the chip
    else:
        ds_row, ds_col = torch.empty_like(a_row), torch.empty_like(a_col)
        d.ds_row, d.ds_col = ptr(ds_row), ptr(ds_col)
    # compulsory bytes: d_out, attention maps, V, queries / keys in; dq (x2), dk (x2), dV out
    fl = 2.0 * N * nh * L * (2 * H * W * 32)
    with _Timed("rcda_bwd", fl, None,
                4.0 * N * (5 * L * E + 2 * (H + W) * E + 2 * H * W * E + nh * L * (Hp + Wp)), fl * _TERMS[d.precision]):
        check(lib().cdetr_rcda_bwd(C.byref(d), stream_ptr()), "cdetr_rcda_bwd")
    # two-level batch (image x head): one launch per contraction for the whole batch of images
    if not fuse_dq:
        gemm_raw(ds_row, Wp, k_row, E, dq_row, E, L, 32, W, b_layout=1, batch=N * nh, sA=L * Wp, sB=32, sC=32,
                 batch_inner=nh, sA2=nh * L * Wp, sB2=W * E, sC2=L * E)
        gemm_raw(ds_col, Hp, k_col, E, dq_col, E, L, 32, H, b_layout=1, batch=N * nh, sA=L * Hp, sB=32, sC=32,
                 batch_inner=nh, sA2=nh * L * Hp, sB2=H * E, sC2=L * E)
    if not fuse_dk:
        wgrad_raw(ds_row, Wp, q_row, E, dk_row, E, L, W, 32, batch=N * nh, sY=L * Wp, sX=32, sW=32,
                  batch_inner=nh, sY2=nh * L * Wp, sX2=L * E, sW2=W * E)
        wgrad_raw(ds_col, Hp, q_col, E, dk_col, E, L, H, 32, batch=N * nh, sY=L * Hp, sX=32, sW=32,
                  batch_inner=nh, sY2=nh * L * Hp, sX2=L * E, sW2=H * E)
    return dq_row, dq_col, dk_row, dk_col, d_v


class RcdaCoreFn(torch.autograd.Function):
    """Fused two-softmax + double contraction of A2/models/row_column_decoupled_attention.py:215-309.
    q_row,q_col [N,L,E]; k_row [N,W,E]; k_col [N,H,E]; v [N,H,W,E]; masks uint8 [N,W] / [N,H] or None -> out [N,L,E]."""

    @staticmethod
    def forward(ctx, q_row, q_col, k_row, k_col, v, mask_row, mask_col, nh):
        q_row, q_col, k_row, k_col, v = [t.contiguous() for t in (q_row, q_col, k_row, k_col, v)]
        out, a_row, a_col = rcda_fwd_raw(q_row, q_col, k_row, k_col, v, mask_row, mask_col, nh)
        ctx.save_for_backward(q_row, q_col, k_row, k_col, v, a_row, a_col)
        ctx.nh = nh
        return out

    @staticmethod
    def backward(ctx, d_out):
        q_row, q_col, k_row, k_col, v, a_row, a_col = ctx.saved_tensors
        return rcda_bwd_raw(d_out, q_row, q_col, k_row, k_col, v, a_row, a_col, ctx.nh) + (None, None, None)


def rcda_core(q_row, q_col, k_row, k_col, v, mask_row, mask_col, nh):
    return RcdaCoreFn.apply(q_row, q_col, k_row, k_col, v, mask_row, mask_col, nh)


# ----------------------------------------------------------------------------------------------------- decoder self-attention
def mha_fwd_raw(qk, v, nh):
    N, L, E2 = qk.shape
    E = E2 // 2
    o = torch.empty((N, L, E), device=qk.device, dtype=torch.float32)
    lse = torch.empty((N, nh, L), device=qk.device, dtype=torch.float32)
    check(lib().cdetr_mha_fwd(ptr(qk), ptr(v), ptr(o), ptr(lse), N, L, nh, (E // nh) ** -0.5, PRECISION, stream_ptr()), "cdetr_mha_fwd")
    return o, lse


def mha_bwd_raw(qk, v, o, d_o, lse, nh):
    N, L, E2 = qk.shape
    E = E2 // 2
    d_qk = torch.empty_like(qk)
    d_v = torch.empty_like(v)
    work = torch.empty((N, nh, L), device=qk.device, dtype=torch.float32)
    prec = bwd_precision() if MHA_BWD_BF16 else PRECISION      # 3: split-bf16 scores, plain-bf16 gradient contractions (A/B: CDETR_MHA_BWD_BF16=0)
    check(lib().cdetr_mha_bwd(ptr(qk), ptr(v), ptr(o), ptr(d_o), ptr(lse), ptr(d_qk), ptr(d_v), ptr(work), N, L, nh,
                              (E // nh) ** -0.5, prec, stream_ptr()), "cdetr_mha_bwd")
    return d_qk, d_v


class MhaCoreFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d)) v per head for the decoder queries (qk [N,L,2E] = q | k, v [N,L,E]) in one fused kernel."""

    @staticmethod
    def forward(ctx, qk, v, nh):
        assert qk.shape[-1] == 2 * nh * 32, "the MHA kernels are specialised for head_dim 32"
        qk, v = qk.contiguous(), v.contiguous()
        o, lse = mha_fwd_raw(qk, v, nh)
        ctx.save_for_backward(qk, v, o, lse)
        ctx.nh = nh
        return o

    @staticmethod
    def backward(ctx, d_o):
        qk, v, o, lse = ctx.saved_tensors
        d_qk, d_v = mha_bwd_raw(qk, v, o, d_o.contiguous(), lse, ctx.nh)
        return d_qk, d_v, None


def mha_core(qk, v, nh):
    return MhaCoreFn.apply(qk, v, nh)


# ----------------------------------------------------------------------------------------------------- layer norm & glue
GN_SPLIT = os.environ.get("CDETR_GN_SPLIT", "1") != "0"      # A/B: 0 = one workgroup per (image, group)


class GroupNormNHWCFn(torch.autograd.Function):
    """nn.GroupNorm(G, C) on an NHWC activation [B,h,w,C] (A2/models/anchor_detr.py:86-92) -- fwd / bwd are one kernel each, no layout
    round trip; dgamma / dbeta accumulate straight into the parameters' gradient buffers."""

    @staticmethod
    def forward(ctx, x, wparam, bparam, G, eps):
        B, h, w, Cc = x.shape
        x = x.contiguous()
        y = torch.empty_like(x)
        mean = torch.empty(B * G, device=x.device, dtype=torch.float32)
        rstd = torch.empty(B * G, device=x.device, dtype=torch.float32)
        ws = torch.empty(B * ((h * w + 31) // 32) * G * 3, device=x.device, dtype=torch.float32) if GN_SPLIT else None      # per-chunk statistics of the split form
        check(lib().cdetr_groupnorm_fwd_ws(ptr(x), ptr(wparam.detach()), ptr(bparam.detach()), ptr(y), ptr(mean), ptr(rstd), B, h * w, Cc, G,
                                           eps, ptr(ws), ws.numel() * 4 if ws is not None else 0, stream_ptr()), "cdetr_groupnorm_fwd_ws")
        ctx.save_for_backward(x, mean, rstd)
        ctx.wparam, ctx.bparam, ctx.G = wparam, bparam, G
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        B, h, w, Cc = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        wparam, bparam = ctx.wparam, ctx.bparam
        gw = grad_buffer(wparam) if wparam.requires_grad else torch.zeros_like(wparam)
        gb = grad_buffer(bparam) if bparam.requires_grad else torch.zeros_like(bparam)
        ws = torch.empty(B * ((h * w + 31) // 32) * ctx.G * 2, device=x.device, dtype=torch.float32) if GN_SPLIT else None
        check(lib().cdetr_groupnorm_bwd_ws(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(wparam.detach()), ptr(dx), ptr(gw), ptr(gb), B, h * w,
                                           Cc, ctx.G, ptr(ws), ws.numel() * 4 if ws is not None else 0, stream_ptr()), "cdetr_groupnorm_bwd_ws")
        return dx, None, None, None, None


def ln_fwd_raw(x2d, weight, bias, eps=1e-5):
    rows, Cc = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty(rows, device=x2d.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x2d.device, dtype=torch.float32)
    check(lib().cdetr_layernorm_fwd(ptr(x2d), ptr(weight), ptr(bias), ptr(y), ptr(mean), ptr(rstd), rows, Cc, eps, stream_ptr()),
          "cdetr_layernorm_fwd")
    return y, mean, rstd


def ln_fwd_add_raw(x2d, weight, bias, eps, a1, a2=None):
    """LayerNorm forward that also returns y + a1 (and y + a2): -> (y, mean, rstd, y + a1, y + a2 or None)."""
    rows, Cc = x2d.shape
    y = torch.empty_like(x2d)
    o1 = torch.empty_like(x2d)
    o2 = torch.empty_like(x2d) if a2 is not None else None
    mean = torch.empty(rows, device=x2d.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x2d.device, dtype=torch.float32)
    check(lib().cdetr_layernorm_fwd_add(ptr(x2d), ptr(weight), ptr(bias), ptr(y), ptr(mean), ptr(rstd), ptr(a1), ptr(a2), ptr(o1), ptr(o2),
                                        rows, Cc, eps, stream_ptr()), "cdetr_layernorm_fwd_add")
    return y, mean, rstd, o1, o2


def ln_bwd_raw(dy2d, x2d, mean, rstd, weight, gw, gb, add=None, merge=None, bcast=None, twin=False):
    """dx (+ add); dgamma / dbeta accumulate into gw / gb.  merge = (g1, g2 | None, acc1 | None, acc2 | None): the incoming gradient is
    dy2d + g1 + g2 and acc1 += g1, acc2 += g2 in place (cdetr_grad_merge folded in; C = 256).  bcast = (Br, Bc, sr, sc, H, W), with merge:
    + sr * Br[n,x] + sc * Bc[n,y] as well (cdetr_bcast_add2_sum folded in).  twin (C = 256): -> (dx, bf16 twin of dx from the same pass)."""
    rows, Cc = x2d.shape
    dx = torch.empty_like(x2d)
    twin = twin and Cc == 256
    dx16 = torch.empty(x2d.shape, device=x2d.device, dtype=torch.bfloat16) if twin else None
    if merge is not None and Cc == 256:
        g1, g2, acc1, acc2 = merge
        Br, Bc, sr, sc, bh, bw = bcast if bcast is not None else (None, None, 0.0, 0.0, 1, 1)
        check(lib().cdetr_layernorm_bwd_merge(ptr(dy2d), ptr(g1), ptr(g2), ptr(acc1), ptr(acc2), ptr(Br), ptr(Bc), sr, sc, bh, bw, ptr(x2d),
                                              ptr(mean), ptr(rstd), ptr(weight), ptr(add), ptr(dx), ptr(gw), ptr(gb), rows, Cc, ptr(dx16), stream_ptr()),
              "cdetr_layernorm_bwd_merge")
        return (dx, dx16) if twin else dx
    if merge is not None:
        assert bcast is None
        dy2d = grad_merge(dy2d, *merge)
    check(lib().cdetr_layernorm_bwd(ptr(dy2d), ptr(x2d), ptr(mean), ptr(rstd), ptr(weight), ptr(add), ptr(dx), ptr(gw), ptr(gb),
                                    rows, Cc, ptr(dx16), stream_ptr()), "cdetr_layernorm_bwd")
    return (dx, dx16) if twin else dx


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim on the fused HIP kernels; parameter gradients accumulate in place into .grad."""

    @staticmethod
    def forward(ctx, x, wparam, bparam, eps):
        shp = x.shape
        x2d = x.reshape(-1, shp[-1]).contiguous()
        y, mean, rstd = ln_fwd_raw(x2d, wparam.detach(), bparam.detach(), eps)
        ctx.save_for_backward(x2d, mean, rstd)
        ctx.wparam, ctx.bparam, ctx.shp = wparam, bparam, shp
        return y.reshape(shp)

    @staticmethod
    def backward(ctx, dy):
        x2d, mean, rstd = ctx.saved_tensors
        dy2d = dy.reshape(x2d.shape).contiguous()
        dx = ln_bwd_raw(dy2d, x2d, mean, rstd, ctx.wparam.detach(), grad_buffer(ctx.wparam), grad_buffer(ctx.bparam))
        return dx.reshape(ctx.shp), None, None, None


def layer_norm(x, weight, bias, eps=1e-5):
    return LayerNormFn.apply(x, weight, bias, eps)


def posadd2(X, Prow, Pcol):
    N, H, W, Cc = X.shape
    Qr, Qc = torch.empty_like(X), torch.empty_like(X)
    check(lib().cdetr_posadd2(ptr(X), ptr(Prow), ptr(Pcol), ptr(Qr), ptr(Qc), N, H, W, Cc, stream_ptr()), "cdetr_posadd2")
    return Qr, Qc


def posadd2_hw_reduce(X, Prow, Pcol):
    """posadd2(X, Prow, Pcol) and hw_reduce(X, X, Prow, Pcol, 1/H, 1/W) in one launch -> (Qr, Qc, Kr, Kc)."""
    N, H, W, Cc = X.shape
    Qr, Qc = torch.empty_like(X), torch.empty_like(X)
    Kr = torch.empty((N, W, Cc), device=X.device, dtype=torch.float32)
    Kc = torch.empty((N, H, Cc), device=X.device, dtype=torch.float32)
    check(lib().cdetr_posadd2_hw_reduce(ptr(X), ptr(Prow), ptr(Pcol), ptr(Qr), ptr(Qc), ptr(Kr), ptr(Kc), N, H, W, Cc, 1.0 / H, 1.0 / W,
                                        stream_ptr()), "cdetr_posadd2_hw_reduce")
    return Qr, Qc, Kr, Kc


def hw_reduce(Xr, Xc, Ar, Ac, scale_r, scale_c):
    N, H, W, Cc = Xr.shape
    Or = torch.empty((N, W, Cc), device=Xr.device, dtype=torch.float32)
    Oc = torch.empty((N, H, Cc), device=Xr.device, dtype=torch.float32)
    check(lib().cdetr_hw_reduce(ptr(Xr), ptr(Xc), ptr(Ar), ptr(Ac), ptr(Or), ptr(Oc), N, H, W, Cc, scale_r, scale_c, stream_ptr()),
          "cdetr_hw_reduce")
    return Or, Oc


def bcast_add2_sum(T, T2, T3, Br, Bc, sr, sc):
    """T + T2 + T3 + sr * Br[n,x,:] + sc * Bc[n,y,:] in one pass (T2 / T3 optional)."""
    N, H, W, Cc = T.shape
    out = torch.empty_like(T)
    check(lib().cdetr_bcast_add2_sum(ptr(T), ptr(T2), ptr(T3), ptr(Br), ptr(Bc), ptr(out), N, H, W, Cc, sr, sc, stream_ptr()), "cdetr_bcast_add2_sum")
    return out


def bcast_add2(T, Br, Bc, sr, sc):
    N, H, W, Cc = T.shape
    out = torch.empty_like(T)
    check(lib().cdetr_bcast_add2(ptr(T), ptr(Br), ptr(Bc), ptr(out), N, H, W, Cc, sr, sc, stream_ptr()), "cdetr_bcast_add2")
    return out


def _wg(dy2d, x2d, wparam, bparam, lo, hi):
    """dW[lo:hi] += dy^T x (+ db[lo:hi] += colsum dy) straight into the parameters' gradient buffers."""
    if not wparam.requires_grad:
        return
    gw = grad_buffer(wparam)[lo:hi]
    gb = grad_buffer(bparam)[lo:hi] if (bparam is not None and bparam.requires_grad) else None
    wgrad_raw(dy2d, dy2d.stride(0), x2d, x2d.stride(0), gw, gw.stride(0), dy2d.shape[0], hi - lo, x2d.shape[1], dbias=gb,
              may_defer=True)


class EncoderLayerFn(torch.autograd.Function):
    """One RCDA encoder layer (A2/models/transformer.py:242-279) as ONE autograd node with a hand-scheduled backward:
    fused positional-add / key-mean prologue, MFMA projections, fused RCDA core, residual adds fused into GEMM epilogues,
    fused LayerNorm, and in backward every multi-consumer gradient sum (src feeds q_row, q_col, v, both key means and the
    residual) chained through the data-gradient epilogues -- no autograd accumulation passes, no elementwise temporaries."""

    @staticmethod
    def forward(ctx, src, posemb_row, posemb_col, mask_row, mask_col, layer, anchor):
        N, H, W, Cc = src.shape
        R = N * H * W
        att = layer.self_attn
        E, nh = att.embed_dim, att.num_heads
        Wi, bi = att.in_proj_weight.detach(), att.in_proj_bias.detach()
        X = src.contiguous()
        Prow, Pcol = posemb_row.contiguous(), posemb_col.contiguous()
        Qr, Qc, Kr, Kc = posadd2_hw_reduce(X, Prow, Pcol)       # positional adds + key means: one launch
        with gemm_queue():           # the five in-projections are independent: one grouped submission
            q_row = linear_fwd(Qr.view(R, Cc), Wi[0:E], bi[0:E]).view(N, H * W, E)
            q_col = linear_fwd(Qc.view(R, Cc), Wi[E:2 * E], bi[E:2 * E]).view(N, H * W, E)
            k_row = linear_fwd(Kr.view(N * W, Cc), Wi[2 * E:3 * E], bi[2 * E:3 * E]).view(N, W, E)
            k_col = linear_fwd(Kc.view(N * H, Cc), Wi[3 * E:4 * E], bi[3 * E:4 * E]).view(N, H, E)
            v = linear_fwd(X.view(R, Cc), Wi[4 * E:5 * E], bi[4 * E:5 * E]).view(N, H, W, E)
        o, a_row, a_col = rcda_fwd_raw(q_row, q_col, k_row, k_col, v, mask_row, mask_col, nh)
        Y1 = linear_fwd(o.view(R, E), att.out_proj.weight.detach(), att.out_proj.bias.detach(), resid=X.view(R, Cc))
        X1, mu1, rs1 = ln_fwd_raw(Y1, layer.norm1.weight.detach(), layer.norm1.bias.detach(), layer.norm1.eps)
        f = layer.ffn
        Hd = linear_fwd(X1, f.linear1.weight.detach(), f.linear1.bias.detach(), relu=True)
        Y2 = linear_fwd(Hd, f.linear2.weight.detach(), f.linear2.bias.detach(), resid=X1)
        X2, mu2, rs2 = ln_fwd_raw(Y2, f.norm2.weight.detach(), f.norm2.bias.detach(), f.norm2.eps)
        ctx.save_for_backward(X, Qr, Qc, Kr, Kc, q_row, q_col, k_row, k_col, v, a_row, a_col, o, Y1, mu1, rs1, X1, Hd, Y2, mu2, rs2)
        ctx.layer, ctx.dims = layer, (N, H, W, Cc, E, nh)
        return X2.view(N, H, W, Cc)

    @staticmethod
    def backward(ctx, dX2):
        with wgrad_queue():          # the layer's 8 parameter gradients: one grouped submission at the end
            return EncoderLayerFn._backward(ctx, dX2)

    @staticmethod
    def _backward(ctx, dX2, accR=None, accC=None, zbuf=None, defer_out=False):
        """accR / accC: d(posemb_row) / d(posemb_col) accumulated by the layers ABOVE (EncoderStackFn): added inside the data-gradient
        epilogues, the returned d(posemb) then already contain them.
        dX2 may be the PARTS of the output gradient, a tuple (t, t2, t3, dKr, dKc) as a layer above returns them under defer_out=True
        (d(src) = t + t2 + t3 + broadcast(dKr) / H + broadcast(dKc) / W): they are summed inside this layer's first LayerNorm backward."""
        (X, Qr, Qc, Kr, Kc, q_row, q_col, k_row, k_col, v, a_row, a_col, o, Y1, mu1, rs1, X1, Hd, Y2, mu2, rs2) = ctx.saved_tensors
        layer = ctx.layer
        N, H, W, Cc, E, nh = ctx.dims
        R = N * H * W
        att, f = layer.self_attn, layer.ffn
        Wi = att.in_proj_weight.detach()
        Wip, bip = att.in_proj_weight, att.in_proj_bias
        ln2 = (Y2, mu2, rs2, f.norm2.weight.detach(), grad_buffer(f.norm2.weight), grad_buffer(f.norm2.bias))
        # ---- FFN (post-norm): X2 = LN2(X1 + relu(X1 W1^T + b1) W2^T + b2)
        # plain-bf16 backward: the LayerNorm backward / the data-gradient epilogues also leave bf16 twins of their outputs, which the next
        # data-gradient GEMM reads through the direct-to-LDS kernel (igemm_dl.hip) -- the chain dY2 -> dHd -> dX1 -> dY1 -> dO of 5000-row GEMMs
        tw = ENC_TWINS and bf16_twins() and MIRROR is not None
        if isinstance(dX2, tuple):
            pt, pt2, pt3, pKr, pKc = dX2
            dY2 = ln_bwd_raw(pt, *ln2, merge=(pt2, pt3, None, None), bcast=(pKr, pKc, 1.0 / H, 1.0 / W, H, W), twin=tw)
        else:
            dY2 = ln_bwd_raw(dX2.reshape(R, Cc).contiguous(), *ln2, twin=tw)
        dY2, dY2h = dY2 if tw else (dY2, None)
        _wg(dY2, Hd, f.linear2.weight, f.linear2.bias, 0, Cc)
        dHd = linear_dgrad(dY2, f.linear2.weight.detach(), gate=Hd, dy16=dY2h, twin=tw)               # ReLU mask fused in the epilogue
        dHd, dHdh = dHd if tw else (dHd, None)
        _wg(dHd, X1, f.linear1.weight, f.linear1.bias, 0, Hd.shape[1])
        dX1 = linear_dgrad(dHd, f.linear1.weight.detach(), resid=dY2, dy16=dHdh)             # + residual branch
        # ---- attention block: X1 = LN1(X + o Wo^T + bo)
        dY1 = ln_bwd_raw(dX1, Y1, mu1, rs1, layer.norm1.weight.detach(), grad_buffer(layer.norm1.weight), grad_buffer(layer.norm1.bias), twin=tw)
        dY1, dY1h = dY1 if tw else (dY1, None)
        o2d = o.view(R, E)
        _wg(dY1, o2d, att.out_proj.weight, att.out_proj.bias, 0, Cc)
        dO = linear_dgrad(dY1, att.out_proj.weight.detach(), dy16=dY1h).view(N, H * W, E)
        dq_row, dq_col, dk_row, dk_col, dv = rcda_bwd_raw(dO, q_row, q_col, k_row, k_col, v, a_row, a_col, nh, zbuf)
        dq_row2, dq_col2, dv2 = dq_row.view(R, E), dq_col.view(R, E), dv.view(R, E)
        dk_row2, dk_col2 = dk_row.view(N * W, E), dk_col.view(N * H, E)
        _wg(dq_row2, Qr.view(R, Cc), Wip, bip, 0, E)
        _wg(dq_col2, Qc.view(R, Cc), Wip, bip, E, 2 * E)
        _wg(dk_row2, Kr.view(N * W, Cc), Wip, bip, 2 * E, 3 * E)
        _wg(dk_col2, Kc.view(N * H, Cc), Wip, bip, 3 * E, 4 * E)
        _wg(dv2, X.view(R, Cc), Wip, bip, 4 * E, 5 * E)
        # ---- d(src): residual + three projection inputs chained through the dgrad epilogues, then the two key means
        # (the three [R, C] products ride in ONE grouped launch -- 474 workgroups instead of three dependent launches of 158 -- and are
        # summed by the broadcast pass below; tools/step_listing.py: 33 -> ~15 us per layer)
        with gemm_queue():
            t = linear_dgrad(dq_row2, Wi[0:E], resid=dY1)
            t2 = linear_dgrad(dq_col2, Wi[E:2 * E])
            t3 = linear_dgrad(dv2, Wi[4 * E:5 * E])
            dKr = linear_dgrad(dk_row2, Wi[2 * E:3 * E])                           # [N*W, C]
            dKc = linear_dgrad(dk_col2, Wi[3 * E:4 * E])                           # [N*H, C]
            # the same two (tiny) products once more with the accumulated d(posemb) of the layers above in the epilogue: they ride in
            # the same grouped launch, and the accumulation costs no launch of its own
            dKrA = linear_dgrad(dk_row2, Wi[2 * E:3 * E], resid=accR.reshape(N * W, Cc)) if accR is not None else dKr
            dKcA = linear_dgrad(dk_col2, Wi[3 * E:4 * E], resid=accC.reshape(N * H, Cc)) if accC is not None else dKc
        # (defer_out: the sum is left to the LayerNorm backward of the layer below)
        dX = (t, t2, t3, dKr, dKc) if defer_out else bcast_add2_sum(t.view(N, H, W, Cc), t2.view(N, H, W, Cc), t3.view(N, H, W, Cc), dKr, dKc,
                                                                    1.0 / H, 1.0 / W)
        # ---- d(posemb): sum over the broadcast axis BEFORE projecting back (linearity) + the key-mean terms
        sr, sc = hw_reduce(dq_row.view(N, H, W, E), dq_col.view(N, H, W, E), None, None, 1.0, 1.0)
        with gemm_queue():
            dProw = linear_dgrad(sr.view(N * W, E), Wi[0:E], resid=dKrA).view(N, W, Cc)
            dPcol = linear_dgrad(sc.view(N * H, E), Wi[E:2 * E], resid=dKcA).view(N, H, Cc)
        return dX, dProw, dPcol, None, None, None, None


class EncoderStackFn(torch.autograd.Function):
    """ALL encoder layers (A2/models/transformer.py:162-163, 242-279) as ONE autograd node: the per-layer forward / backward of
    EncoderLayerFn, plus what only a whole-stack node can do -- the gradients of the two positional embeddings, which every layer
    consumes, are accumulated across the layers inside GEMM epilogues (no autograd accumulation launches: the stack used to cost 12
    tensor adds), and the parameter gradients of all layers leave in a few grouped launches at the end of the stack's backward."""

    @staticmethod
    def forward(ctx, src, posemb_row, posemb_col, mask_row, mask_col, layers, anchor, taps):
        x = src
        ctxs = []
        for li, layer in enumerate(layers):
            c = _Ctx()
            x = EncoderLayerFn.forward(c, x, posemb_row, posemb_col, mask_row, mask_col, layer, None)
            ctxs.append(c)
            if taps is not None:
                taps[f"enc{li}"] = x.detach()
        ctx.ctxs = ctxs
        ctx.set_materialize_grads(False)      # (an embedding output nobody consumed brings None, not a zero tensor + its fill)
        # the two positional embeddings leave the node again (aliases): the decoder takes THESE, so its gradients with respect to them arrive here
        # as output gradients and seed the accumulation below -- autograd's two accumulation adds of d(posemb) disappear from the main chain
        return x, posemb_row.view_as(posemb_row), posemb_col.view_as(posemb_col)

    @staticmethod
    def backward(ctx, dX, dPR=None, dPC=None):
        accR = dPR.contiguous() if dPR is not None else None
        accC = dPC.contiguous() if dPC is not None else None
        c0 = ctx.ctxs[0]
        N, H, W, Cc, E, nh = c0.dims
        zn = N * H * W * E + N * W * E + N * H * E                  # per layer: dV + both key gradients (rcda_zero_numel)
        zall = zeros_flat(len(ctx.ctxs) * zn, dX.device)      # ONE fill for the whole stack (or none: ops.ZeroArena)
        with wgrad_queue():
            for li in range(len(ctx.ctxs) - 1, -1, -1):
                c = ctx.ctxs[li]
                dX, accR, accC = EncoderLayerFn._backward(c, dX, accR, accC, zall[li * zn:(li + 1) * zn], defer_out=(li > 0 and ENC_DEFER))[:3]
                c.saved_tensors = None
        ctx.ctxs = None
        return dX, accR, accC, None, None, None, None, None


class _Ctx:
    """Stand-in for an autograd ctx when a Function's forward / backward bodies are driven by an enclosing node."""

    def save_for_backward(self, *ts):
        self.saved_tensors = ts

    def set_materialize_grads(self, flag):
        pass


def add2(T, A, B=None):
    """(T + A, T + B) in one pass (B optional)."""
    O1 = torch.empty_like(T)
    O2 = torch.empty_like(T) if B is not None else None
    check(lib().cdetr_add2(ptr(T), ptr(A), ptr(B), ptr(O1), ptr(O2), T.numel(), stream_ptr()), "cdetr_add2")
    return O1, O2


def grad_merge(base, g1, g2=None, acc1=None, acc2=None):
    """out = base + g1 (+ g2); acc1 += g1, acc2 += g2 in place (gradient accumulators of shared inputs)."""
    out = torch.empty_like(base)
    check(lib().cdetr_grad_merge(ptr(base), ptr(g1), ptr(g2), ptr(acc1), ptr(acc2), ptr(out), base.numel(), stream_ptr()),
          "cdetr_grad_merge")
    return out


class DecoderStackFn(torch.autograd.Function):
    """All decoder layers (A2/models/transformer.py:316-409, single feature level) as ONE autograd node.
    forward(tgt, query_pos, query_pos_x, query_pos_y, memory, k_row_mean, k_col_mean, mask_row, mask_col, layers, anchor)
    -> one output [N, L, E] per layer.  The backward is hand-scheduled: every multi-consumer gradient sum is chained
    through a data-gradient epilogue (`resid=`) or one cdetr_grad_merge pass; the gradients of the inputs shared by all
    layers (the three query-position terms, the memory and both key means) accumulate across layers inside the node, so
    autograd never runs an accumulation kernel for them."""

    @staticmethod
    def forward(ctx, tgt, qpos, qx, qy, memory, krm, kcm, mask_row, mask_col, layers, anchor, posemb_row=None, posemb_col=None):
        """krm / kcm: the key means mean_H(memory) + posemb_row, mean_W(memory) + posemb_col -- or None with posemb_row / posemb_col
        given: then they are formed here (one launch) and their backward folds into the memory gradient (one launch) instead of
        four tensor launches each way."""
        N, L, E = tgt.shape
        _, H, W, _ = memory.shape
        M = N * L
        tgt, qpos, qx, qy = tgt.contiguous(), qpos.contiguous(), qx.contiguous(), qy.contiguous()
        memory = memory.contiguous()
        mem2 = memory.view(N * H * W, E)
        ctx.own_means = krm is None
        if ctx.own_means:
            krm, kcm = hw_reduce(memory, memory, posemb_row.contiguous(), posemb_col.contiguous(), 1.0 / H, 1.0 / W)
        krm2, kcm2 = krm.contiguous().view(N * W, E), kcm.contiguous().view(N * H, E)
        outs = []
        saved = []
        x = tgt.view(M, E)
        # the key / value projections of the encoder memory do not depend on the decoder state: all layers' up front, grouped
        mem_side = []
        with gemm_queue():
            for layer in layers:
                ca = layer.cross_attn
                Wc, bc = ca.in_proj_weight.detach(), ca.in_proj_bias.detach()
                mem_side.append((linear_fwd(krm2, Wc[2 * E:3 * E], bc[2 * E:3 * E]).view(N, W, E),
                                 linear_fwd(kcm2, Wc[3 * E:4 * E], bc[3 * E:4 * E]).view(N, H, E),
                                 linear_fwd(mem2, Wc[4 * E:5 * E], bc[4 * E:5 * E]).view(N, H, W, E)))
        a1_next = None
        for li, layer in enumerate(layers):
            sa, ca, f = layer.self_attn, layer.cross_attn, layer.ffn
            nh = sa.num_heads
            Ws, bs = sa.in_proj_weight.detach(), sa.in_proj_bias.detach()
            Wc, bc = ca.in_proj_weight.detach(), ca.in_proj_bias.detach()
            if a1_next is None:
                a1, _ = add2(x, qpos.view(M, E))
            else:
                a1 = a1_next          # written by the previous layer's last LayerNorm
            with gemm_queue():
                qk = linear_fwd(a1, Ws[0:2 * E], bs[0:2 * E])
                vs = linear_fwd(x, Ws[2 * E:3 * E], bs[2 * E:3 * E])
            o1, lse = mha_fwd_raw(qk.view(N, L, 2 * E), vs.view(N, L, E), nh)
            Y2 = linear_fwd(o1.view(M, E), sa.out_proj.weight.detach(), sa.out_proj.bias.detach(), resid=x)
            T1, mu2, rs2, qr_in, qc_in = ln_fwd_add_raw(Y2, layer.norm2.weight.detach(), layer.norm2.bias.detach(), layer.norm2.eps,
                                                        qx.view(M, E), qy.view(M, E))
            with gemm_queue():
                q_row = linear_fwd(qr_in, Wc[0:E], bc[0:E]).view(N, L, E)
                q_col = linear_fwd(qc_in, Wc[E:2 * E], bc[E:2 * E]).view(N, L, E)
            k_row, k_col, v = mem_side[li]
            o2, a_row, a_col = rcda_fwd_raw(q_row, q_col, k_row, k_col, v, mask_row, mask_col, ca.num_heads)
            Y1 = linear_fwd(o2.view(M, E), ca.out_proj.weight.detach(), ca.out_proj.bias.detach(), resid=T1)
            T2, mu1, rs1 = ln_fwd_raw(Y1, layer.norm1.weight.detach(), layer.norm1.bias.detach(), layer.norm1.eps)
            Hd = linear_fwd(T2, f.linear1.weight.detach(), f.linear1.bias.detach(), relu=True)
            Y3 = linear_fwd(Hd, f.linear2.weight.detach(), f.linear2.bias.detach(), resid=T2)
            if li + 1 < len(layers):      # the next layer's tgt + query_pos comes out of this LayerNorm's pass
                out, mu3, rs3, a1_next, _ = ln_fwd_add_raw(Y3, f.norm2.weight.detach(), f.norm2.bias.detach(), f.norm2.eps, qpos.view(M, E))
            else:
                out, mu3, rs3 = ln_fwd_raw(Y3, f.norm2.weight.detach(), f.norm2.bias.detach(), f.norm2.eps)
            outs.append(out.view(N, L, E))
            saved.append((x, a1, qk, vs, o1, lse, Y2, mu2, rs2, T1, qr_in, qc_in, q_row, q_col, k_row, k_col, v, a_row, a_col, o2,
                          Y1, mu1, rs1, T2, Hd, Y3, mu3, rs3))
            x = out
        ctx.layers, ctx.saved, ctx.dims = layers, saved, (N, L, E, H, W)
        ctx.shared = (mem2, krm2, kcm2)
        ctx.set_materialize_grads(False)      # layers whose output feeds no loss term get None, not a zero tensor
        return tuple(outs)

    @staticmethod
    def backward(ctx, *d_outs):
        with wgrad_queue():          # ~13 small parameter gradients per layer, all layers: submitted in groups at the end
            return DecoderStackFn._backward(ctx, *d_outs)

    @staticmethod
    def _backward(ctx, *d_outs):
        layers, saved = ctx.layers, ctx.saved
        N, L, E, H, W = ctx.dims
        mem2, krm2, kcm2 = ctx.shared
        M = N * L
        dev = mem2.device
        zn = N * H * W * E + N * W * E + N * H * E                              # per layer: dV + both key gradients (rcda_zero_numel)
        zall = zeros_flat(3 * M * E + len(layers) * zn, dev)     # ONE fill for the whole stack's accumulators (or none: ops.ZeroArena)
        acc = zall[:3 * M * E].view(3, M, E)                                    # d(query_pos), d(query_pos_x), d(query_pos_y)
        acc_p, acc_x, acc_y = acc[0], acc[1], acc[2]
        dMem = dKrm = dKcm = None
        pend_t = pend_g = None                                                  # the layer above's d(x) = pend_t + pend_g, summed by its consumer
        for li in range(len(layers) - 1, -1, -1):
            layer = layers[li]
            sa, ca, f = layer.self_attn, layer.cross_attn, layer.ffn
            (x, a1, qk, vs, o1, lse, Y2, mu2, rs2, T1, qr_in, qc_in, q_row, q_col, k_row, k_col, v, a_row, a_col, o2,
             Y1, mu1, rs1, T2, Hd, Y3, mu3, rs3) = saved[li]
            saved[li] = None
            g_out = d_outs[li].reshape(M, E).contiguous() if d_outs[li] is not None else None
            Ws, Wc = sa.in_proj_weight.detach(), ca.in_proj_weight.detach()
            # ---- FFN: out = LN3(T2 + relu(T2 W1^T + b1) W2^T + b2).  The gradient of this layer's output = the layer above's d(x)
            # (residual + v-path `pend_t`, plus the q/k-path `pend_g` whose value also accumulates into d(qpos)) + this layer's own output
            # gradient (aux losses / the last layer): summed inside the LayerNorm backward that consumes it
            ln3 = (Y3, mu3, rs3, f.norm2.weight.detach(), grad_buffer(f.norm2.weight), grad_buffer(f.norm2.bias))
            if pend_t is None:
                if g_out is None:       # this layer's output feeds nothing (cannot happen for the last layer)
                    continue
                dY3 = ln_bwd_raw(g_out, *ln3)
            else:
                dY3 = ln_bwd_raw(pend_t, *ln3, merge=(pend_g, g_out, acc_p, None))
            _wg(dY3, Hd, f.linear2.weight, f.linear2.bias, 0, E)
            dHd = linear_dgrad(dY3, f.linear2.weight.detach(), gate=Hd)
            _wg(dHd, T2, f.linear1.weight, f.linear1.bias, 0, Hd.shape[1])
            dT2 = linear_dgrad(dHd, f.linear1.weight.detach(), resid=dY3)
            # ---- cross attention: T2 = LN1(T1 + rcda(...) Wo^T + bo)
            dY1 = ln_bwd_raw(dT2, Y1, mu1, rs1, layer.norm1.weight.detach(), grad_buffer(layer.norm1.weight), grad_buffer(layer.norm1.bias))
            _wg(dY1, o2.view(M, E), ca.out_proj.weight, ca.out_proj.bias, 0, E)
            dO2 = linear_dgrad(dY1, ca.out_proj.weight.detach()).view(N, L, E)
            dq_row, dq_col, dk_row, dk_col, dv = rcda_bwd_raw(dO2, q_row, q_col, k_row, k_col, v, a_row, a_col, ca.num_heads,
                                                              zall[3 * M * E + li * zn:3 * M * E + (li + 1) * zn])
            dq_row2, dq_col2 = dq_row.view(M, E), dq_col.view(M, E)
            dk_row2, dk_col2, dv2 = dk_row.view(N * W, E), dk_col.view(N * H, E), dv.view(N * H * W, E)
            Wcp, bcp = ca.in_proj_weight, ca.in_proj_bias
            _wg(dq_row2, qr_in, Wcp, bcp, 0, E)
            _wg(dq_col2, qc_in, Wcp, bcp, E, 2 * E)
            _wg(dk_row2, krm2, Wcp, bcp, 2 * E, 3 * E)
            _wg(dk_col2, kcm2, Wcp, bcp, 3 * E, 4 * E)
            _wg(dv2, mem2, Wcp, bcp, 4 * E, 5 * E)
            with gemm_queue():         # the chained accumulators read the PREVIOUS layer's sums, written before this block
                gx = linear_dgrad(dq_row2, Wc[0:E])
                gy = linear_dgrad(dq_col2, Wc[E:2 * E])
                dKrm = linear_dgrad(dk_row2, Wc[2 * E:3 * E], resid=dKrm)      # shared inputs: chained across layers
                dKcm = linear_dgrad(dk_col2, Wc[3 * E:4 * E], resid=dKcm)
                dMem = linear_dgrad(dv2, Wc[4 * E:5 * E], resid=dMem)
            # d(T1) = dY1 + both query projections, d(qx) += gx, d(qy) += gy: formed inside the LayerNorm backward that consumes it
            # ---- self attention: T1 = LN2(x + mha((x + qpos) Wqk, x Wv) Wo^T + bo)
            dY2 = ln_bwd_raw(dY1, Y2, mu2, rs2, layer.norm2.weight.detach(), grad_buffer(layer.norm2.weight), grad_buffer(layer.norm2.bias),
                             merge=(gx, gy, acc_x, acc_y))
            _wg(dY2, o1.view(M, E), sa.out_proj.weight, sa.out_proj.bias, 0, E)
            dO1 = linear_dgrad(dY2, sa.out_proj.weight.detach()).view(N, L, E)
            dqk, dvs = mha_bwd_raw(qk.view(N, L, 2 * E), vs.view(N, L, E), o1, dO1, lse, sa.num_heads)
            Wsp, bsp = sa.in_proj_weight, sa.in_proj_bias
            _wg(dqk.view(M, 2 * E), a1, Wsp, bsp, 0, 2 * E)
            _wg(dvs.view(M, E), x, Wsp, bsp, 2 * E, 3 * E)
            with gemm_queue():
                ga1 = linear_dgrad(dqk.view(M, 2 * E), Ws[0:2 * E])
                t = linear_dgrad(dvs.view(M, E), Ws[2 * E:3 * E], resid=dY2)
            pend_t, pend_g = t, ga1                                            # d(x) = t + ga1 and d(qpos) += ga1: formed by the consumer
        dx = grad_merge(pend_t, pend_g, None, acc_p, None)                     # the first layer's input gradient leaves the node: merged for real
        if ctx.own_means:        # memory also fed the two key means: broadcast their gradients back in the same pass
            dMem = bcast_add2(dMem.view(N, H, W, E), dKrm, dKcm, 1.0 / H, 1.0 / W)
            return (dx.view(N, L, E), acc_p.view(N, L, E), acc_x.view(N, L, E), acc_y.view(N, L, E), dMem, None, None, None, None, None,
                    None, dKrm.view(N, W, E), dKcm.view(N, H, E))
        return (dx.view(N, L, E), acc_p.view(N, L, E), acc_x.view(N, L, E), acc_y.view(N, L, E), dMem.view(N, H, W, E),
                dKrm.view(N, W, E), dKcm.view(N, H, E), None, None, None, None, None, None)


# ----------------------------------------------------------------------------------------------------- matcher
class MatchPlan:
    """Description of one batch of targets for the matcher / criterion kernels: per-image target counts as a DEVICE offset table
    (`tgt_off`, read by every kernel), plus the static sizes the launches are dimensioned by (`Mmax`, `nc_max`, `cost_off`).

    `MatchPlan(sizes, Q, device)`: exact plan of one tuple of counts (tables built once per tuple).
    `MatchPlan.capacity(B, Q, Tcap, device)`: ONE plan for every batch of B images with at most `Tcap` targets each -- launch
    dimensions, cost-matrix slots (`b * Q * Tcap`) and index strides come from the capacity, the actual counts live only in
    `tgt_off` and are rewritten by `set_counts()` before each step: a captured HIP graph serves any target counts."""

    def __init__(self, sizes, Q, device, _capacity=None):
        self.Q = Q
        self.device = device
        self.Tcap = _capacity
        if _capacity is None:
            self._set_host(sizes)
            coff = [0]
            for s in self.sizes:
                coff.append(coff[-1] + Q * s)
            self.tgt_off = torch.tensor(self.tgt_off_host, dtype=torch.int32, device=device)
            self.sizes_f = torch.tensor([float(s) for s in self.sizes], dtype=torch.float32, device=device)
            self.Mmax = max(max(self.M), 1)
            self.nc_max = max([Q] + self.sizes)
        else:
            B = int(sizes)
            self.B = B
            coff = [b * Q * _capacity for b in range(B + 1)]
            self.tgt_off = torch.zeros(B + 1, dtype=torch.int32, device=device)
            self.sizes_f = torch.zeros(B, dtype=torch.float32, device=device)
            self.Mmax = max(min(Q, _capacity), 1)
            self.nc_max = max(Q, _capacity)
            self._set_host([0] * B)
        self.cost_off = torch.tensor(coff[:-1], dtype=torch.int64, device=device)
        self.cost_numel = max(coff[-1], 1)
        self.cost_off_host = coff

    @classmethod
    def capacity(cls, B, Q, Tcap, device):
        return cls(B, Q, device, _capacity=int(Tcap))

    def _set_host(self, sizes):
        self.sizes = [int(s) for s in sizes]
        self.B = len(self.sizes)
        off = [0]
        for s in self.sizes:
            off.append(off[-1] + s)
        self.tgt_off_host = off
        self.M = [min(self.Q, s) for s in self.sizes]

    def set_counts(self, sizes):
        """Capacity plan: the counts of the batch about to run (stream-ordered copies into the device tables)."""
        assert self.Tcap is not None and len(sizes) == self.B and max(sizes, default=0) <= self.Tcap
        self._set_host(sizes)
        self.tgt_off.copy_(torch.tensor(self.tgt_off_host, dtype=torch.int32))
        self.sizes_f.copy_(torch.tensor([float(s) for s in self.sizes], dtype=torch.float32))


class PackedTargets:
    """The targets of one batch in the form the device matcher / criterion consume: boxes [ΣT (or B*Tcap), 4] fp32 and labels
    int64 concatenated in image order, and the MatchPlan holding the per-image offsets.  `SetCriterion.forward` accepts it in
    place of the reference's list of dicts; the graph-cached train step keeps one per captured graph (fixed addresses)."""

    def __init__(self, boxes, labels, plan):
        self.boxes, self.labels, self.plan = boxes, labels, plan

    @classmethod
    def with_capacity(cls, B, Q, Tcap, device):
        return cls(torch.zeros((B * Tcap, 4), dtype=torch.float32, device=device), torch.zeros(B * Tcap, dtype=torch.int64, device=device),
                   MatchPlan.capacity(B, Q, Tcap, device))

    def load(self, targets):
        """Copy a list of {"boxes", "labels"} dicts (the reference's target format) into the fixed buffers."""
        sizes = [int(t["boxes"].shape[0]) for t in targets]
        n = sum(sizes)
        if n:
            self.boxes[:n].copy_(torch.cat([t["boxes"].reshape(-1, 4) for t in targets]))
            self.labels[:n].copy_(torch.cat([t["labels"].reshape(-1) for t in targets]))
        self.plan.set_counts(sizes)
        return n


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
    idx = torch.empty((2, plan.B, plan.Mmax), dtype=torch.int64, device=dev)     # cleared by the kernels themselves
    idx_i, idx_j = idx[0], idx[1]
    status = torch.empty(plan.B, dtype=torch.int32, device=dev)
    check(lib().cdetr_lsap(ptr(cost), ptr(plan.cost_off), ptr(plan.tgt_off), plan.B, plan.Q, plan.nc_max, plan.Mmax,
                           ptr(idx_i), ptr(idx_j), ptr(status), stream_ptr()), "cdetr_lsap")
    return idx_i, idx_j, status


class CriterionFn(torch.autograd.Function):
    """SetCriterion's six scalars in one launch (cdetr_criterion_fwd); returns (vec, total) with
    vec = [loss_ce, class_error, cardinality_error, loss_bbox, loss_giou, loss_variance] and total = sum_k w6[k] vec[k] (the weighted
    loss of A2/engine.py:37, formed by the same launch; None without `w6`).  The forward kernel already leaves the gradient of every
    loss w.r.t. logits / boxes / vars, so backward is one scaled sum (cdetr_criterion_bwd) of whichever of (vec, total) was used."""

    @staticmethod
    def forward(ctx, logits, boxes, pvars, tgt_boxes, tgt_labels, plan, idx_i, idx_j, num_boxes, num_classes, alpha, w6=None):
        B, Q, Cc = logits.shape
        logits, boxes, pvars = logits.contiguous(), boxes.contiguous(), pvars.contiguous()
        dev = logits.device
        losses = torch.empty(7, device=dev, dtype=torch.float32)
        g = torch.empty(B * Q * (Cc + 14), device=dev, dtype=torch.float32)
        n = B * Q
        g_logits, g_l1, g_giou, g_vb, g_vars = (g[:n * Cc], g[n * Cc:n * (Cc + 4)], g[n * (Cc + 4):n * (Cc + 8)],
                                                g[n * (Cc + 8):n * (Cc + 12)], g[n * (Cc + 12):])
        d = CriterionDesc()
        d.B, d.Q, d.C, d.num_classes, d.Mmax, d.alpha = B, Q, Cc, num_classes, plan.Mmax, alpha
        d.logits, d.boxes, d.vars = ptr(logits), ptr(boxes), ptr(pvars)
        d.tgt_boxes = ptr(tgt_boxes) if tgt_boxes.numel() else ptr(losses)
        d.tgt_labels = ptr(tgt_labels) if tgt_labels.numel() else ptr(losses)
        d.tgt_off, d.idx_i, d.idx_j, d.num_boxes, d.losses = ptr(plan.tgt_off), ptr(idx_i), ptr(idx_j), ptr(num_boxes), ptr(losses)
        d.g_logits, d.g_l1, d.g_giou, d.g_var_box, d.g_vars = ptr(g_logits), ptr(g_l1), ptr(g_giou), ptr(g_vb), ptr(g_vars)
        d.loss_weights = ptr(w6)
        check(lib().cdetr_criterion_fwd(C.byref(d), stream_ptr()), "cdetr_criterion_fwd")
        ctx.g, ctx.dims, ctx.w6 = (g_logits, g_l1, g_giou, g_vb, g_vars), (B, Q, Cc), w6
        ctx.set_materialize_grads(False)
        vec = losses[:6]
        if w6 is None:
            return vec, None
        return vec, losses[6]

    @staticmethod
    def backward(ctx, g6, gt):
        B, Q, Cc = ctx.dims
        g_logits, g_l1, g_giou, g_vb, g_vars = ctx.g
        dev = g_logits.device
        d_logits = torch.empty((B, Q, Cc), device=dev, dtype=torch.float32)
        d_boxes = torch.empty((B, Q, 4), device=dev, dtype=torch.float32)
        d_vars = torch.empty((B, Q, 2), device=dev, dtype=torch.float32)
        g6 = g6.contiguous() if g6 is not None else None
        gt = gt.reshape(1).contiguous() if gt is not None else None
        if g6 is None and gt is None:
            return (None,) * 12
        check(lib().cdetr_criterion_bwd(ptr(g6), ptr(gt), ptr(ctx.w6), ptr(g_logits), ptr(g_l1), ptr(g_giou), ptr(g_vb), ptr(g_vars),
                                        ptr(d_logits), ptr(d_boxes), ptr(d_vars), B * Q, Cc, stream_ptr()), "cdetr_criterion_bwd")
        return (d_logits, d_boxes, d_vars) + (None,) * 9


class SineEmbedFn(torch.autograd.Function):
    """pos2posemb1d / pos2posemb2d (A2/models/transformer.py:474-494) in one launch per coordinate.
    pos [..., ncoord] (ncoord 1 or 2); two coordinates give the reference's (y, x) concatenation of two nfeat-wide halves."""

    @staticmethod
    def forward(ctx, pos, nfeat, temperature, two_d):
        p = pos.contiguous().to(torch.float32)
        lead = p.shape[:-1] if two_d else p.shape
        rows = 1
        for d_ in lead:
            rows *= int(d_)
        width = 2 * nfeat if two_d else nfeat
        out = torch.empty(tuple(lead) + (width,), device=p.device, dtype=torch.float32)
        if two_d:      # first half from pos[..., 1] (y), second half from pos[..., 0] (x)
            check(lib().cdetr_sine_embed(p.data_ptr() + 4, 2, out.data_ptr(), width, rows, nfeat, temperature, stream_ptr()), "cdetr_sine_embed")
            check(lib().cdetr_sine_embed(p.data_ptr(), 2, out.data_ptr() + 4 * nfeat, width, rows, nfeat, temperature, stream_ptr()), "cdetr_sine_embed")
        else:
            check(lib().cdetr_sine_embed(ptr(p), 1, ptr(out), width, rows, nfeat, temperature, stream_ptr()), "cdetr_sine_embed")
        ctx.save_for_backward(p)
        ctx.cfg = (rows, nfeat, temperature, two_d, width, pos.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        (p,) = ctx.saved_tensors
        rows, nfeat, temperature, two_d, width, shape = ctx.cfg
        dout = dout.contiguous()
        dp = torch.empty_like(p)
        if two_d:
            check(lib().cdetr_sine_embed_bwd(p.data_ptr() + 4, 2, dout.data_ptr(), width, dp.data_ptr() + 4, 2, rows, nfeat, temperature, 0,
                                             stream_ptr()), "cdetr_sine_embed_bwd")
            check(lib().cdetr_sine_embed_bwd(p.data_ptr(), 2, dout.data_ptr() + 4 * nfeat, width, dp.data_ptr(), 2, rows, nfeat, temperature, 0,
                                             stream_ptr()), "cdetr_sine_embed_bwd")
        else:
            check(lib().cdetr_sine_embed_bwd(ptr(p), 1, ptr(dout), width, ptr(dp), 1, rows, nfeat, temperature, 0, stream_ptr()),
                  "cdetr_sine_embed_bwd")
        return dp.view(shape), None, None, None


def sine_embed(pos, nfeat, temperature=10000.0, two_d=False):
    return SineEmbedFn.apply(pos, nfeat, float(temperature), two_d)


class SineEmbedXYFn(torch.autograd.Function):
    """(pos2posemb1d(p[..., 0]), pos2posemb1d(p[..., 1])) for points p [..., 2] (A2/models/transformer.py:378-379) read in place with
    an element stride of 2: no `p[..., 0].contiguous()` copies forward, and the backward writes both coordinates' gradients
    straight into one [..., 2] tensor (no select_backward zero-fill + copy + add)."""

    @staticmethod
    def forward(ctx, pos, nfeat, temperature):
        p = pos.contiguous().to(torch.float32)
        rows = p.numel() // 2
        ex = torch.empty(tuple(p.shape[:-1]) + (nfeat,), device=p.device, dtype=torch.float32)
        ey = torch.empty_like(ex)
        check(lib().cdetr_sine_embed(p.data_ptr(), 2, ptr(ex), nfeat, rows, nfeat, temperature, stream_ptr()), "cdetr_sine_embed")
        check(lib().cdetr_sine_embed(p.data_ptr() + 4, 2, ptr(ey), nfeat, rows, nfeat, temperature, stream_ptr()), "cdetr_sine_embed")
        ctx.save_for_backward(p)
        ctx.cfg = (rows, nfeat, temperature, pos.shape)
        return ex, ey

    @staticmethod
    def backward(ctx, dex, dey):
        (p,) = ctx.saved_tensors
        rows, nfeat, temperature, shape = ctx.cfg
        dp = torch.empty_like(p) if (dex is not None and dey is not None) else torch.zeros_like(p)
        if dex is not None:
            check(lib().cdetr_sine_embed_bwd(p.data_ptr(), 2, ptr(dex.contiguous()), nfeat, dp.data_ptr(), 2, rows, nfeat, temperature, 0,
                                             stream_ptr()), "cdetr_sine_embed_bwd")
        if dey is not None:
            check(lib().cdetr_sine_embed_bwd(p.data_ptr() + 4, 2, ptr(dey.contiguous()), nfeat, dp.data_ptr() + 4, 2, rows, nfeat, temperature, 0,
                                             stream_ptr()), "cdetr_sine_embed_bwd")
        return dp.view(shape), None, None


def sine_embed_xy(points, nfeat, temperature=10000.0):
    return SineEmbedXYFn.apply(points, nfeat, float(temperature))
