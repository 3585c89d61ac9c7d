"""ctypes binding of libcdetr_hip.so (include/cdetr_hip.h).  No torch types cross the boundary: raw device
pointers (tensor.data_ptr()), sizes and the current hipStream_t.

The product path has NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CDETR_LIB") or os.path.join(_HERE, "lib", "libcdetr_hip.so")      # CDETR_LIB: experiment builds (tools/)

ROWS_DENSE, ROWS_CONV_FWD, ROWS_CONV_DGRAD = 0, 1, 2
_p = C.c_void_p


class ConvGeom(C.Structure):
    _fields_ = [("mode", C.c_int32), ("Ha", C.c_int32), ("Wa", C.c_int32), ("Hc", C.c_int32), ("Wc", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dil", C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("taps", C.c_int32), ("batch", C.c_int32),
                ("b_layout", C.c_int32), ("relu", C.c_int32), ("precision", C.c_int32), ("out_scale", C.c_float),
                ("A", _p), ("lda", C.c_int64), ("sA", C.c_int64),
                ("B", _p), ("ldb", C.c_int64), ("sB", C.c_int64),
                ("C", _p), ("ldc", C.c_int64), ("sC", C.c_int64),
                ("w_scale", _p), ("bias", _p), ("resid", _p), ("ldr", C.c_int64), ("gate", _p), ("ldg", C.c_int64),
                ("g", ConvGeom), ("B_split", _p), ("batch_inner", C.c_int32), ("flags", C.c_int32),
                ("sA2", C.c_int64), ("sB2", C.c_int64), ("sC2", C.c_int64), ("A16", _p), ("C16", _p),
                ("splitk_ws", _p), ("splitk_ws_bytes", C.c_int64), ("A16lo", _p), ("B16", _p), ("gate16", _p), ("C16lo", _p)]


class WgradDesc(C.Structure):
    _fields_ = [("P", C.c_int32), ("Nout", C.c_int32), ("Cin", C.c_int32), ("taps", C.c_int32), ("batch", C.c_int32), ("precision", C.c_int32),
                ("dY", _p), ("ldy", C.c_int64), ("sY", C.c_int64),
                ("X", _p), ("ldx", C.c_int64), ("sX", C.c_int64),
                ("dW", _p), ("ldw", C.c_int64), ("sW", C.c_int64),
                ("w_scale", _p), ("dbias", _p), ("g", ConvGeom), ("batch_inner", C.c_int32), ("wg_target", C.c_int32),
                ("sY2", C.c_int64), ("sX2", C.c_int64), ("sW2", C.c_int64), ("dY16", _p), ("X16", _p)]


class RcdaFwdDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("nh", C.c_int32),
                ("precision", C.c_int32), ("scale", C.c_float), ("q_row", _p), ("q_col", _p), ("k_row", _p), ("k_col", _p), ("v", _p),
                ("mask_row", _p), ("mask_col", _p), ("out", _p), ("a_row", _p), ("a_col", _p), ("ws", _p), ("ws_bytes", C.c_int64)]


class RcdaBwdDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("nh", C.c_int32),
                ("precision", C.c_int32), ("scale", C.c_float), ("d_out", _p), ("a_row", _p), ("a_col", _p), ("v", _p),
                ("ds_row", _p), ("ds_col", _p), ("d_v", _p), ("k_row", _p), ("k_col", _p), ("dq_row", _p), ("dq_col", _p),
                ("q_row", _p), ("q_col", _p), ("dk_row", _p), ("dk_col", _p)]


class CriterionDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("Q", C.c_int32), ("C", C.c_int32), ("num_classes", C.c_int32), ("Mmax", C.c_int32), ("alpha", C.c_float),
                ("logits", _p), ("boxes", _p), ("vars", _p), ("tgt_boxes", _p), ("tgt_labels", _p), ("tgt_off", _p), ("idx_i", _p),
                ("idx_j", _p), ("num_boxes", _p), ("losses", _p), ("g_logits", _p), ("g_l1", _p), ("g_giou", _p), ("g_var_box", _p),
                ("g_vars", _p), ("loss_weights", _p)]


class MirrorItem(C.Structure):
    _fields_ = [("src", _p), ("dst", _p), ("dst_split", _p), ("scale", _p), ("R", C.c_int32), ("C", C.c_int32), ("taps", C.c_int32),
                ("tile0", C.c_int32), ("transpose", C.c_int32), ("pad_", C.c_int32), ("dst_hi", _p)]


EXPORTS = ["cdetr_gemm", "cdetr_gemm_dl", "cdetr_gemm_group", "cdetr_wgrad", "cdetr_wgrad_group", "cdetr_colsum", "cdetr_sumsq", "cdetr_adamw_step", "cdetr_adamw_step2", "cdetr_relu_mask", "cdetr_relu_mask2", "cdetr_layernorm_fwd", "cdetr_layernorm_fwd_add", "cdetr_layernorm_bwd", "cdetr_layernorm_bwd_merge", "cdetr_groupnorm_fwd", "cdetr_groupnorm_bwd", "cdetr_groupnorm_fwd_ws", "cdetr_groupnorm_bwd_ws", "cdetr_posadd2",
           "cdetr_hw_reduce", "cdetr_posadd2_hw_reduce", "cdetr_bcast_add2", "cdetr_bcast_add2_sum", "cdetr_add2", "cdetr_grad_merge", "cdetr_sine_embed", "cdetr_sine_embed_bwd", "cdetr_maxpool3x3s2", "cdetr_maxpool3x3s2_split", "cdetr_weight_mirror", "cdetr_weight_images", "cdetr_rcda_fwd", "cdetr_rcda_bwd", "cdetr_mha_fwd", "cdetr_mha_bwd",
           "cdetr_mask_prep", "cdetr_stem_pack", "cdetr_exemplar_fwd", "cdetr_exemplar_bwd", "cdetr_aggr_weight_fwd", "cdetr_aggr_weight_bwd",
           "cdetr_box_head_fwd", "cdetr_box_head_bwd", "cdetr_match_cost", "cdetr_lsap", "cdetr_criterion_fwd", "cdetr_criterion_bwd", "cdetr_last_error", "cdetr_abi_version", "cdetr_delay", "cdetr_flag_signal", "cdetr_flag_wait"]

_lib = None


def lib():
    """Load (once) and return the library; raises if it is missing -- there is no CPU/eager fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m counting_detr_amd.build` "
                               "(the Counting-DETR MI355X path has no fallback implementation)")
        L = C.CDLL(LIB_PATH)
        L.cdetr_last_error.restype = C.c_char_p
        L.cdetr_abi_version.restype = C.c_int
        L.cdetr_gemm_dl.restype = C.c_int
        L.cdetr_gemm_dl.argtypes = [_p, C.c_int32, C.c_int32, _p]
        for name in ("cdetr_gemm", "cdetr_wgrad", "cdetr_rcda_fwd", "cdetr_rcda_bwd"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [_p, _p]
        L.cdetr_delay.restype = C.c_int
        L.cdetr_delay.argtypes = [C.c_int32, _p]
        L.cdetr_flag_signal.restype = C.c_int
        L.cdetr_flag_signal.argtypes = [_p, _p]
        L.cdetr_flag_wait.restype = C.c_int
        L.cdetr_flag_wait.argtypes = [_p, _p, C.c_int32, C.c_int32, _p]
        L.cdetr_gemm_group.restype = C.c_int
        L.cdetr_gemm_group.argtypes = [_p, C.c_int32, _p]
        L.cdetr_wgrad_group.restype = C.c_int
        L.cdetr_wgrad_group.argtypes = [_p, C.c_int32, _p]
        L.cdetr_colsum.restype = C.c_int
        L.cdetr_colsum.argtypes = [_p, C.c_int64, C.c_int32, C.c_int32, _p, _p]
        L.cdetr_sumsq.restype = C.c_int
        L.cdetr_sumsq.argtypes = [_p, C.c_int64, _p, _p, _p]
        L.cdetr_adamw_step.restype = C.c_int
        L.cdetr_adamw_step.argtypes = [_p, _p, _p, _p, _p, C.c_int64, _p, _p] + [C.c_float] * 6 + [_p]
        L.cdetr_adamw_step2.restype = C.c_int
        L.cdetr_adamw_step2.argtypes = [_p] * 5 + [C.c_float, C.c_float, C.c_int64, C.c_int64, _p, _p] + [C.c_float] * 6 + [_p]
        L.cdetr_relu_mask.restype = C.c_int
        L.cdetr_relu_mask.argtypes = [_p, _p, _p, C.c_int64, C.c_float, _p]
        L.cdetr_relu_mask2.restype = C.c_int
        L.cdetr_relu_mask2.argtypes = [_p, _p, _p, _p, C.c_int64, C.c_float, _p]
        L.cdetr_layernorm_fwd.restype = C.c_int
        L.cdetr_layernorm_fwd.argtypes = [_p] * 6 + [C.c_int32, C.c_int32, C.c_float, _p]
        L.cdetr_layernorm_fwd_add.restype = C.c_int
        L.cdetr_layernorm_fwd_add.argtypes = [_p] * 10 + [C.c_int32, C.c_int32, C.c_float, _p]
        L.cdetr_layernorm_bwd.restype = C.c_int
        L.cdetr_layernorm_bwd.argtypes = [_p] * 9 + [C.c_int32, C.c_int32, _p, _p]
        L.cdetr_layernorm_bwd_merge.restype = C.c_int
        L.cdetr_layernorm_bwd_merge.argtypes = [_p] * 7 + [C.c_float, C.c_float, C.c_int32, C.c_int32] + [_p] * 8 + [C.c_int32, C.c_int32, _p, _p]
        L.cdetr_groupnorm_fwd.restype = C.c_int
        L.cdetr_groupnorm_fwd.argtypes = [_p] * 6 + [C.c_int32] * 4 + [C.c_float, _p]
        L.cdetr_groupnorm_bwd.restype = C.c_int
        L.cdetr_groupnorm_bwd.argtypes = [_p] * 8 + [C.c_int32] * 4 + [_p]
        L.cdetr_groupnorm_fwd_ws.restype = C.c_int
        L.cdetr_groupnorm_fwd_ws.argtypes = [_p] * 6 + [C.c_int32] * 4 + [C.c_float, _p, C.c_int64, _p]
        L.cdetr_groupnorm_bwd_ws.restype = C.c_int
        L.cdetr_groupnorm_bwd_ws.argtypes = [_p] * 8 + [C.c_int32] * 4 + [_p, C.c_int64, _p]
        L.cdetr_posadd2.restype = C.c_int
        L.cdetr_posadd2.argtypes = [_p] * 5 + [C.c_int32] * 4 + [_p]
        L.cdetr_posadd2_hw_reduce.restype = C.c_int
        L.cdetr_posadd2_hw_reduce.argtypes = [_p] * 7 + [C.c_int32] * 4 + [C.c_float, C.c_float, _p]
        L.cdetr_hw_reduce.restype = C.c_int
        L.cdetr_hw_reduce.argtypes = [_p] * 6 + [C.c_int32] * 4 + [C.c_float, C.c_float, _p]
        L.cdetr_criterion_fwd.restype = C.c_int
        L.cdetr_criterion_fwd.argtypes = [_p, _p]
        L.cdetr_criterion_bwd.restype = C.c_int
        L.cdetr_criterion_bwd.argtypes = [_p] * 11 + [C.c_int32, C.c_int32, _p]
        L.cdetr_sine_embed.restype = C.c_int
        L.cdetr_sine_embed.argtypes = [_p, C.c_int32, _p, C.c_int64, C.c_int32, C.c_int32, C.c_float, _p]
        L.cdetr_sine_embed_bwd.restype = C.c_int
        L.cdetr_sine_embed_bwd.argtypes = [_p, C.c_int32, _p, C.c_int64, _p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _p]
        L.cdetr_add2.restype = C.c_int
        L.cdetr_add2.argtypes = [_p] * 5 + [C.c_int64, _p]
        L.cdetr_grad_merge.restype = C.c_int
        L.cdetr_grad_merge.argtypes = [_p] * 6 + [C.c_int64, _p]
        L.cdetr_weight_images.restype = C.c_int
        L.cdetr_weight_images.argtypes = [_p, C.c_int32, C.c_int32, _p]
        L.cdetr_weight_mirror.restype = C.c_int
        L.cdetr_weight_mirror.argtypes = [_p, C.c_int32, C.c_int32, _p]
        L.cdetr_bcast_add2.restype = C.c_int
        L.cdetr_bcast_add2.argtypes = [_p] * 4 + [C.c_int32] * 4 + [C.c_float, C.c_float, _p]
        L.cdetr_bcast_add2_sum.restype = C.c_int
        L.cdetr_bcast_add2_sum.argtypes = [_p] * 6 + [C.c_int32] * 4 + [C.c_float, C.c_float, _p]
        L.cdetr_maxpool3x3s2.restype = C.c_int
        L.cdetr_maxpool3x3s2.argtypes = [_p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p]
        L.cdetr_maxpool3x3s2_split.restype = C.c_int
        L.cdetr_maxpool3x3s2_split.argtypes = [_p, _p, _p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p]
        L.cdetr_mha_fwd.restype = C.c_int
        L.cdetr_mha_fwd.argtypes = [_p] * 4 + [C.c_int32] * 3 + [C.c_float, C.c_int32, _p]
        L.cdetr_mha_bwd.restype = C.c_int
        L.cdetr_mha_bwd.argtypes = [_p] * 8 + [C.c_int32] * 3 + [C.c_float, C.c_int32, _p]
        L.cdetr_match_cost.restype = C.c_int
        L.cdetr_match_cost.argtypes = [_p, C.c_int32, _p, _p, _p, _p, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                       C.c_float, _p, _p]
        L.cdetr_lsap.restype = C.c_int
        L.cdetr_lsap.argtypes = [_p, _p, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p, _p, _p, _p]
        for name, args in (("cdetr_mask_prep", [_p] + [C.c_int32] * 5 + [_p] * 7),
                           ("cdetr_stem_pack", [_p, _p] + [C.c_int32] * 7 + [_p]),
                           ("cdetr_exemplar_fwd", [_p, _p, _p] + [C.c_int32] * 6 + [_p] * 4),
                           ("cdetr_exemplar_bwd", [_p] * 4 + [C.c_int32] * 4 + [_p]),
                           ("cdetr_aggr_weight_fwd", [_p] * 4 + [C.c_int32] * 3 + [_p]),
                           ("cdetr_aggr_weight_bwd", [_p] * 5 + [C.c_int32] * 3 + [_p]),
                           ("cdetr_box_head_fwd", [_p] * 3 + [C.c_int32] * 2 + [_p]),
                           ("cdetr_box_head_bwd", [_p] * 5 + [C.c_int32] * 2 + [_p])):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = args
        if L.cdetr_abi_version() != 2:
            raise RuntimeError("libcdetr_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().cdetr_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """hipStream_t of torch's current stream on the current device.  The raw accessor costs ~0.3 us; torch.cuda.current_stream()
    builds a Stream object and resolves the device index in Python (~9 us, i.e. ~6 ms of host time over the ~700 launches of a step)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The product path only ever runs on the GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("counting_detr_amd kernels need device tensors (no CPU fallback exists)")
    return t.data_ptr()
