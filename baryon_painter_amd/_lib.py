"""ctypes binding of ``libbp_hip.so`` (the C ABI declared in include/bp_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``csrc/build.py``.  There is NO
fallback: if it is missing or a call fails, a ``RuntimeError`` is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbp_hip.so")

BP_OK = 0
BP_EINVAL, BP_EUNSUPPORTED, BP_ELAUNCH, BP_EWORKSPACE = -1, -2, -3, -4
IMPL_AUTO, IMPL_DIRECT, IMPL_MFMA, IMPL_BF16 = 0, 1, 2, 3
IMPL_SHARED = 0x100
IMPL_DEFER = 0x200
PACK_FWD, PACK_BWD = 0, 1
F32, BF16 = 0, 1


class View(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("c", C.c_int32), ("cstride", C.c_int32), ("coff", C.c_int32), ("dtype", C.c_int32)]


class Pointwise(C.Structure):
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("slope", C.c_void_p)]


class Conv(C.Structure):
    _fields_ = [("transposed", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("out_pad", C.c_int32)]


class BnTrain(C.Structure):
    _fields_ = [("count", C.c_double), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float),
                ("momentum", C.c_float), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("num_batches_tracked", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("save_mean", C.c_void_p), ("save_invstd", C.c_void_p)]


class BnBackwardFin(C.Structure):
    _fields_ = [("count", C.c_double), ("gamma", C.c_void_p), ("save_mean", C.c_void_p), ("save_invstd", C.c_void_p),
                ("param_grad_scale", C.c_float), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("coef_abc", C.c_void_p)]


class Latent(C.Structure):
    _fields_ = [("n", C.c_int32), ("L", C.c_int32), ("zc", C.c_int32), ("zh", C.c_int32),
                ("zw", C.c_int32), ("min_z_var", C.c_float)]


class Loglik(C.Structure):
    _fields_ = [("n", C.c_int32), ("L", C.c_int32), ("c", C.c_int32), ("h", C.c_int32),
                ("w", C.c_int32), ("mu_softplus", C.c_int32), ("predict_var", C.c_int32),
                ("alpha_var", C.c_float), ("beta_kl", C.c_float), ("likelihood_scaling", C.c_float)]


_P = C.c_void_p
_VP = C.POINTER(View)
_PWP = C.POINTER(Pointwise)
_CP = C.POINTER(Conv)

# name -> (restype, argtypes); one entry per declaration in include/bp_hip.h
SIGNATURES = {
    "bp_version": (C.c_int, []),
    "bp_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "bp_conv_ws_kind": (C.c_int, [_CP, C.c_int, _VP, _VP]),
    "bp_strerror": (C.c_char_p, [C.c_int]),
    "bp_conv_packed_floats": (C.c_int64, [_CP, C.c_int]),
    "bp_conv_kernel_id": (C.c_int, [_CP, C.c_int]),
    "bp_conv_pack": (C.c_int, [_CP, C.c_int, _P, _P, _P]),
    "bp_conv_pack_job_bytes": (C.c_int32, []),
    "bp_conv_pack_job": (C.c_int, [_CP, C.c_int, _P, _P, _P, C.POINTER(C.c_int64)]),
    "bp_conv_pack_jobs": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P]),
    "bp_conv_bf16_packed_elems": (C.c_int64, [_CP, C.c_int]),
    "bp_conv_bf16_pack": (C.c_int, [_CP, C.c_int, _P, _P, _P]),
    "bp_conv_bf16_supported": (C.c_int, [_CP, C.c_int, _VP, _VP]),
    "bp_conv_forward": (C.c_int, [_CP, _VP, _PWP, _P, _P, _P, _VP, C.c_int, _P]),
    "bp_conv_backward_data": (C.c_int, [_CP, _VP, _P, _P, _VP, C.c_int, _P]),
    "bp_conv_stats_workspace": (C.c_size_t, [_CP, C.c_int, _VP, _VP, C.c_int]),
    "bp_conv_forward_stats": (C.c_int, [_CP, _VP, _PWP, _P, _VP, _P, _P, C.c_size_t, C.c_int, _P]),
    "bp_conv_forward_bn": (C.c_int, [_CP, _VP, _PWP, _P, _VP, _P, C.POINTER(BnTrain), _P, C.c_size_t, C.c_int, _P]),
    "bp_conv_backward_data_stats": (C.c_int, [_CP, _VP, _P, _VP, _VP, _PWP, _P, _P, C.c_size_t, _P]),
    "bp_conv_backward_data_act_workspace": (C.c_size_t, [_CP, _VP, _VP]),
    "bp_conv_backward_data_act": (C.c_int, [_CP, _VP, _P, _VP, _VP, _PWP, _P, _P, C.c_size_t, _P]),
    "bp_conv_backward_weight_workspace": (C.c_size_t, [_CP, _VP, _VP]),
    "bp_conv_backward_weight": (C.c_int, [_CP, _VP, _PWP, _VP, _P, _P, _P, C.c_size_t, C.c_int, _P]),
    "bp_wgrad_defer_begin": (C.c_int, []),
    "bp_wgrad_defer_flush": (C.c_int, [C.c_int, _P]),
    "bp_peer_handle_bytes": (C.c_int, []),
    "bp_peer_max_doubles": (C.c_int, []),
    "bp_peer_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p), _P]),
    "bp_peer_open": (C.c_int, [_P, _P]),
    "bp_peer_all_reduce": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P]),
    "bp_peer_status": (C.c_int64, [_P]),
    "bp_peer_bind": (C.c_int, [_P]),
    "bp_peer_slots": (C.c_int, []),
    "bp_peer_exchange_check": (C.c_int, [_P, _P, C.c_int, _P]),
    "bp_peer_destroy": (C.c_int, [_P]),
    "bp_channel_sums_workspace": (C.c_size_t, [_VP]),
    "bp_channel_sums": (C.c_int, [_VP, _P, _P, C.c_size_t, _P]),
    "bp_bn_finalize": (C.c_int, [_P, C.c_double, C.c_int32, _P, _P, C.c_float, C.c_float, _P, _P, _P,
                                 _P, _P, _P, _P, _P]),
    "bp_bn_eval_pointwise": (C.c_int, [C.c_int32, _P, _P, _P, _P, C.c_float, _P, _P, _P]),
    "bp_act_backward_workspace": (C.c_size_t, [_VP]),
    "bp_act_backward": (C.c_int, [_VP, _VP, _VP, _PWP, _VP, _VP, _P, _P, C.c_size_t, _P]),
    "bp_act_backward_bn": (C.c_int, [_VP, _VP, _VP, _PWP, _VP, _VP, _P, C.POINTER(BnBackwardFin), _P, C.c_size_t, _P]),
    "bp_bn_backward_finalize": (C.c_int, [_P, C.c_double, C.c_int32, _P, _P, _P, C.c_float, _P, _P, _P, _P]),
    "bp_bn_backward_apply": (C.c_int, [_VP, _VP, _P, _VP, _P]),
    "bp_act_bn_backward_apply": (C.c_int, [_VP, _VP, _VP, _PWP, _VP, _P, _VP, _P]),
    "bp_sums_to_float": (C.c_int, [_P, C.c_int32, _P, _P]),
    "bp_prelu_slope_grad": (C.c_int, [_P, C.c_int32, _P, _P]),
    "bp_residual_forward": (C.c_int, [_VP, _PWP, _VP, _PWP, C.c_float, _VP, _P]),
    "bp_nchw_to_view": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _VP, _P]),
    "bp_view_to_nchw": (C.c_int, [_VP, _PWP, C.c_int32, _P, _P]),
    "bp_fill": (C.c_int, [_P, C.c_int64, C.c_float, _P]),
    "bp_paint_load": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _VP, _P]),
    "bp_paint_load2": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _VP, _VP, _P]),
    "bp_paint_store": (C.c_int, [_VP, _PWP, C.c_int32, _P, _P, _P]),
    "bp_philox_normal": (C.c_int, [C.c_uint64, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "bp_philox_normal_dev": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "bp_latent_forward": (C.c_int, [C.POINTER(Latent), _VP, _PWP, _VP, _PWP, _P, _P, _VP, _P, _P,
                                    C.c_size_t, _P]),
    "bp_latent_backward": (C.c_int, [C.POINTER(Latent), _VP, _P, _P, _P, C.c_float, _VP, _VP, _P]),
    "bp_loglik_workspace": (C.c_size_t, [C.POINTER(Loglik)]),
    "bp_loglik_forward": (C.c_int, [C.POINTER(Loglik), _P, _VP, _VP, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "bp_loglik_backward": (C.c_int, [C.POINTER(Loglik), _P, _VP, _VP, _P, _VP, _VP, _P]),
    "bp_unary_forward": (C.c_int, [_VP, _PWP, C.c_int32, _VP, _P]),
    "bp_bce_logits": (C.c_int, [_VP, C.c_int32, C.c_int32, C.c_float, _P, _P, C.c_size_t, _P]),
    "bp_bce_logits_grad": (C.c_int, [_VP, C.c_int32, C.c_int32, C.c_float, C.c_float, _VP, _P]),
    "bp_l1_sum": (C.c_int, [_VP, _P, _P, _P, C.c_size_t, _P]),
    "bp_tanh_l1_backward": (C.c_int, [_VP, _P, _VP, C.c_float, _VP, _P]),
    "bp_gather_tiles": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, _P]),
    "bp_adam_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                               C.c_int32, _P]),
    "bp_adam_step_dev": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, _P]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch first: its wheel bundles the HIP runtime this process must share.  Loading libbp_hip.so before torch
    # brings in the system runtime instead and the first launch fails (seen as "kernel launch failed" when
    # __graft_entry__.build() and smoke() ran in one process).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP kernels are not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != BP_OK:
        msg = load().bp_strerror(rc).decode()
        raise RuntimeError(f"libbp_hip: {what} failed: {msg} (code {rc})")


def view(t, n, h, w, c, cstride=None, coff=0):
    """bp_view over tensor ``t`` (any shape, contiguous) interpreted as (n,h,w,cstride)."""
    cstride = c if cstride is None else cstride
    return View(t.data_ptr() if hasattr(t, "data_ptr") else int(t), n, h, w, c, cstride, coff)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())
