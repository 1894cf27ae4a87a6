"""ctypes binding of libdswin_hip.so (the C ABI declared in include/dswin_hip.h).

There is no fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdswin_hip.so")

DSW_F32, DSW_BF16 = 0, 1
QKV_CHUNKED, QKV_INTERLEAVED = 0, 1

_lib = None
_lock = threading.Lock()

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_Z = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/dswin_hip.h declares
SIGNATURES = {
    "dswin_version": (_I, []),
    "dswin_last_error_string": (ctypes.c_char_p, []),
    "dsw_select_fwd": (_I, [_P, _P, _I, _P, _F] + [_I] * 7 + [_P] * 12 + [_P]),
    "dsw_select_bwd": (_I, [_P, _P, _P, _P, _P, _I, _F, _I, _I, _I, _P, _P, _P, _P]),
    "dsw_select_bwd_sd": (_I, [_P, _P, _P, _P, _P, _I, _F, _I, _I, _I, _P, _I, _P, _P, _P]),
    "dsw_gather_fwd": (_I, [_P, _I, _P, _P, _P] + [_I] * 6 + [_P, _I, _I, _P]),
    "dsw_gather_bwd": (_I, [_P, _I, _P, _P, _P] + [_I] * 7 + [_P, _I, _I, _P]),
    "dsw_scatter_mean_fwd": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P] + [_I] * 7 + [_P, _I, _P]),
    "dsw_scatter_mean_bwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P] + [_I] * 7 +
                             [_P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P]),
    "dsw_scatter_mean_bwd_add": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P] + [_I] * 7 +
                                 [_P, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "dsw_win_attn_fwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _F, _P, _P, _P, _P, _F, ctypes.c_ulonglong, _P, _P]),
    "dsw_win_attn_bwd_workspace": (_Z, [_I] * 8),
    "dsw_win_attn_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _F, _P, _P, _P,
                              _P, _P, _P, _P, _P, _Z, _F, ctypes.c_ulonglong, _P, _P]),
    "dsw_unshuffle_nhwc_ln_fwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _F, _P, _I, _P, _P, _P]),
    "dsw_unshuffle_blocks": (_I, [_I, _I, _I, _I]),
    "dsw_unshuffle_nhwc_ln_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dsw_shuffle_nchw_fwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dsw_unshuffle_nhwc_fwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dsw_scatter_mean_fwd_live": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P] + [_I] * 7 + [_P, _I, _P, _P]),
    "dsw_dwconv3x3_fwd_src": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P]),
    "dsw_dwconv3x3_residual_fwd_src": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "dsw_dwconv3x3_fwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _I, _P, _P]),
    "dsw_dwconv3x3_residual_fwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "dsw_cast_rows": (_I, [_P, _I, ctypes.c_longlong, _P, _I, ctypes.c_longlong, ctypes.c_longlong,
                           ctypes.c_longlong, _I, _I, _P]),
    "dsw_weight_prep": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "dsw_weight_prep_batch": (_I, [_I, _P, _P, _P, _P, _P, _P, _P]),
    "dsw_linear": (_I, [_P, _I, ctypes.c_longlong, ctypes.c_longlong, _I, _P, _I, _I, _P, _I, _P, _I, ctypes.c_longlong, _I, _I, _P,
                        ctypes.c_longlong, _P, _I, _P]),
    "dsw_linear_gated_silu": (_I, [_P, _I, ctypes.c_longlong, ctypes.c_longlong, _I, _P, _I, _I, _P, _P, ctypes.c_longlong, _P]),
    "dsw_linear_rmsnorm": (_I, [_P, _I, ctypes.c_longlong, ctypes.c_longlong, _I, _F, _P, _I, _I, _P, _I, _P, _I, ctypes.c_longlong, _I, _P]),
    "dsw_linear_rmsnorm_gated_silu": (_I, [_P, _I, ctypes.c_longlong, ctypes.c_longlong, _I, _F, _P, _I, _I, _P, _P, ctypes.c_longlong, _P]),
    "dsw_mlp_gated_prep": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "dsw_mlp_gated_residual": (_I, [_P, ctypes.c_longlong, ctypes.c_longlong, _I, _F, _P, _P, _P, _P, _I, _P, _P, ctypes.c_longlong, _P]),
    "dsw_linear_gelu": (_I, [_P, _I, ctypes.c_longlong, ctypes.c_longlong, _I, _P, _I, _I, _P, _I, _P, ctypes.c_longlong, _I, _P, ctypes.c_longlong,
                             _P, ctypes.c_longlong, _P]),
    "dsw_linear_gelu_bwd": (_I, [_P, _I, ctypes.c_longlong, ctypes.c_longlong, _I, _P, _I, _I, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _I,
                                 _P, ctypes.c_longlong, _F, ctypes.c_ulonglong, _P]),
    "dsw_linear_dropout_in": (_I, [_P, ctypes.c_longlong, ctypes.c_longlong, _I, _P, _I, _I, _P, _I, ctypes.c_longlong, _I, _P, ctypes.c_longlong,
                                   _F, ctypes.c_ulonglong, _P]),
    "dsw_linear_residual_dropout": (_I, [_P, _I, ctypes.c_longlong, ctypes.c_longlong, _I, _P, _I, _I, _P, _I, _P, ctypes.c_longlong, _P, _P,
                                         ctypes.c_longlong, _I, _F, ctypes.c_ulonglong, _P]),
    "dsw_linear_residual": (_I, [_P, _I, ctypes.c_longlong, ctypes.c_longlong, _I, _P, _I, _I, _P, _I, _P, ctypes.c_longlong, _P, _P, ctypes.c_longlong, _I, _P]),
    "dsw_sd_stencil_fwd": (_I, [_P, _I, _P, _I, _I, _I, _P, _P]),
    "dsw_sd_gather_bwd": (_I, [_P, _I, _I, _I, _P, _P]),
    "dsw_sparse_disc_tail_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P]),
    "dsw_sparse_disc_tail_bwd": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "dsw_linear_wgrad_workspace": (_Z, [ctypes.c_longlong, _I, _I]),
    "dsw_linear_wgrad": (_I, [_P, ctypes.c_longlong, _P, ctypes.c_longlong, ctypes.c_longlong, _I, _I, _P, _P, _P, _P, _I, _P]),
    "dsw_rmsnorm_blocks": (_I, [ctypes.c_longlong]),
    "dsw_rmsnorm_fwd": (_I, [_P, _P, _F, ctypes.c_longlong, _I, _P, _I, ctypes.c_longlong, _P, _P]),
    "dsw_rmsnorm_bwd": (_I, [_P, _I, ctypes.c_longlong, _P, _P, _P, ctypes.c_longlong, _I, _P, _P, _P, _P]),
    "dsw_gated_silu_fwd": (_I, [_P, _I, ctypes.c_longlong, _I, ctypes.c_longlong, _P, _I, ctypes.c_longlong, _P]),
    "dsw_gated_silu_bwd": (_I, [_P, _I, ctypes.c_longlong, _P, _I, ctypes.c_longlong, _I, ctypes.c_longlong, _P, _P]),
    "dsw_gelu_fwd": (_I, [_P, _I, ctypes.c_longlong, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _P]),
    "dsw_shuffle2_gelu_nhwc": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "dsw_shuffle2_gelu_nhwc_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "dsw_gelu_bwd": (_I, [_P, ctypes.c_longlong, _P, _I, ctypes.c_longlong, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _P]),
    "dsw_rezero_blocks": (_I, [ctypes.c_longlong, ctypes.c_longlong]),
    "dsw_rezero_fwd": (_I, [_P, _P, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _I, _P, _P]),
    "dsw_rezero_bwd": (_I, [_P, _P, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _I, _P, ctypes.c_longlong, _P, _P]),
    "dsw_rezero_dropout_fwd": (_I, [_P, _P, _I, ctypes.c_longlong, _P, _P, _F, ctypes.c_longlong, _I, _P, _P]),
    "dsw_rezero_dropout_bwd": (_I, [_P, _P, _I, ctypes.c_longlong, _P, _P, _F, ctypes.c_longlong, _I, _P, ctypes.c_longlong, _P, _P]),
    "dsw_rezero_dropout_seeded_fwd": (_I, [_P, _P, _I, ctypes.c_longlong, _P, _F, ctypes.c_ulonglong, ctypes.c_longlong, _I, _P, _P]),
    "dsw_rezero_dropout_seeded_bwd": (_I, [_P, _P, _I, ctypes.c_longlong, _P, _F, ctypes.c_ulonglong, ctypes.c_longlong, _I, _P,
                                           ctypes.c_longlong, _P, _P]),
    "dsw_conv3x3_nhwc": (_I, [_P, _I, ctypes.c_longlong, _I, _I, _I, _I, _P, _I, _I, _P, _I, _P, _I, ctypes.c_longlong, _I, _P, ctypes.c_longlong, _P]),
    "dsw_conv3x3_nhwc_residual_copy": (_I, [_P, _I, ctypes.c_longlong, _I, _I, _I, _I, _P, _I, _I, _P, _I, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _I,
                                            _P, ctypes.c_longlong, _P]),
    "dsw_conv3x3_nhwc_residual": (_I, [_P, _I, ctypes.c_longlong, _I, _I, _I, _I, _P, _I, _I, _P, _I, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _I, _P]),
    "dsw_conv3x3_wgrad_workspace": (_Z, [_I, _I, _I, _I, _I]),
    "dsw_conv3x3_wgrad": (_I, [_P, ctypes.c_longlong, _P, ctypes.c_longlong, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "dsw_layernorm_fwd": (_I, [_P, _P, _P, _F, ctypes.c_longlong, _I, _P, _I, ctypes.c_longlong, _P, _P, _P]),
    "dsw_layernorm_bwd": (_I, [_P, _I, ctypes.c_longlong, _P, _P, _P, _P, ctypes.c_longlong, _I, _P, _P, _P, ctypes.c_longlong, _I, _P, _P]),
    "dsw_ema_update": (_I, [_I, _P, _P, _P, _I, _F, _F, _P]),
    "dsw_win_attn_proj_fwd": (_I, [_P, ctypes.c_longlong, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _F, _P, _P, _P, _P]),
    "dsw_adam_update": (_I, [_I, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _P]),
    "dsw_colsum_blocks": (_I, [ctypes.c_longlong]),
    "dsw_colsum": (_I, [_P, _I, ctypes.c_longlong, _I, _P, _P, _P]),
    "dsw_colsum_static_max_floats": (_Z, []),
    "dsw_colsum_static": (_I, [_P, _I, ctypes.c_longlong, _I, _P, _P]),
    "dsw_dwconv3x3_bwd_blocks": (_I, [_I, _I, _I]),
    "dsw_dwconv3x3_bwd": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P]),
    "dsw_sparse_disc_fwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "dsw_sparse_disc_bwd_blocks": (_I, [_I, _I]),
    "dsw_sparse_disc_bwd_scratch": (_Z, [_I, _I, _I, _I, _I]),
    "dsw_sparse_disc_bwd": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I,
                                 _P, _P, _P, _P, _P, _P, _P, _P]),
}


def load():
    """dlopen the library once and attach the signatures; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "visiontransformer_amd: %s is missing - build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950); there is no CPU or PyTorch fallback" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def dtype_code(t):
    if t == torch.float32:
        return DSW_F32
    if t == torch.bfloat16:
        return DSW_BF16
    raise RuntimeError("visiontransformer_amd: unsupported dtype %s (float32 and bfloat16 are built)" % t)


def ptr(t):
    """device pointer of a tensor (None -> NULL); the tensor must be contiguous"""
    if t is None:
        return None
    if not t.is_contiguous():
        raise RuntimeError("visiontransformer_amd: non-contiguous tensor passed to the C ABI")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream(device=None):
    """the current HIP stream of `device` as an integer handle (a module of a few hundred launches per forward
    asks for it that often: the raw query is ten times cheaper than building a torch.cuda.Stream object)"""
    if _raw_stream is not None:
        idx = None if device is None else (device if isinstance(device, int) else device.index)
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


_entry = {}


def call(name, *args):
    """invoke an int-returning entry point and turn a nonzero status into RuntimeError"""
    fn = _entry.get(name)
    if fn is None:
        fn = _entry[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        msg = load().dswin_last_error_string().decode("utf-8", "replace")
        raise RuntimeError("%s failed (code %d): %s" % (name, rc, msg))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "visiontransformer_amd kernels run on the GPU only (got a %s tensor); "
                "there is no CPU path" % t.device)
