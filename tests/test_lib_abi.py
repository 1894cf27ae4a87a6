"""CPU-side checks of the C ABI: the built library loads, exports every function that
include/dswin_hip.h declares, and the ctypes table covers exactly that set.  No kernel runs."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "dswin_hip.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:dsw|dswin)_[a-z0-9_]+)\s*\(", txt)))


def _lib_path():
    from visiontransformer_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.LIB_PATH


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for must in ("dsw_select_fwd", "dsw_gather_fwd", "dsw_gather_bwd", "dsw_scatter_mean_fwd",
                 "dsw_scatter_mean_bwd", "dsw_win_attn_fwd", "dsw_win_attn_bwd", "dswin_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib_path())
    for name in _declared():
        assert hasattr(lib, name), "libdswin_hip.so lacks %s" % name


def test_ctypes_table_matches_header():
    from visiontransformer_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.dswin_version() >= 100
    assert lib.dswin_last_error_string() is not None


def test_argument_errors_are_reported_without_a_gpu():
    """argument validation happens before any HIP call, so it can be exercised on the CPU box"""
    from visiontransformer_amd import _lib
    lib = _lib.load()
    rc = lib.dsw_win_attn_fwd(None, 0, 0, 1, 0, 0, 4, 8, 2, 0, 0, None, 1.0, None, None, None, None, 0.0, 0, None, None)
    assert rc == 2 and b"window size" in lib.dswin_last_error_string()
    rc = lib.dsw_win_attn_fwd(None, 0, 0, 1, 0, 0, 8, 8, 2, 20, 0, None, 1.0, None, None, None, None, 0.0, 0, None, None)
    assert rc == 1 and b"row strides" in lib.dswin_last_error_string()
    rc = lib.dsw_gather_fwd(None, 0, None, None, None, 1, 1, 10, 16, 8, 8, None, 0, 0, None)
    assert rc == 1 and b"multiples of the window size" in lib.dswin_last_error_string()


def test_argument_errors_of_the_one_kernel_mlp():
    """dsw_mlp_gated_residual / dsw_mlp_gated_prep refuse widths beyond their tiles and misaligned operands before any launch"""
    import ctypes as C
    from visiontransformer_amd import _lib
    lib = _lib.load()
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)
    f = C.c_float(1e-6)
    rc = lib.dsw_mlp_gated_residual(p, 200, 4, 200, f, p, p, p, p, 12, p, p, 200, None)
    assert rc == 2 and b"C=200" in lib.dswin_last_error_string()
    rc = lib.dsw_mlp_gated_residual(p, 180, 4, 180, f, p, p, p, p, 17, p, p, 180, None)
    assert rc == 2 and b"hidden chunks" in lib.dswin_last_error_string()
    rc = lib.dsw_mlp_gated_residual(C.c_void_p(p.value + 4), 180, 4, 180, f, p, p, p, p, 12, p, p, 180, None)
    assert rc == 2 and b"16-byte aligned" in lib.dswin_last_error_string()
    assert lib.dsw_mlp_gated_residual(p, 180, 0, 180, f, p, p, p, p, 12, p, p, 180, None) == 0      # no rows: nothing to do
    rc = lib.dsw_mlp_gated_prep(p, p, None, None, p, None, None, 600, 180, 180, p, p, p, p, None)
    assert rc == 2 and b"H=600" in lib.dswin_last_error_string()


def test_cpu_tensors_are_refused():
    import torch
    from visiontransformer_amd import functional as K
    with pytest.raises(RuntimeError, match="GPU only"):
        K.window_attention(torch.zeros(1, 64, 24), None, 8, 2, 0)
