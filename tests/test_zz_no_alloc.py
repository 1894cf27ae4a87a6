"""dsw_colsum_static: the column-sum entry point for callers that cannot allocate (its per-block partial sums
live in a static device array inside the library).  Checked through the C ABI against dsw_colsum (caller-provided
scratch) bit for bit - same kernels, same summation order - on two calls in a row, the second on a smaller
input, so stale partials of the first call would show."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, x, dev):
    rows, cols = x.shape
    code = 0 if x.dtype == torch.float32 else 1
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ref = torch.empty(cols, dtype=torch.float32, device=dev)
    part = torch.empty(lib.dsw_colsum_blocks(rows) * cols, dtype=torch.float32, device=dev)
    rc = lib.dsw_colsum(ctypes.c_void_p(x.data_ptr()), code, rows, cols, ctypes.c_void_p(ref.data_ptr()),
                        ctypes.c_void_p(part.data_ptr()), st)
    assert rc == 0, lib.dswin_last_error_string()
    got = torch.full((cols,), float("nan"), dtype=torch.float32, device=dev)
    rc = lib.dsw_colsum_static(ctypes.c_void_p(x.data_ptr()), code, rows, cols, ctypes.c_void_p(got.data_ptr()), st)
    assert rc == 0, lib.dswin_last_error_string()
    torch.cuda.synchronize(dev)
    return got, ref


def test_colsum_static_matches_colsum_on_two_calls_in_a_row(cuda0):
    from visiontransformer_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device=cuda0).manual_seed(11)
    with torch.cuda.device(cuda0):
        big = torch.randn(70001, 576, device=cuda0, generator=gen).to(torch.bfloat16)
        small = torch.randn(333, 180, device=cuda0, generator=gen)
        for x in (big, small):
            got, ref = _both(lib, x, cuda0)
            assert torch.equal(got, ref)
            exact = x.double().sum(0)
            assert float((got.double() - exact).abs().max()) <= 1e-3 * float(x.double().abs().sum(0).max())
        # sizes beyond the static buffer are refused with a message, nothing is launched
        cap = lib.dsw_colsum_static_max_floats()
        rows = (cap // 2048 + 1) * 128
        out = torch.zeros(2048, dtype=torch.float32, device=cuda0)
        rc = lib.dsw_colsum_static(ctypes.c_void_p(big.data_ptr()), 1, rows, 2048, ctypes.c_void_p(out.data_ptr()),
                                   ctypes.c_void_p(torch.cuda.current_stream(cuda0).cuda_stream))
        assert rc != 0 and b"static buffer" in lib.dswin_last_error_string()
        torch.cuda.synchronize(cuda0)
        assert float(out.abs().max()) == 0.0
