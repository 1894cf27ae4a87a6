#!/usr/bin/env python3
"""The projection rows of the bench (q|k|v forward from the fp32 stream with the bf16 copy, out-projection with the
residual, q|k|v input gradient, weight gradient) at 131 072 tokens, medians of five rounds of 20 launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from visiontransformer_amd import _lib  # noqa: E402
from visiontransformer_amd._lib import call, dtype_code, ptr, stream  # noqa: E402

dev = torch.device("cuda:0")
_lib.load()
bf, f32 = torch.bfloat16, torch.float32
M, C = 131072, 180
c64 = lambda n: (n + 63) // 64 * 64


def prep(N, K):
    w = torch.randn(N, K, device=dev) * 0.1
    wp = torch.empty(c64(N), c64(K), device=dev, dtype=bf)
    wpt = torch.empty(c64(K), c64(N), device=dev, dtype=bf)
    call("dsw_weight_prep", ptr(w), dtype_code(w.dtype), N, K, ptr(wp), ptr(wpt), stream(dev))
    return wp, wpt


def lin(a, K, wp, N, c, ncols, a_copy=None):
    call("dsw_linear", ptr(a), dtype_code(a.dtype), a.stride(0), a.shape[0], K, ptr(wp), wp.shape[0], wp.stride(0), None, N, ptr(c),
         dtype_code(c.dtype), c.stride(0), ncols, 0, None if a_copy is None else ptr(a_copy), 0 if a_copy is None else a_copy.stride(0), None, 0,
         stream(dev))


x = torch.randn(M, C, device=dev)
wp, wpt = prep(3 * C, C)
qkv = torch.empty(M, 576, device=dev, dtype=bf)
xk = torch.empty(M, 192, device=dev, dtype=bf)
rows = [("q|k|v forward (fp32 in, bf16 out, copy)", lambda: lin(x, C, wp, 3 * C, qkv, 576, xk), M * (C * 4 + 576 * 2 + 192 * 2))]
z = torch.randn(M, 192, device=dev).to(bf)
wo, wot = prep(C, C)
zo = torch.empty(M, 192, device=dev, dtype=bf)
rows.append(("projection bf16 -> bf16, 192 -> 192", lambda: lin(z, 192, wo, C, zo, 192), M * (192 * 2 + 192 * 2)))
dq = torch.randn(M, 576, device=dev).to(bf)
dx = torch.empty(M, C, device=dev)
rows.append(("q|k|v input gradient (bf16 in, fp32 out)", lambda: lin(dq, 576, wpt, 0, dx, C), M * (576 * 2 + C * 4)))
dy = torch.randn(M, C, device=dev)
dz = torch.empty(M, 192, device=dev, dtype=bf)
dyk = torch.empty(M, 192, device=dev, dtype=bf)
rows.append(("out-projection input gradient (fp32 in, bf16 out, copy)", lambda: lin(dy, C, wot, 0, dz, 192, dyk), M * (C * 4 + 192 * 2 * 2)))
for name, fn, byt in rows:
    t = bench.time_calls(fn, 20, warm=5, rounds=5)
    print("%-58s %6.1f us  %5.2f TB/s" % (name, t * 1e6, byt / t / 1e12), flush=True)
