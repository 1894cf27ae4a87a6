#!/usr/bin/env python3
"""time dsw_linear on the big shapes only (for variant libraries; values are not checked)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiontransformer_amd import _lib
import bench
dev = torch.device("cuda:0"); _lib.load()
bf, f32 = torch.bfloat16, torch.float32
def t(M, K, lda, a_dt, N, ncols, c_dt, copy):
    a = torch.randn(M, lda, device=dev).to(a_dt); wp = (torch.randn(ncols if ncols % 64 == 0 else (ncols + 63) // 64 * 64, (K + 63) // 64 * 64, device=dev) * 0.1).to(bf)
    bias = torch.randn(N, device=dev); c = torch.empty(M, ncols, device=dev, dtype=c_dt)
    ac = torch.empty(M, (K + 63) // 64 * 64, device=dev, dtype=bf) if copy else None
    f = lambda: _lib.call("dsw_linear", _lib.ptr(a), _lib.dtype_code(a_dt), lda, M, K, _lib.ptr(wp), wp.shape[0], wp.shape[1], _lib.ptr(bias), N,
                          _lib.ptr(c), _lib.dtype_code(c_dt), ncols, ncols, 0, _lib.ptr(ac) if copy else None, ac.shape[1] if copy else 0, None, 0, _lib.stream(dev))
    return bench.time_calls(f, 30) * 1e6
M = 131072
print("qkv bf16 %.1f | qkv f32+copy %.1f | out bf16 %.1f | dgrad->f32 %.1f | dgrad->bf16 %.1f" % (
    t(M, 192, 192, bf, 540, 576, bf, 0), t(M, 180, 180, f32, 540, 576, bf, 1), t(M, 192, 192, bf, 180, 192, bf, 0),
    t(M, 576, 576, bf, 180, 180, f32, 0), t(M, 576, 576, bf, 180, 192, bf, 0)), flush=True)
