#!/usr/bin/env python3
"""dsw_linear with fp32 results into dense rows (N floats) against rows padded to 64 floats: what the partially
written 128-byte lines of the dense layout cost.  bf16 padded input, K = N = 180, the token counts of the bench
shape and of a 960x540 frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiontransformer_amd import _lib
import bench

dev = torch.device("cuda:0")
_lib.load()
bf, f32 = torch.bfloat16, torch.float32
K = N = 180
Kp = Np = 192
w = torch.randn(N, K, device=dev) * 0.1
wp = torch.empty(Np, Kp, device=dev, dtype=bf)
wpt = torch.empty(Kp, Np, device=dev, dtype=bf)
_lib.call("dsw_weight_prep", _lib.ptr(w), _lib.dtype_code(w.dtype), N, K, _lib.ptr(wp), _lib.ptr(wpt), _lib.stream(dev))
bias = torch.randn(N, device=dev)
for M in (131072, 522240):
    a = torch.zeros(M, Kp, device=dev, dtype=bf)
    a[:, :K] = torch.randn(M, K, device=dev).to(bf)
    for c_dt, ld in ((f32, N), (f32, Np), (bf, Np)):
        c = torch.empty(M, ld, device=dev, dtype=c_dt)
        f = lambda: _lib.call("dsw_linear", _lib.ptr(a), _lib.dtype_code(bf), Kp, M, Kp, _lib.ptr(wp), Np, Kp, _lib.ptr(bias), N, _lib.ptr(c),
                              _lib.dtype_code(c_dt), ld, ld, 0, None, 0, None, 0, _lib.stream(dev))
        t = bench.time_calls(f, 20)
        byt = M * (Kp * 2 + ld * c.element_size())
        print("M=%6d out %s ld=%3d : %7.1f us  %6.0f GB/s" % (M, str(c_dt)[6:], ld, t * 1e6, byt / t / 1e9), flush=True)
