#!/usr/bin/env python3
"""a few launches of dsw_linear (qkv shape, bf16 in / out) and dsw_linear_wgrad for counter collection"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiontransformer_amd import _lib
dev = torch.device("cuda:0"); _lib.load()
bf = torch.bfloat16
M = 131072
a = torch.randn(M, 192, device=dev).to(bf); wp = (torch.randn(576, 192, device=dev) * 0.1).to(bf)
bias = torch.randn(540, device=dev); c = torch.empty(M, 576, device=dev, dtype=bf)
dw = torch.empty(540, 180, device=dev); db = torch.empty(540, device=dev)
ws = torch.empty(_lib.load().dsw_linear_wgrad_workspace(M, 540, 180), device=dev)
for _ in range(3):
    _lib.call("dsw_linear", _lib.ptr(a), _lib.dtype_code(bf), 192, M, 192, _lib.ptr(wp), 576, 192, _lib.ptr(bias), 540, _lib.ptr(c), _lib.dtype_code(bf),
              576, 576, 0, None, 0, None, 0, _lib.stream(dev))
    _lib.call("dsw_linear_wgrad", _lib.ptr(c), 576, _lib.ptr(a), 192, M, 540, 180, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws), None, 0, _lib.stream(dev))
torch.cuda.synchronize()
