#!/usr/bin/env python3
"""Kernel times of the windowed attention with LePE (forward + backward) at the two MSFT shapes of workload C."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from visiontransformer_amd import functional as K
from visiontransformer_amd._lib import QKV_CHUNKED

dev = torch.device("cuda:0")
PMC = os.environ.get("PMC") == "1"   # under rocprofv3 --pmc: two plain steps of the first shape, no torch profiler
for (B, H, C, heads) in ((32, 64, 216, 4), (32, 32, 96, 6))[:1 if PMC else 2]:
    torch.manual_seed(0)
    qkv = torch.randn(B, H * H, 3 * C, device=dev, dtype=torch.bfloat16, requires_grad=True)
    lw = torch.randn(C, 1, 3, 3, device=dev, requires_grad=True)
    lb = torch.randn(C, device=dev, requires_grad=True)
    NOLEPE = os.environ.get("NOLEPE") == "1"
    DROP = float(os.environ.get("DROP", "0.1"))
    def step():
        z = K.window_attention(qkv, None, 8, heads, QKV_CHUNKED, grid=(B, H, H), lepe_w=None if NOLEPE else lw,
                               lepe_b=None if NOLEPE else lb, dropout_p=DROP, dropout_seed=5)
        z.backward(torch.ones_like(z))
    for _ in range(2 if PMC else 3):
        step()
    torch.cuda.synchronize()
    if PMC:
        break
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    print(f"B={B} {H}x{H} C={C} heads={heads} d={C // heads}")
    for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total):
        if e.device_time_total > 0 and "dsw" in e.key:
            print("  %8.1f us  x%-3d %s" % (e.device_time_total / e.count, e.count // 10, e.key[:90]))
