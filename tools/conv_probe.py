#!/usr/bin/env python3
"""MIOpen timing of the model-level convolutions of MSFT config C / DSWT (fwd + bwd): bf16 vs fp32,
immediate mode vs benchmark (find) mode."""
import sys
import torch, torch.nn.functional as F
def t(fn, it=3):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
dev = "cuda"
SHAPES = ((8, 3, 192, 64, 3), (8, 32, 3, 256, 3), (8, 64, 32, 256, 3), (8, 24, 24, 32, 3), (8, 96, 24, 64, 1),
          (8, 216, 192, 64, 1), (8, 192, 64, 64, 3), (8, 64, 64, 64, 3), (8, 64, 256, 128, 3),
          (32, 3, 180, 64, 3), (32, 16, 3, 256, 3), (32, 64, 16, 256, 3), (32, 180, 64, 64, 3))
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for (B, Ci, Co, H, k) in SHAPES:
        row = []
        for dt in (torch.bfloat16, torch.float32):
            x = torch.randn(B, Ci, H, H, device=dev, dtype=dt, requires_grad=True)
            w = torch.randn(Co, Ci, k, k, device=dev, dtype=dt, requires_grad=True)
            def f():
                y = F.conv2d(x, w, padding=k // 2)
                y.sum().backward()
            row.append(t(f))
        print("bench=%d B%d %d->%d %dx%d k%d : bf16 %8.2f ms  fp32 %8.2f ms" % (bench, B, Ci, Co, H, H, k, row[0], row[1]), flush=True)
