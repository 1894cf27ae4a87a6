#!/usr/bin/env python3
"""hipBLASLt timing of the projection GEMM shapes, with and without padding to multiples of 64."""
import torch

def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it

dev = "cuda"
T = 131072
for K, N in ((180, 540), (192, 540), (192, 576), (180, 180), (192, 192), (540, 180), (576, 192), (576, 180)):
    x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    us = t(lambda: torch.addmm(b, x, w.t()))
    mb = (T * K + T * N) * 2 / 1e6
    print("fwd  M=%d K=%d N=%d : %7.1f us  (%5.1f MB -> %4.2f TB/s, %5.1f TF/s)" % (T, K, N, us, mb, mb / us / 1e6 * 1e6 / 1e6 * 1e0, 2.0 * T * K * N / us / 1e6))
# strided views: x padded to 192 but logical 180
xp = torch.randn(T, 192, device=dev, dtype=torch.bfloat16)
w = torch.randn(540, 180, device=dev, dtype=torch.bfloat16)
print("fwd strided A (lda 192, K 180): %7.1f us" % t(lambda: xp[:, :180] @ w.t()))
yp = torch.empty(T, 576, device=dev, dtype=torch.bfloat16)
print("fwd out= strided C (ldc 576, N 540): %7.1f us" % t(lambda: torch.mm(xp[:, :180], w.t(), out=yp[:, :540])))
# weight gradient: split-K bmm
for K, N in ((180, 540), (192, 576), (180, 180)):
    dy = torch.randn(T, N, device=dev, dtype=torch.bfloat16)
    x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
    S = T // 2048
    us = t(lambda: torch.bmm(dy.view(S, 2048, N).transpose(1, 2), x.view(S, 2048, K)))
    print("wgrad bmm K=%d N=%d : %7.1f us" % (K, N, us))
    for ch in (1024, 4096, 8192):
        S2 = T // ch
        us = t(lambda: torch.bmm(dy.view(S2, ch, N).transpose(1, 2), x.view(S2, ch, K)))
        print("      chunk %5d : %7.1f us" % (ch, us))
