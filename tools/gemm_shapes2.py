#!/usr/bin/env python3
import torch
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it
dev = "cuda"; T = 131072
print("Y = X[T,K] W[N,K]^T + b")
for K, N in ((180, 576), (184, 544), (184, 540), (192, 544), (192, 560), (540, 192), (544, 192), (544, 184), (560, 192), (184, 184), (180, 192), (192, 180)):
    x = torch.randn(T, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    print("  K=%d N=%d : %7.1f us" % (K, N, t(lambda: torch.addmm(b, x, w.t()))))
print("dX = dY[T,N] W[N,K]  (no transpose)")
for N, K in ((540, 180), (576, 192), (544, 192), (544, 184), (180, 180), (192, 192), (184, 184)):
    dy = torch.randn(T, N, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    print("  N=%d K=%d : %7.1f us" % (N, K, t(lambda: dy @ w)))
print("strided: dY view [T,540] of [T,576] @ W[540,180]; and padded weights")
dyp = torch.randn(T, 576, device=dev, dtype=torch.bfloat16); w = torch.randn(540, 180, device=dev, dtype=torch.bfloat16)
print("  view: %7.1f us" % t(lambda: dyp[:, :540] @ w))
