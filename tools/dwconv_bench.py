#!/usr/bin/env python3
"""Times DS-MSA's depthwise 3x3 convolution (out_conv, DSwinT_arch.py:198, :317) at the training shape (32 x 64 x 64 x
180, fp32) and at a frame's (1 x 544 x 960 x 180, bf16 in): the plain forward, the forward with the residual inside,
and the backward (data gradient + weight / bias gradient kernels).  Medians of five rounds, three buffer sets in turn."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import functional as K  # noqa: E402


def timeit(fn, iters=20, rounds=5):
    for _ in range(3):
        fn()
    res = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / iters)
    return statistics.median(res)


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    C = 180
    w = torch.randn(C, 1, 3, 3, device=dev, requires_grad=True)
    b = torch.randn(C, device=dev, requires_grad=True)
    for shape, dt in (((32, 64, 64, C), torch.float32), ((1, 544, 960, C), torch.bfloat16)):
        sets = [(torch.randn(shape, device=dev).to(dt).requires_grad_(), torch.randn(shape, device=dev)) for _ in range(3)]
        turn = [0]

        def nxt():
            turn[0] += 1
            return sets[turn[0] % 3]

        with torch.no_grad():
            t_f = timeit(lambda: K.dwconv3x3_nhwc(nxt()[0], w, b, out_dtype=torch.float32))
            t_r = timeit(lambda: K.dwconv3x3_plus_residual_nhwc(*((lambda s: (s[0], w, b, s[1]))(nxt()))))
        outs = [K.dwconv3x3_nhwc(x, w, b, out_dtype=torch.float32) for x, _ in sets]

        def bwd():
            turn[0] += 1
            i = turn[0] % 3
            return torch.autograd.grad(outs[i], (sets[i][0], w, b), sets[i][1], retain_graph=True)
        t_b = timeit(bwd)
        n = 1
        for d in shape:
            n *= d
        eb = 2 if dt == torch.bfloat16 else 4
        print("%s %s: fwd %.1f us (%.2f TB/s)  fwd+residual %.1f us (%.2f TB/s)  bwd (dx + dw, db) %.1f us" % (
            tuple(shape), str(dt)[6:], t_f, n * (eb + 4) / t_f / 1e6, t_r, n * (eb + 8) / t_r / 1e6, t_b), flush=True)


if __name__ == "__main__":
    main()
