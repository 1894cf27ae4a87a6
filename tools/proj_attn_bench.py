#!/usr/bin/env python3
"""Inference forward of the attention branch at the config-B shapes: q|k|v projection + attention as two kernels
against the one fused kernel (dsw_win_attn_proj_fwd).  Prints event times, how many output elements differ, and
one JSON line with the figures."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import functional as K  # noqa: E402
from visiontransformer_amd._lib import QKV_CHUNKED, QKV_INTERLEAVED  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, HW, C, heads, w = 32, 64, 180, 6, 16
    Cp = K._ceil_pad(C)
    res = {}
    for name, lead, layout, grid in (("wmsa", (B, HW, HW), QKV_INTERLEAVED, (B, HW, HW)),
                                     ("dsmsa", (128, w * w), QKV_CHUNKED, None)):
        T = 1
        for n in lead:
            T *= n
        xp = K._mark_padded(torch.zeros(T, Cp, dtype=torch.bfloat16, device=dev))
        xp[:, :C] = torch.randn(T, C, device=dev).bfloat16()
        x = xp[:, :C].view(lead + (C,))
        lin = torch.nn.Linear(C, 3 * C).to(dev)
        tab = torch.randn((2 * w - 1) ** 2, heads, device=dev) * 0.02
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            def separate():
                return K.window_attention(K.linear(x, lin, pad_out=True), tab, w, heads, layout, grid=grid, pad_out=True)

            def fused():
                return K.projected_window_attention(x, lin, tab, w, heads, layout, grid=grid, pad_out=True)

            def only_linear():
                return K.linear(x, lin, pad_out=True)
            a, b = separate(), fused()
            # the three are timed in turn, five rounds, and the medians taken: clocks drift between runs
            rounds = [(timeit(separate), timeit(fused), timeit(only_linear)) for _ in range(5)]
            t_sep, t_fused, t_lin = (sorted(r[i] for r in rounds)[2] for i in range(3))
        diff = int((a != b).sum())
        flop = 4.0 * T * w * w * C + 2.0 * T * Cp * 3 * (heads * 32)     # attention + projection as the kernel runs it
        alg = T * (Cp + Cp) * 2                                          # x in, out written; weights and table are L2-resident
        print("%-6s tokens %7d: projection %6.1f us + attention %6.1f us = %6.1f us;  fused %6.1f us  (%.2fx)  "
              "differing elements %d of %d;  fused: %.1f TFLOP/s, %.2f TB/s of x + out" %
              (name, T, t_lin, t_sep - t_lin, t_sep, t_fused, t_sep / t_fused, diff, a.numel(), flop / t_fused / 1e6, alg / t_fused / 1e6))
        res[name] = dict(tokens=T, separate_us=round(t_sep, 1), projection_us=round(t_lin, 1), fused_us=round(t_fused, 1),
                         differing=diff, elements=a.numel(), fused_tflops=round(flop / t_fused / 1e6, 1),
                         mfma_frac=round(flop / t_fused / 1e6 / 2500.0, 4), hbm_frac=round(alg / t_fused / 1e6 / 8.0, 4))
    print(json.dumps({"proj_attn_fwd": res}))


if __name__ == "__main__":
    main()
