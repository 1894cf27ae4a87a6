#!/usr/bin/env python3
"""Time the fused window-attention kernels alone (config-B shapes), with and without bias.
Meant to be run under `rocprofv3 --kernel-trace --stats` or on its own (event timing)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import functional as K  # noqa: E402
from visiontransformer_amd._lib import QKV_CHUNKED, QKV_INTERLEAVED  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
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
    dtype = torch.bfloat16 if len(sys.argv) < 2 or sys.argv[1] != "f32" else torch.float32
    for name, shape, layout, grid in (("wmsa", (B, HW, HW, 3 * C), QKV_INTERLEAVED, (B, HW, HW)),
                                      ("dsmsa", (128, w * w, 3 * C), QKV_CHUNKED, None)):
        for bias in (True, False):
            qkv = torch.randn(shape, device=dev).to(dtype).requires_grad_()
            tab = (torch.randn((2 * w - 1) ** 2, heads, device=dev) * 0.02).requires_grad_() if bias else None
            o = K.window_attention(qkv, tab, w, heads, layout, grid=grid)
            do = torch.randn_like(o)
            tf = timeit(lambda: K.window_attention(qkv.detach(), None if tab is None else tab.detach(), w, heads, layout, grid=grid))
            ins = (qkv, tab) if bias else (qkv,)
            tb = timeit(lambda: torch.autograd.grad(o, ins, do, retain_graph=True))
            T = qkv.numel() // (3 * C)
            print("%-6s bias=%d %s  fwd %8.1f us (%6.1f TF/s)   bwd %8.1f us (%6.1f TF/s)" %
                  (name, bias, str(dtype)[6:], tf, 4.0 * T * w * w * C / tf / 1e6, tb, 8.0 * T * w * w * C / tb / 1e6))


if __name__ == "__main__":
    main()
