#!/usr/bin/env python3
"""Times the fused window-attention forward and backward at the config-B shapes on the row pitches the modules use
(q|k|v rows of 576, output / gradient rows of 192 elements) and on dense rows, five rounds, medians.
DSW_ATTN_FWD_GENERAL=1 / DSW_ATTN_BWD_GENERAL=1 time the general kernels instead."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import functional as K  # noqa: E402
from visiontransformer_amd._lib import QKV_CHUNKED, QKV_INTERLEAVED  # noqa: E402


def timeit(fn, iters=int(os.environ.get("W16_ITERS", "20")), rounds=int(os.environ.get("W16_ROUNDS", "5"))):
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
    B, HW, C, heads, w = 32, 64, 180, 6, 16
    a = torch.randn(4096, 4096, device=dev)
    for _ in range(200):  # bring the clocks up before the first measurement
        a = a @ a * 1e-3
    torch.cuda.synchronize()
    # "contig": the same 3 072 (window, head) work items as "wmsa" with every item's operands in one contiguous block of
    # memory (one head, 16x16 images, a token's q|k|v in 180 bytes of a 256-byte row): what a head-major layout would
    # give the kernels
    for name, lead, layout, grid in (("wmsa", (B, HW, HW), QKV_INTERLEAVED, (B, HW, HW)), ("dsmsa", (128, w * w), QKV_CHUNKED, None),
                                     ("contig", (3072, w, w), QKV_INTERLEAVED, (3072, w, w)),
                                     ("contig2", (1536, w, w), QKV_INTERLEAVED, (1536, w, w))):
        if os.environ.get("W16_SHAPE", name) != name:
            continue
        # "contig2": two heads, 360 of a 384-byte row used, a window's two work items in one 96 KB block
        C, heads = {"contig": (30, 1), "contig2": (60, 2)}.get(name, (180, 6))
        qp, op = {"contig": (128, 64), "contig2": (192, 64)}.get(name, (576, 192))
        for padded in (True, False, True):
            if os.environ.get("W16_PITCH", "both") not in ("both", "padded" if padded else "dense"):
                continue
            for bias in (True, False):
                T = 1
                for n in lead:
                    T *= n
                # W16_SETS buffer sets taken in turn: with one set (about 200 MB forward, 400 MB backward) the
                # operands of the next call are what the last one left in the 256 MB Infinity Cache; with three or
                # more every call reads from HBM, as the kernels do inside a training step
                nsets = int(os.environ.get("W16_SETS", "1"))
                sets = []
                tab = (torch.randn((2 * w - 1) ** 2, heads, device=dev) * 0.02).requires_grad_() if bias else None
                for _ in range(nsets):
                    if padded:
                        buf = K._mark_padded(torch.zeros(T, qp, device=dev, dtype=torch.bfloat16))
                        buf[:, :3 * C] = torch.randn(T, 3 * C, device=dev).bfloat16()
                        qkv = buf[:, :3 * C].view(lead + (3 * C,)).requires_grad_()
                        gbuf = torch.zeros(T, op, device=dev, dtype=torch.bfloat16)
                        gbuf[:, :C] = torch.randn(T, C, device=dev).bfloat16()
                        do = gbuf[:, :C].view(lead + (C,))
                    else:
                        qkv = torch.randn(lead + (3 * C,), device=dev).bfloat16().requires_grad_()
                        do = torch.randn(lead + (C,), device=dev).bfloat16()
                    o = K.window_attention(qkv, tab, w, heads, layout, grid=grid, pad_out=padded)
                    sets.append((qkv, do, o))
                turn = [0]

                def fwd():
                    qkv = sets[turn[0] % nsets][0]
                    turn[0] += 1
                    return K.window_attention(qkv, tab, w, heads, layout, grid=grid, pad_out=padded)

                def bwd():
                    qkv, do, o = sets[turn[0] % nsets]
                    turn[0] += 1
                    return torch.autograd.grad(o, (qkv, tab) if bias else (qkv,), do, retain_graph=True)
                with torch.no_grad():
                    tf = timeit(fwd)
                tb = timeit(bwd)
                del sets
                fb = (4 * T * C * 2 + T * heads * 4) / 1e6
                bb = (8 * T * C * 2 + T * heads * 4) / 1e6
                print("%-6s pitch=%s bias=%d  fwd %7.1f us (%5.2f TB/s, %5.1f TF/s)   bwd %7.1f us (%5.2f TB/s, %5.1f TF/s)" %
                      (name, "576/192" if padded else "dense  ", bias, tf, fb / tf, 4.0 * T * w * w * C / tf / 1e6, tb, bb / tb,
                       8.0 * T * w * w * C / tb / 1e6), flush=True)


if __name__ == "__main__":
    main()
