#!/usr/bin/env python3
"""Diagnostic (library built with -DBWD30_STAMPS): where a workgroup of the head-dim-30 attention backward spends
its life.  Prints, over all workgroups of one launch at the WindowedMSA shape, the median shader-clock cycles between
the stamps: 0 entry, 1 operands arrived, 2 staged, 3 past the first barrier, 4 phase 1 done, 5 middle section done
(dK / dV stored, K / V restaged, table gradient out), 6 phase 2 done, 7 dQ stored."""
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import functional as K, _lib  # noqa: E402
from visiontransformer_amd._lib import QKV_INTERLEAVED  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, HW, C, heads, w = 32, 64, 180, 6, 16
    T = B * HW * HW
    buf = K._mark_padded(torch.zeros(T, 576, device=dev, dtype=torch.bfloat16))
    buf[:, :3 * C] = torch.randn(T, 3 * C, device=dev).bfloat16()
    qkv = buf[:, :3 * C].view(B, HW, HW, 3 * C).requires_grad_()
    gbuf = torch.zeros(T, 192, device=dev, dtype=torch.bfloat16)
    gbuf[:, :C] = torch.randn(T, C, device=dev).bfloat16()
    do = gbuf[:, :C].view(B, HW, HW, C)
    for bias in (True, False):
        tab = (torch.randn((2 * w - 1) ** 2, heads, device=dev) * 0.02).requires_grad_() if bias else None
        o = K.window_attention(qkv, tab, w, heads, QKV_INTERLEAVED, grid=(B, HW, HW), pad_out=True)
        for _ in range(5):
            torch.autograd.grad(o, (qkv, tab) if bias else (qkv,), do, retain_graph=True)
        torch.cuda.synchronize()
        n = 3072 * 16
        host = (ctypes.c_ulonglong * n)()
        lib = _lib.load()
        rc = lib.dsw_debug_read_stamps(host, n)
        assert rc == 0, rc
        st = [host[i * 16:(i + 1) * 16] for i in range(3072)]
        names = ["load", "stage", "barrier", "phase1", "middle", "phase2", "dq store"]
        print("bias=%d  median cycles per workgroup (shader clock):" % bias)
        for j, nm in enumerate(names):
            d = [s[j + 1] - s[j] for s in st]
            print("   %-9s %8.0f  (p10 %8.0f, p90 %8.0f)" % (nm, statistics.median(d), sorted(d)[len(d) // 10], sorted(d)[9 * len(d) // 10]))
        for a, b, nm in ((0, 12, "issue loads"), (12, 1, "table+arrive"), (4, 8, "dK dV out"), (8, 9, "K V to LDS"), (9, 10, "table grad"), (10, 11, "padding"), (11, 5, "mid barrier")):
            d = sorted(s[b] - s[a] for s in st)
            print("     %-12s %8.0f  (p10 %8.0f, p90 %8.0f)" % (nm, d[len(d) // 2], d[len(d) // 10], d[9 * len(d) // 10]))
        life = [s[7] - s[0] for s in st]
        t0 = min(s[0] for s in st)
        t1 = max(s[7] for s in st)
        print("   life      %8.0f   whole launch %d cycles; sum of lives / 512 slots = %.0f" % (statistics.median(life), t1 - t0, sum(life) / 512))


if __name__ == "__main__":
    main()
