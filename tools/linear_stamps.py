#!/usr/bin/env python3
"""Diagnostic (library built with -DLINEAR_STAMPS): where a workgroup of the projection kernel spends its life at the
bench shapes.  Median shader-clock cycles between the stamps of all workgroups of one launch: 0 entry, 1 operand
loads issued, 2 first slab written to LDS (= loads arrived), 3 past the barrier, 4 first chunk done (MFMAs + store),
5 all chunks done, 6 results out (several-slab form: accumulators stored at the end)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import functional as K, _lib  # noqa: E402


def report(tag, nblk):
    torch.cuda.synchronize()
    n = nblk * 8
    host = (ctypes.c_ulonglong * n)()
    assert _lib.load().dsw_debug_read_linear_stamps(host, n) == 0
    st = [host[i * 8:(i + 1) * 8] for i in range(nblk)]
    names = ["issue loads", "arrive+LDS", "barrier", "chunk 0", "other chunks", "final stores"]
    print(tag)
    for j, nm in enumerate(names):
        d = sorted(s[j + 1] - s[j] for s in st if s[j + 1] >= s[j])
        if d:
            print("   %-13s %8.0f  (p10 %8.0f, p90 %8.0f)" % (nm, d[len(d) // 2], d[len(d) // 10], d[9 * len(d) // 10]))
    life = sorted(s[6] - s[0] for s in st)
    print("   life          %8.0f" % life[len(life) // 2])


def main():
    dev = torch.device("cuda:0")
    T, C = 32 * 64 * 64, 180
    lin_qkv = torch.nn.Linear(C, 3 * C).to(dev)
    lin_out = torch.nn.Linear(C, C).to(dev)
    x = torch.randn(T, C, device=dev, requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(3):
            qkv = K.linear(x, lin_qkv, pad_out=True)
        report("qkv forward: fp32 rows [T,180] -> bf16 [T,576]", T // 128)
        g = torch.randn_like(qkv)
        gp = K._mark_padded(torch.zeros(T, 576, device=dev, dtype=torch.bfloat16))
        gp[:, :540] = g
        for _ in range(3):
            torch.autograd.grad(qkv, x, gp[:, :540], retain_graph=True)
        report("qkv input gradient: bf16 [T,576] -> fp32 [T,180] (last launch of the backward may be the weight gradient's)", T // 128)
        z = qkv[:, :C]
        for _ in range(3):
            K.linear(z.detach(), lin_out, pad_out=True)
        report("out projection forward: bf16 rows (pitch 576) -> bf16 [T,192]", T // 128)


if __name__ == "__main__":
    main()
