#!/usr/bin/env python3
"""Diagnostic (library built with -DML_STAMPS): where a workgroup of the one-kernel MLP spends its life on a 544x960
frame.  Median shader-clock cycles (100 MHz clock ticks x the ratio printed) between the stamps of the first 4 080
workgroups: 0 entry, 1 loads issued, 2 operands staged + barrier, 3 chunk 0 first product, 4 gated, 5 second product,
6 next chunk's weights staged (two barriers), 7 all chunks, 8 residual requested, 9 results out."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import functional as K, _lib, models  # noqa: E402

dev = torch.device("cuda:0")
mlp = models.GatedMLP(180, ffn_scale=2).to(dev).eval()
norm = torch.nn.RMSNorm(180).to(dev)
r = torch.full((1,), 0.2, device=dev)
T = 544 * 960
x = torch.randn(T, 180, device=dev)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(3):
        mlp(x, residual=(x, r), prenorm=norm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    mlp(x, residual=(x, r), prenorm=norm)
    e1.record()
torch.cuda.synchronize()
nblk = (T + 127) // 128
n = nblk * 16
host = (ctypes.c_ulonglong * n)()
assert _lib.load().dsw_debug_read_mlp_stamps(host, n) == 0
st = [host[i * 16:(i + 1) * 16] for i in range(nblk)]
names = ["issue loads", "arrive+stage", "fc1 (chunk 0)", "gate (chunk 0)", "fc2 (chunk 0)", "restage (chunk 0)", "chunks 1..", "request resid", "results out"]
t_all = max(s[9] for s in st) - min(s[0] for s in st)
print("launch %.1f us, stamps span %d ticks" % (e0.elapsed_time(e1) * 1e3, t_all))
for j, nm in enumerate(names):
    d = sorted(s[j + 1] - s[j] for s in st if s[j + 1] >= s[j])
    print("   %-18s %8.0f  (p10 %8.0f, p90 %8.0f)" % (nm, d[len(d) // 2], d[len(d) // 10], d[9 * len(d) // 10]))
life = sorted(s[9] - s[0] for s in st)
print("   life               %8.0f" % life[len(life) // 2])
