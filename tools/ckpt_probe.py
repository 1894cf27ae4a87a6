#!/usr/bin/env python3
"""Gradients of a small DSWT under bf16 autocast: plain run twice, and under whole-model non-reentrant
checkpointing; prints the relative differences per parameter (largest first)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint
from conftest import rel_err
from test_gpu_trainer import _build

dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "dswt"
gen = torch.Generator(device=dev).manual_seed(5)
lr = torch.rand(2, 3, 32, 32, device=dev, generator=gen) * 2 - 1
hr = torch.rand(2, 3, 128, 128, device=dev, generator=gen) * 2 - 1
runs = []
for mode in ("plain", "plain", "ckpt", "nodrop_plain", "nodrop_ckpt"):
    model = _build(kind, dev)
    if mode.startswith("nodrop"):
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    torch.manual_seed(1)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = checkpoint(model, lr, use_reentrant=False) if mode.endswith("ckpt") else model(lr)
        loss = F.l1_loss(y, hr)
    (loss * 65536.0).backward()
    runs.append((mode, float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters()}))
for a, b in ((0, 1), (0, 2), (3, 4)):
    errs = sorted(((rel_err(runs[b][2][n], g), n) for n, g in runs[a][2].items()), reverse=True)
    print(runs[a][0], "vs", runs[b][0], "loss", runs[a][1], runs[b][1])
    for e, n in errs[:12]:
        print("   %.3e  %s" % (e, n))
    print("   zero-difference parameters: %d of %d" % (sum(1 for e, _ in errs if e == 0), len(errs)))
