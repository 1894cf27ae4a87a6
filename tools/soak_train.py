#!/usr/bin/env python3
"""Overfits one fixed synthetic batch for a number of steps with either network (bf16 autocast, L1 loss, the package's
Adam), once with every training-side fusion of the package on and once with them off (stock NCHW ends, separate
residual / activation / norm passes), from the same initial weights and seeds: prints both loss curves and the largest
relative difference.  A check that the fused training path optimises the same function, over more steps than the unit
tests run.  usage: [SOAK_FULL=1] tools/soak_train.py [dswt|msft] [steps]"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import functional as K, models, optim  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "dswt"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = dict(models.DSWT_DEFAULT if which == "dswt" else models.MSFT_CONFIG_C)
if os.environ.get("SOAK_FULL") != "1":     # SOAK_FULL=1: the whole default network (8 x 3 blocks)
    cfg.update(n_blocks=2, n_sub_blocks=2)
net0 = (models.DSWTransformer(**cfg) if which == "dswt" else models.MSFTransformer(**cfg)).to(dev).train()
with torch.no_grad():      # open the ReZero gates a little so that every branch takes part from the first step
    for n, p in net0.named_parameters():
        if "rezero" in n:
            p.fill_(0.1)
x = torch.rand(8, 3, 64, 64, device=dev)
tgt = torch.nn.functional.interpolate(x, scale_factor=4, mode="bicubic").clamp(0, 1)
FLAGS = [(K, "FUSE_GELU"), (K, "FUSE_RESIDUAL_DROPOUT"), (K, "FUSE_CONV_RESIDUAL_TRAINING"), (K, "GRAD_MEETING"), (models, "NHWC_ENDS_TRAINING")]
curves = {}
for fused in (True, False):
    for mod, name in FLAGS:
        setattr(mod, name, fused)
    net = copy.deepcopy(net0)
    opt = optim.Adam(net.parameters(), lr=2e-4, betas=(0.9, 0.99))
    torch.manual_seed(1)
    losses = []
    for i in range(steps):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x)
        loss = torch.nn.functional.l1_loss(y.float(), tgt)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(loss.detach().item())
    curves[fused] = losses
    assert all(l == l for l in losses), "NaN in the loss"
for mod, name in FLAGS:
    setattr(mod, name, True)
a, b = curves[True], curves[False]
print("%s, %d steps: fused %.4f -> %.4f, separate %.4f -> %.4f" % (which, steps, a[0], a[-1], b[0], b[-1]))
print("loss every 10th step, fused   :", " ".join("%.4f" % v for v in a[::10]))
print("loss every 10th step, separate:", " ".join("%.4f" % v for v in b[::10]))
if os.environ.get("SOAK_ALL") == "1":
    print("fused   :", " ".join("%.3f" % v for v in a))
    print("separate:", " ".join("%.3f" % v for v in b))
print("largest relative difference of the two curves: %.3e" % max(abs(u - v) / max(abs(v), 1e-9) for u, v in zip(a, b)))
