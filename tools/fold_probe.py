#!/usr/bin/env python3
"""Inference of a one-block default-width DSWT with the RMSNorm folded into the projections and without: which entry
points ran, how far the two outputs are apart, and the time per frame-sized call of the two forms."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import functional as K, models  # noqa: E402
from visiontransformer_amd.parallel import GraphedForward  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = dict(models.DSWT_DEFAULT)
cfg.update(n_blocks=2)
net = models.DSWTransformer(**cfg).to(dev).eval()
with torch.no_grad():
    for n, p in net.named_parameters():
        if "rezero" in n:
            p.fill_(0.3)
x = torch.rand(1, 3, 256, 256, device=dev) * 2 - 1
calls = []
real = K.call


def spy(name, *a):
    calls.append(name)
    return real(name, *a)


outs = {}
for fold in (True, False):
    K.FOLD_RMSNORM = fold
    calls.clear()
    K.call = spy
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        outs[fold] = net(x).float()
    K.call = real
    print("fold", fold, {n: calls.count(n) for n in ("dsw_linear_rmsnorm", "dsw_linear_rmsnorm_gated_silu", "dsw_rmsnorm_fwd", "dsw_linear", "dsw_linear_gated_silu")})
print("rel diff", float((outs[True] - outs[False]).norm() / outs[False].norm()))
big = torch.rand(1, 3, 544, 960, device=dev) * 2 - 1
for fold in (True, False, True, False):
    K.FOLD_RMSNORM = fold
    run = GraphedForward(net)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(3):
            run(big)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run(big)
        torch.cuda.synchronize()
    print("fold %s: %.2f ms per 544x960 frame (2 blocks)" % (fold, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
