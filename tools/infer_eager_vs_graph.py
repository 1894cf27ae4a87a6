#!/usr/bin/env python3
"""One 544x960 frame through the DSWT default network (eval, bf16 autocast): eager launches against the replay of a
captured graph (parallel.GraphedForward), ms per frame, interleaved twice."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import models  # noqa: E402
from visiontransformer_amd.parallel import GraphedForward  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = models.dswt_from_config(models.DSWT_DEFAULT).to(dev).eval()
with torch.no_grad():
    for n, p in net.named_parameters():
        if "rezero" in n:
            p.fill_(0.2)
x = torch.rand(1, 3, 544, 960, device=dev)
runner = GraphedForward(net)


def timed(fn, iters=8):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(3):
            fn(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


from visiontransformer_amd.parallel import tiled_super_resolve  # noqa: E402
lr = torch.rand(1, 3, 540, 960, device=dev)
runner2 = GraphedForward(net)
for _ in range(2):
    print("eager %.2f ms   graph %.2f ms" % (timed(net), timed(runner)), flush=True)
    x, keep = lr, x
    print("540x960 frame through tiled_super_resolve: eager %.2f ms   graph %.2f ms" % (
        timed(lambda t: tiled_super_resolve(net, t, 4, tile="balanced", halo=32, reduce=False)),
        timed(lambda t: tiled_super_resolve(runner2, t, 4, tile="balanced", halo=32, reduce=False))), flush=True)
    x = keep
