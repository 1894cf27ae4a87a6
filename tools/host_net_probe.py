#!/usr/bin/env python3
"""Whole-network training step (DSWT default / MSFT headline config, batch 32 x 64x64): how long the host takes to
enqueue a step against how long the GPU takes to run it."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import models, optim  # noqa: E402

dev = torch.device("cuda:0")
for name, build in (("dswt_default", lambda: models.DSWTransformer(**models.DSWT_DEFAULT)),
                    ("msft_config_c", lambda: models.MSFTransformer(**models.MSFT_CONFIG_C))):
    torch.manual_seed(0)
    net = build().to(dev).train()
    opt = optim.Adam(net.parameters(), lr=2e-4, betas=(0.9, 0.99))
    x = torch.rand(32, 3, 64, 64, device=dev) * 2 - 1
    tgt = torch.rand(32, 3, 256, 256, device=dev) * 2 - 1

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x)
        torch.nn.functional.l1_loss(y.float(), tgt).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    for n in (3, 6):
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s %d steps: enqueued in %.1f ms/step, drained in %.1f ms/step" % (name, n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
    del net, opt
    torch.cuda.empty_cache()
