#!/usr/bin/env python3
"""x + r * mlp(norm(x)) of a DSWT block (C = 180, hidden 360) in inference at a frame's and at the training batch's token
count: the one-kernel form (dsw_mlp_gated_residual) against the two projection kernels.  Medians of five rounds, three
buffer sets in turn; then one forward of the DSWT default network on a 544x960 frame with either form."""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import functional as K, models  # noqa: E402


def timeit(fn, iters=10, rounds=5):
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
    mlp = models.GatedMLP(180, ffn_scale=2).to(dev).eval()
    norm = torch.nn.RMSNorm(180).to(dev)
    r = torch.full((1,), 0.2, device=dev)
    for T in (544 * 960, 32 * 64 * 64):
        xs = [torch.randn(T, 180, device=dev) for _ in range(3)]
        turn = [0]

        def run():
            turn[0] += 1
            x = xs[turn[0] % 3]
            return mlp(x, residual=(x, r), prenorm=norm)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            res = {}
            for fuse in (True, False, True):
                K.FUSE_MLP = fuse
                res[fuse] = timeit(run)
        K.FUSE_MLP = True
        print("%7d tokens: one kernel %.1f us (%.2f TB/s on x + y)   two kernels %.1f us" % (T, res[True], T * 180 * 8 / res[True] / 1e6, res[False]), flush=True)
    if os.environ.get("MLP_NET", "1") == "1":
        net = models.dswt_from_config(models.DSWT_DEFAULT).to(dev).eval()
        with torch.no_grad():
            for n, p in net.named_parameters():
                if "rezero" in n:
                    p.fill_(0.2)
        x = torch.rand(1, 3, 544, 960, device=dev)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            for fuse in (True, False, True, False):
                K.FUSE_MLP = fuse
                for _ in range(2):
                    net(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    net(x)
                torch.cuda.synchronize()
                print("frame, FUSE_MLP=%s: %.2f ms" % (fuse, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
        K.FUSE_MLP = True


if __name__ == "__main__":
    main()
