#!/usr/bin/env python3
"""How long the host takes to enqueue one bench step against how long the GPU takes to run it: N steps are
enqueued back to back on an idle stream; the time until the last launch returns is the host's share (when it is the
shorter one the queue runs ahead of the GPU), the time until the stream drains is the step."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    from visiontransformer_amd import _lib
    _lib.load()
    model = bench.build_pair(dev)
    gen = torch.Generator(device=dev).manual_seed(100)
    x = torch.randn(bench.B_PER_GPU, bench.HW, bench.HW, bench.C, device=dev, generator=gen).requires_grad_()
    model.sdwa.gumbel_noise = -torch.empty(bench.B_PER_GPU, (bench.HW // bench.WIN) ** 2, device=dev).exponential_(generator=gen).log()

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = model(x)
        y.float().sum().backward()
        model.zero_grad(set_to_none=True)
        x.grad = None

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    gc.collect()
    gc.freeze()
    for n in (5, 10, 20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%2d steps: enqueued in %.3f ms/step, drained in %.3f ms/step" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)


if __name__ == "__main__":
    main()
