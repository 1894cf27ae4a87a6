#!/usr/bin/env python3
"""cProfile of the bench step's host side (where does the Python / launch time go)."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
from visiontransformer_amd import _lib
_lib.load()
model = bench.build_pair(dev)
gen = torch.Generator(device=dev).manual_seed(100)
x = torch.randn(bench.B_PER_GPU, bench.HW, bench.HW, bench.C, device=dev, generator=gen).requires_grad_()
model.sdwa.gumbel_noise = -torch.empty(bench.B_PER_GPU, 16, device=dev).exponential_(generator=gen).log()

def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = model(x)
    y.float().sum().backward()
    model.zero_grad(set_to_none=True)
    x.grad = None

for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step, with final sync %.3f ms/step" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
