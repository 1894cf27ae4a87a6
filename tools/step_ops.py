#!/usr/bin/env python3
"""torch profiler view of one bench step: aten ops (with input shapes) that launch kernels of their own"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
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
    model.zero_grad(set_to_none=True); x.grad = None
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=4) if e.key.startswith("aten::") and e.self_device_time_total > 3]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:24]:
    st = [s for s in e.stack if "visiontransformer_amd" in s or "bench.py" in s][:2]
    print("%-28s %7.1f us x%d  %s  %s" % (e.key, e.self_device_time_total, e.count, str(e.input_shapes)[:70], " | ".join(s.split("/")[-1][:60] for s in st)))
