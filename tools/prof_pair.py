#!/usr/bin/env python3
"""torch profiler view of one step of bench.py's attention pair: which aten ops (with shapes) launch the kernels that are not the library's"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from visiontransformer_amd import _lib
dev = torch.device("cuda:0"); _lib.load()
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


for _ in range(10):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print("%-28s calls %3d  gpu %8.1f us/step  %s" % (e.key, e.count // 3, e.device_time_total / 3, str(e.input_shapes)[:110]))
print()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=14, max_name_column_width=40, max_src_column_width=90))
