#!/usr/bin/env python3
"""Which model-level convolutions of the MSFT training step are slow: per Conv2d module, the
input's dtype / strides and the synchronised forward time; the slow backward calls by shape
from the torch profiler."""
import time
import torch
from torch import nn
from visiontransformer_amd.models import MSFT_CONFIG_C, MSFTransformer

dev = torch.device("cuda:0")
torch.manual_seed(0)
import sys
cfg = dict(MSFT_CONFIG_C)
if "full" not in sys.argv:
    cfg["n_blocks"] = 1; cfg["n_sub_blocks"] = 1
net = MSFTransformer(**cfg).to(dev).train()
import os
x = torch.randn(int(os.environ.get("C_BATCH", "8")), 3, 64, 64, device=dev)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(x)
    y.float().square().mean().backward()
    net.zero_grad(set_to_none=True)
for _ in range(2): step()
torch.cuda.synchronize()
names = {m: n for n, m in net.named_modules()}
t0 = {}
def pre(m, inp):
    torch.cuda.synchronize(); t0[m] = time.perf_counter()
def post(m, inp, out):
    torch.cuda.synchronize()
    i = inp[0]
    print("fwd %-40s %6.2f ms in %s %s strides %s contiguous %s w %s" % (names[m], (time.perf_counter() - t0[m]) * 1e3,
          tuple(i.shape), i.dtype, i.stride(), i.is_contiguous(), m.weight.dtype), flush=True)
hs = []
for m in net.modules():
    if isinstance(m, nn.Conv2d) and m.groups == 1:
        hs += [m.register_forward_pre_hook(pre), m.register_forward_hook(post)]
step()
for h in hs: h.remove()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=50))
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=90))
