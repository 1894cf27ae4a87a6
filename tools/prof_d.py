#!/usr/bin/env python3
"""torch profiler view of DSWT x4 inference on one LR tile (rows, columns on the command line; default 320x320; bf16): top kernels"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from visiontransformer_amd import _lib
from visiontransformer_amd.models import DSWT_DEFAULT, DSWTransformer
dev = torch.device("cuda:0"); _lib.load()
torch.manual_seed(0)
m = DSWTransformer(**DSWT_DEFAULT).to(dev).eval()
Hh, Ww = (int(a) for a in (sys.argv[1:3] if len(sys.argv) > 2 else (320, 320)))
x = torch.rand(1, 3, Hh, Ww, device=dev)
def run():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return m(x)
for _ in range(2): run()
torch.cuda.synchronize()
import time
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); print("tile time %.1f ms" % ((time.perf_counter() - t0) * 1e3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=int(os.environ.get("ROWS", "28")), max_name_column_width=80))
