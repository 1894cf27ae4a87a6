#!/usr/bin/env python3
"""torch profiler view of one DSWT (or, MODEL=msft, MSFT) training step (batch 32 of 64x64 LR, bf16): top kernels / ops"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from visiontransformer_amd import _lib
from visiontransformer_amd.models import DSWT_DEFAULT, DSWTransformer
dev = torch.device("cuda:0"); _lib.load()
torch.manual_seed(0)
if os.environ.get("MODEL") == "msft":
    from visiontransformer_amd.models import MSFT_CONFIG_C, MSFTransformer
    m = MSFTransformer(**MSFT_CONFIG_C).to(dev).train()
    for sub in m.modules():
        if hasattr(sub, "attn_drop"):
            sub.attn_drop = 0.0
else:
    m = DSWTransformer(**DSWT_DEFAULT).to(dev).train()
if os.environ.get("STATIC_TRAINING") == "0":      # read the pass count on the host, process only the passing windows
    for sub in m.modules():
        if hasattr(sub, "static_training"):
            sub.static_training = False
B = int(os.environ.get("T_BATCH", "32"))
x = torch.rand(B, 3, 64, 64, device=dev); tgt = torch.rand(B, 3, 256, 256, device=dev)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
    torch.nn.functional.l1_loss(y.float(), tgt).backward()
    m.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=int(os.environ.get("ROWS", "40")), max_name_column_width=90))
if os.environ.get("SHAPES"):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof2:
        step(); torch.cuda.synchronize()
    for e in sorted(prof2.key_averages(group_by_input_shape=True), key=lambda e: -e.device_time_total):
        if e.key in ("aten::mul", "aten::copy_", "aten::sum", "aten::add_", "aten::add", "aten::to", "aten::_to_copy", "aten::fill_", "aten::zero_"):
            print("%-16s %5d calls %9.1f us  %s" % (e.key, e.count, e.device_time_total, str(e.input_shapes)[:140]))
