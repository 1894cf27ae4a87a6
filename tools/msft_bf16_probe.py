#!/usr/bin/env python3
"""The MSFT golden network's parameter gradients under bf16 autocast against its fp32 run, three times in one process:
the largest relative errors of the channels-last trunk and of the module-by-module path (the spread that the bar of
tests/test_gpu_modules.py::test_msft_model_bf16_stream_against_fp32 has to allow for)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from conftest import rel_err
from test_gpu_modules import load_golden, _load_into
from visiontransformer_amd import models
cuda0 = torch.device('cuda:0')
g, m = load_golden("model_msft")
mod = _load_into(models.MSFTransformer(**m["cfg"]), g, cuda0).eval()
with torch.no_grad():
    for n, p in mod.named_parameters():
        if n.endswith("rezero"):
            p.fill_(0.5)
for sub in mod.modules():
    if hasattr(sub, "attn_drop"):
        sub.attn_drop = 0.0
x = g["x"].to(cuda0)
for rep in range(3):
    res = []
    for stream, autocast in ((False, False), (True, True), (False, True)):
        models.NHWC_STREAM = stream
        mod.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            y = mod(x)
        y.float().square().mean().backward()
        res.append((y.float().detach(), {n: p.grad.float().clone() for n, p in mod.named_parameters() if p.grad is not None}))
    (y32, g32), (y16, g16), (y16n, g16n) = res
    errs = sorted(((rel_err(g16[n], g32[n]), n) for n in g32 if g32[n].abs().max() > 0), reverse=True)[:4]
    errn = sorted(((rel_err(g16n[n], g32[n]), n) for n in g32 if g32[n].abs().max() > 0), reverse=True)[:3]
    print("stream:", [(round(e, 4), n) for e, n in errs]); print("nchw  :", [(round(e, 4), n) for e, n in errn], flush=True)
