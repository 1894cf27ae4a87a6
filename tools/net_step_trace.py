#!/usr/bin/env python3
"""Training steps of the DSWT default / MSFT config C network (batch 32 x 64x64, bf16 autocast, L1 loss, the package's
Adam) for a rocprofv3 kernel trace: rocprofv3 --kernel-trace --output-format csv -- python3 tools/net_step_trace.py
[dswt|msft] [steps]  (tools/trace_summary.py reads the trace).  Prints the time per step of the steps after the first
four.  DSW_FUSE_GELU=0: the MLP's activation as kernels of its own; DSW_BATCH_PREP=0: one weight-preparation launch per
projection."""
import os
import sys

import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import models, optim  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "dswt"
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = (models.DSWTransformer(**models.DSWT_DEFAULT) if which == "dswt" else models.MSFTransformer(**models.MSFT_CONFIG_C)).to(dev).train()
opt = optim.Adam(net.parameters(), lr=2e-4, betas=(0.9, 0.99))
x = torch.rand(32, 3, 64, 64, device=dev) * 2 - 1
tgt = torch.rand(32, 3, 256, 256, device=dev) * 2 - 1
from visiontransformer_amd import functional as K  # noqa: E402
K.FUSE_GELU = os.environ.get("DSW_FUSE_GELU", "1") != "0"
K.BATCH_WEIGHT_PREP = os.environ.get("DSW_BATCH_PREP", "1") != "0"
models.NHWC_ENDS_TRAINING = os.environ.get("DSW_NHWC_ENDS_TRAINING", "1") == "1"
K.FUSE_RESIDUAL_DROPOUT = os.environ.get("DSW_FUSE_RESID_DROP", "1") != "0"
K.FUSE_CONV_RESIDUAL_TRAINING = os.environ.get("DSW_FUSE_CONV_RESID", "1") != "0"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
t0 = None
for i in range(steps):
    if i == 4:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(x)
    torch.nn.functional.l1_loss(y.float(), tgt).backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
if t0 is not None:
    print("%s: %.2f ms per step (FUSE_GELU=%s BATCH_WEIGHT_PREP=%s NHWC_ENDS_TRAINING=%s)" % (
        which, (time.perf_counter() - t0) / (steps - 4) * 1e3, K.FUSE_GELU, K.BATCH_WEIGHT_PREP, models.NHWC_ENDS_TRAINING) + " FUSE_RESIDUAL_DROPOUT=%s" % K.FUSE_RESIDUAL_DROPOUT, flush=True)
