#!/usr/bin/env python3
"""Three training steps of the DSWT default network (batch 32 x 64x64, bf16 autocast, L1 loss, the package's Adam) for a
rocprofv3 kernel trace: rocprofv3 --kernel-trace --stats -- python3 tools/net_step_trace.py [dswt|msft]"""
import os
import sys

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
for _ in range(4):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(x)
    torch.nn.functional.l1_loss(y.float(), tgt).backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
