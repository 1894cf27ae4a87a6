#!/usr/bin/env python3
"""Control for tools/ddp_rehearsal.py: a stock nn.Sequential under DistributedDataParallel, two ranks on ONE GPU over gloo."""
import faulthandler, os, sys, time
import torch
import torch.distributed as dist
T0 = time.time()
faulthandler.dump_traceback_later(40, repeat=False, file=sys.stderr)
rank = int(os.environ["RANK"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="gloo")
torch.manual_seed(0)
if os.environ.get("HEAVY") == "1":    # kernels of a few hundred microseconds each, so that the host runs ahead of the GPU; and a buffer
    layers = []
    for _ in range(6):
        layers += [torch.nn.Linear(2048, 2048), torch.nn.GELU()]
    m = torch.nn.Sequential(*layers).to(dev)
    m.register_buffer("table", torch.arange(1024, device=dev))
    x = torch.randn(16384, 2048, device=dev)
else:
    m = torch.nn.Sequential(torch.nn.Linear(180, 540), torch.nn.GELU(), torch.nn.Linear(540, 180)).to(dev)
    x = torch.randn(4096, 180, device=dev)
ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0])
for step in range(4):
    ddp(x).sum().backward()
    torch.cuda.synchronize()
    print("[rank %d] %5.1f s step %d done" % (rank, time.time() - T0, step), flush=True)
    ddp.zero_grad(set_to_none=True)
dist.destroy_process_group()
