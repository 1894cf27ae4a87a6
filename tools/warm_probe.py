"""wall time of the first steps of the bench pair in a fresh process (each step synchronised)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
pair = bench.build_pair(dev)
pair.sdwa.static_training = sys.argv[1] == "1"
x = torch.randn(32, 64, 64, 180, device=dev, requires_grad=True)
pair.sdwa.gumbel_noise = -torch.empty(32, 16, device=dev).exponential_().log()
ts = []
import gc
gc.callbacks.append(lambda ph, info: print('  gc', ph, info) if ph == 'stop' and info.get('generation', 0) >= 1 else None)
for i in range(60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = pair(x)
    y.float().sum().backward()
    pair.zero_grad(set_to_none=True); x.grad = None
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    ts.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
    if ts[-1][1] > 5: print('  slow step', i, ts[-1], 'reserved MB', torch.cuda.memory_reserved() >> 20, 'num_alloc_retries', torch.cuda.memory_stats()['num_alloc_retries'], 'segments', torch.cuda.memory_stats()['segment.all.allocated'])
print("static", sys.argv[1], " host/wall ms per step:", " ".join("%.1f/%.1f" % t for t in ts))
