"""bench pair step: wall time per step and host-only enqueue time, with and without the host read of the pass count"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
pair = bench.build_pair(dev)
x = torch.randn(32, 64, 64, 180, device=dev, requires_grad=True)
noise = -torch.empty(32, 16, device=dev).exponential_().log()
pair.sdwa.gumbel_noise = noise
def step():
    x.grad = None
    pair.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = pair(x)
    y.float().sum().backward()
for static in (False, True, False, True):
    pair.sdwa.static_training = static
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("static=%s  host enqueue %.3f ms/step   wall %.3f ms/step" % (static, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
print("--- fresh pair, static, wall per block of 5 steps")
pair = bench.build_pair(dev)
pair.sdwa.gumbel_noise = noise
for blk in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    print("  block %d: %.3f ms/step  alloc %.0f MB reserved %.0f MB" % (blk, (time.perf_counter() - t0) / 5 * 1e3, torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20))
