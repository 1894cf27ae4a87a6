#!/usr/bin/env python3
"""Host-side cost of one DSWT x4 inference forward on a small tile (where the launch rate, not the GPU,
bounds the time): cProfile over a few forwards, top functions by own time."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiontransformer_amd import _lib
from visiontransformer_amd.models import DSWT_DEFAULT, dswt_from_config

dev = torch.device("cuda:0"); _lib.load()
torch.manual_seed(0)
m = dswt_from_config(DSWT_DEFAULT).to(dev).eval()
H, W = (int(a) for a in (sys.argv[1:3] if len(sys.argv) > 2 else (304, 272)))
x = torch.rand(1, 3, H, W, device=dev)


def run(n):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(n):
            m(x)


run(3); torch.cuda.synchronize()
t0 = time.perf_counter(); run(5); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("per forward: host %.2f ms, host+drain %.2f ms" % ((t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3))
pr = cProfile.Profile(); pr.enable(); run(5); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(int(os.environ.get("ROWS", "45")))
