"""Selection / gather / scatter (and optionally projection) kernels at the bench shapes, one JSON line.
usage: python tools/window_bench.py [--proj] [--k K]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from visiontransformer_amd import _lib
    _lib.load()
    k = int(sys.argv[sys.argv.index("--k") + 1]) if "--k" in sys.argv else 4
    out = {}
    x = torch.randn(bench.B_PER_GPU, bench.HW, bench.HW, bench.C, device=dev)
    bench._window_rooflines(dev, 30, 5, k, x, out)
    if "--proj" in sys.argv:
        bench._projection_rooflines(dev, 30, 5, x, out)
    for name, v in out.items():
        if "achieved" in v:
            print("%-18s %7.1f us  %7.1f GB/s  frac %.3f" % (name, v["us"], v["achieved"], v["achieved"] / v["peak"]))
        else:
            print("%-18s %7.1f us" % (name, v["us"]))
