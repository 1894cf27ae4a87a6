#!/usr/bin/env python3
"""Every tile of the balanced plans for 1 / 2 / 4 / 8 ranks of the 960x540 frame, run on one GPU (all
ranks' shares one after the other): the assembled frames must be finite and close to one another
in the interior (tile-local top-k makes them differ slightly by design)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from visiontransformer_amd import _lib
from visiontransformer_amd.models import DSWT_DEFAULT, dswt_from_config
from visiontransformer_amd.parallel import tiled_super_resolve

dev = torch.device("cuda:0"); _lib.load()
torch.manual_seed(0)
model = dswt_from_config(DSWT_DEFAULT).to(dev).eval()
lr = torch.rand(1, 3, *bench.SR_LR, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
frames = {}
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for w in (1, 2, 4, 8):
        frames[w] = sum(tiled_super_resolve(model, lr, 4, tile="balanced", halo=32, rank=r, world=w, reduce=False) for r in range(w))
        torch.cuda.synchronize()
        d = (frames[w] - frames[1]).abs()
        print("world", w, "finite", bool(torch.isfinite(frames[w]).all()), "mean |diff to whole frame| %.3e max %.3e of mean |y| %.3e"
              % (float(d.mean()), float(d.max()), float(frames[1].abs().mean())), flush=True)
