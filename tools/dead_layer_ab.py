#!/usr/bin/env python3
"""A/B of the dead-layer early-out of the static DS-MSA path on a 544x960 frame of the DSWT default network (eval, bf16
autocast, ReZero gates at 0.2): with functional.SKIP_DEAD_SCATTER the scatter of a layer in which no window passed
leaves its map unwritten and the closing convolution copies the layer's input instead.  Prints the time per frame
either way, whether the two outputs are bit-equal, and how many of the 24 layers are without a passing window."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visiontransformer_amd import functional as K, models
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = models.dswt_from_config(models.DSWT_DEFAULT).to(dev).eval()
with torch.no_grad():
    for n, p in net.named_parameters():
        if "rezero" in n:
            p.fill_(0.2)
x = torch.rand(1, 3, 544, 960, device=dev)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    outs = {}
    for skip in (True, False, True, False):
        K.SKIP_DEAD_SCATTER = skip
        for _ in range(2):
            y = net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            y = net(x)
        torch.cuda.synchronize()
        outs[skip] = y.float().clone()
        print("frame, SKIP_DEAD_SCATTER=%s: %.2f ms" % (skip, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    print("same output:", torch.equal(outs[True], outs[False]))
    dead = sum(1 for m in net.modules() if hasattr(m, "last_selection") and m.last_selection is not None and int(m.last_selection.n_dev.item()) == 0)
    print("DS-MSA layers without a passing window:", dead)
