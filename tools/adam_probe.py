"""Time one optimizer step on the DSWT default network: torch.optim.Adam (multi-tensor), torch's fused Adam, and
the package's one-pass kernel.  Prints host wall time per step (synchronised) and the bytes the update moves."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import models, optim  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = models.DSWTransformer(**models.DSWT_DEFAULT).to(dev)
    params = list(model.parameters())
    numel = sum(p.numel() for p in params)
    for p in params:
        p.grad = torch.randn_like(p) * 1e-3
    print("tensors %d, elements %d, update traffic %.1f MB" % (len(params), numel, numel * 28 / 1e6))
    for name, make in (("torch foreach", lambda: torch.optim.Adam(params, lr=2e-4, betas=(0.9, 0.99), foreach=True)),
                       ("torch fused", lambda: torch.optim.Adam(params, lr=2e-4, betas=(0.9, 0.99), fused=True)),
                       ("one-pass kernel", lambda: optim.Adam(params, lr=2e-4, betas=(0.9, 0.99)))):
        opt = make()
        for _ in range(3):
            opt.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            opt.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(n):
            opt.step()
        ev1.record()
        torch.cuda.synchronize()
        print("%-16s %.3f ms/step wall, %.3f ms/step between events" % (name, ms, ev0.elapsed_time(ev1) / n))


if __name__ == "__main__":
    main()
