"""Step torch.optim.Adam (multi-tensor) and the one-pass kernel side by side from identical states; after every step
report which of p / exp_avg / exp_avg_sq differ, print one offending element with its inputs, then copy torch's state
over the kernel's so that every step starts from the same bits."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import optim  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
n = 1 << 20
for scale in (1.0, 1e-3):
    p_t = torch.nn.Parameter(torch.randn(n, device=dev, generator=gen) * 0.1)
    opt = torch.optim.Adam([p_t], lr=2e-4, betas=(0.9, 0.99), foreach=True)
    p = p_t.detach().clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 7):
        g = torch.randn(n, device=dev, generator=gen) * scale
        p0, m0, v0 = p.clone(), m.clone(), v.clone()
        p_t.grad = g.clone()
        opt.step()
        optim.adam_update_([p], [g], [m], [v], step=step, lr=2e-4, beta1=0.9, beta2=0.99, eps=1e-8)
        st = opt.state[p_t]
        bad = {k: int((a != b).sum()) for k, a, b in (("p", p, p_t.detach()), ("m", m, st["exp_avg"]), ("v", v, st["exp_avg_sq"]))}
        print("scale %g step %d (torch counts %g): differing elements %s" % (scale, step, float(st["step"]), bad))
        for k, a, b in (("m", m, st["exp_avg"]), ("v", v, st["exp_avg_sq"]), ("p", p, p_t.detach())):
            w = (a != b).nonzero()
            if len(w):
                i = int(w[0])
                print("   first %s: kernel %r torch %r | inputs p %r g %r m %r v %r" % (
                    k, a[i].item(), b[i].item(), p0[i].item(), g[i].item(), m0[i].item(), v0[i].item()))
                break
        p.copy_(p_t.detach())
        m.copy_(st["exp_avg"])
        v.copy_(st["exp_avg_sq"])
