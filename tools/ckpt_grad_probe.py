"""Which part of the backward differs between a plain and a whole-model-checkpointed training step?
Runs plain, plain, checkpointed (same seeds) with the stock convolutions in their default and in their
deterministic mode and prints, per setting, the flat-gradient distance and the one-element ReZero gradients
next to the sum of |per-block partial sums| that each of them is a cancelling sum of."""
import os
import sys

import torch
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_trainer import _build  # noqa: E402
from visiontransformer_amd import functional as K  # noqa: E402

dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "dswt"
gen = torch.Generator(device=dev).manual_seed(5)
lr = torch.rand(2, 3, 32, 32, device=dev, generator=gen) * 2 - 1
hr = torch.rand(2, 3, 128, 128, device=dev, generator=gen) * 2 - 1

mags = []
_orig = K._RezeroFn.backward


class _Once:
    """the context with its saved tensors unpacked once (non-reentrant checkpointing allows one unpack)"""

    def __init__(self, ctx):
        object.__setattr__(self, "_ctx", ctx)
        object.__setattr__(self, "saved_tensors", ctx.saved_tensors)

    def __getattr__(self, k):
        return getattr(self._ctx, k)


def _spy(ctx, dy):
    ctx = _Once(ctx)
    out = _orig(ctx, dy)
    b2, r32, *rest = ctx.saved_tensors
    mask = rest[0].float() * ctx.meta[-1] if rest else 1.0
    terms = dy.reshape(b2.shape[0], -1).float() * b2[:, :dy.shape[-1]].float() * mask
    mags.append((float(terms.sum()), float(terms.abs().sum())))
    return out


K._RezeroFn.backward = staticmethod(_spy)


def run(use_ckpt):
    mags.clear()
    model = _build(kind, dev)
    torch.manual_seed(1)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = checkpoint(model, lr, use_reentrant=False) if use_ckpt else model(lr)
        loss = F.l1_loss(y, hr)
    (loss * 65536.0).backward()
    return {n: p.grad.clone() for n, p in model.named_parameters()}, list(mags), float(loss)


def dist(a, b):
    fa = torch.cat([a[n].flatten().double() for n in sorted(a)])
    fb = torch.cat([b[n].flatten().double() for n in sorted(b)])
    return float((fa - fb).norm() / fb.norm())


for det in (False, True):
    torch.backends.cudnn.deterministic = det
    torch.use_deterministic_algorithms(det, warn_only=True)
    g0, m0, l0 = run(False)
    g1, m1, l1 = run(False)
    g2, m2, l2 = run(True)
    print("deterministic=%s losses %r %r %r" % (det, l0, l1, l2))
    print("  flat rel dist plain-plain %.3e  ckpt-plain %.3e" % (dist(g1, g0), dist(g2, g0)))
    f0 = torch.cat([g0[n].flatten().double() for n in sorted(g0)])
    print("  whole-gradient rms %.4e  norm %.4e  numel %d" % (float(f0.norm()) / f0.numel() ** 0.5, float(f0.norm()), f0.numel()))
    nbit = sum(int(not torch.equal(g0[n], g1[n])) for n in g0)
    nbit2 = sum(int(not torch.equal(g0[n], g2[n])) for n in g0)
    print("  tensors not bit-equal: plain-plain %d  ckpt-plain %d of %d" % (nbit, nbit2, len(g0)))
    first = [n for n in g0 if not torch.equal(g0[n], g2[n])]
    print("  last-in-graph tensors that differ ckpt-plain:", first[-6:])
    for n in g0:
        if "rezero" in n:
            a, b, c = (float(g[n]) for g in (g0, g1, g2))
            print("  %-32s plain %.6e plain' %.6e ckpt %.6e  |pp| %.2e |cp| %.2e" %
                  (n, a, b, c, abs(b - a) / max(abs(a), 1e-30), abs(c - a) / max(abs(a), 1e-30)))
    print("  rezero sums vs sum|terms| (backward order):")
    for (s, t), (s2, t2) in zip(m0, m2):
        print("    plain sum %.5e abs %.5e ratio %.1e | ckpt sum %.5e" % (s, t, abs(s) / max(t, 1e-30), s2))
