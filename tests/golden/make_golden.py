#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Runs only in the build container (needs /root/reference, CPU only).  The
reference's Python never travels: what is committed is this script and the
.npz data it writes (inputs, parameters, Gumbel noise, integer decisions,
outputs, gradients).

* msft_arch.py imports and runs as it is.
* DSwinT_arch.py imports ``torchvision`` and ``torch_scatter``, which are not
  installed here and cannot be.  ``torch_scatter`` names are imported but never
  called; ``torchvision.ops.roi_align`` is called once (DSwinT_arch.py:297).
  Two entries in ``sys.modules`` let the import succeed: roi_align is the
  oracle's restatement of torchvision's published algorithm, so that single op
  is "parity unpinned"; every other line that executes is the reference's own.
* The Gumbel draw inside F.gumbel_softmax is made explicit by seeding torch,
  drawing ``-log(Exp(1))`` of the same shape, then re-seeding before the
  module call: the module's own draw repeats it (DS-MSA draws nothing else
  before it).  The real F.gumbel_softmax runs.

usage:  python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

from oracle import dswin_oracle  # noqa: E402


def _import_reference():
    tv = types.ModuleType("torchvision")
    tv.ops = types.ModuleType("torchvision.ops")
    tv.ops.roi_align = dswin_oracle.roi_align
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.ops", tv.ops)
    ts = types.ModuleType("torch_scatter")
    ts.scatter_max = ts.scatter_mean = None
    sys.modules.setdefault("torch_scatter", ts)
    import DSwinT_arch
    import msft_arch
    return DSwinT_arch, msft_arch


def _randomise(mod, gen, dsmsa=False):
    """Parameters drawn so that every branch is exercised: non-zero biases and
    bias tables, gates on both sides of 0.5, offsets spread over the window."""
    with torch.no_grad():
        for name, p in mod.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=gen) / fan_in ** 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
            if name.endswith("relative_position_bias_table"):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.5)
        if dsmsa:
            last = mod.sparse_discriminator[4]
            last.weight.mul_(6.0)
            last.bias[1] += 0.6


def _save(name, arrays, meta):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB  {meta}")


def _grads(mod, x, y, gen):
    r = torch.randn(y.shape, generator=gen)
    mod.zero_grad()
    (y * r).sum().backward()
    g = {"r": r, "dx": x.grad}
    for n, p in mod.named_parameters():
        g["grad." + n] = p.grad if p.grad is not None else torch.zeros_like(p)
    return g


def _state(mod):
    return {"param." + k: v for k, v in mod.state_dict().items()}


def dswt_wmsa(D, name, B, H, W, C, heads, w, seed):
    gen = torch.Generator().manual_seed(seed)
    mod = D.WindowedMSA(w, C, heads)
    _randomise(mod, gen)
    x = torch.randn(B, H, W, C, generator=gen).requires_grad_()
    y = mod(x)
    arrays = {"x": x, "y": y, **_state(mod), **_grads(mod, x, y, gen)}
    _save(name, arrays, dict(kind="dswt_wmsa", B=B, H=H, W=W, C=C, heads=heads, w=w))


def dswt_dsmsa(D, name, B, H, W, C, heads, w, seed, training, frac=0.25, tau=1.0):
    gen = torch.Generator().manual_seed(seed)
    mod = D.DeformableSparseWindowAttention(w, C, heads)
    _randomise(mod, gen, dsmsa=True)
    mod.num_win_selected = frac
    mod.temperature = tau
    mod.train(training)
    x = torch.randn(B, H, W, C, generator=gen).requires_grad_()
    rec = {}
    orig_cand, orig_cords = mod.candidate_indexes, mod.cords

    def cand(xx, scores):
        out = orig_cand(xx, scores)
        rec.update(cand_b=out[0], cand_h=out[1], cand_w=out[2])
        return out

    def crd(b, h, ww, offsets, HH, WW):
        out = orig_cords(b, h, ww, offsets, HH, WW)
        rec.update(b_idx=b, h_idx=h, w_idx=ww, h1=out[0], w1=out[1])
        return out

    mod.candidate_indexes, mod.cords = cand, crd
    nwin = (H // w) * (W // w)
    torch.manual_seed(seed + 1000)
    noise = -torch.empty(B, nwin).exponential_().log()
    torch.manual_seed(seed + 1000)
    y = mod(x)
    arrays = {"x": x, "y": y, "noise": noise, **_state(mod), **_grads(mod, x, y, gen)}
    arrays.update({"aux." + k: v for k, v in rec.items()})
    npass = int(rec["b_idx"].numel()) if "b_idx" in rec else 0
    _save(name, arrays, dict(kind="dswt_dsmsa", B=B, H=H, W=W, C=C, heads=heads, w=w,
                             training=training, num_win_selected=frac, temperature=tau,
                             n_candidates=int(rec["cand_b"].numel()), n_pass=npass))


def msft_wmsa(M, name, B, H, W, C, heads, w, seed):
    gen = torch.Generator().manual_seed(seed)
    mod = M.WindowedMSA(w, C, heads)
    mod.attn_drop = 0.0          # the reference applies dropout even in eval (msft_arch.py:136)
    _randomise(mod, gen)
    x = torch.randn(B, H * W, C, generator=gen).requires_grad_()
    y = mod(x, H, W)
    arrays = {"x": x, "y": y, **_state(mod), **_grads(mod, x, y, gen)}
    _save(name, arrays, dict(kind="msft_wmsa", B=B, H=H, W=W, C=C, heads=heads, w=w))


def msft_patch(M, name, cls, B, c, H, W, scale, seed):
    gen = torch.Generator().manual_seed(seed)
    merging = cls == "PatchMerging"
    dim = c * scale * scale if merging else c
    mod = getattr(M, cls)(dim, scale)
    _randomise(mod, gen)
    x = torch.randn(B, c, H, W, generator=gen).requires_grad_()
    y = mod(x)
    arrays = {"x": x, "y": y, **_state(mod), **_grads(mod, x, y, gen)}
    _save(name, arrays, dict(kind="msft_" + cls, B=B, c=c, H=H, W=W, scale=scale, dim=dim))


def msft_vitblock(M, name, B, C, H, W, heads, w, seed):
    gen = torch.Generator().manual_seed(seed)
    mod = M.ViTBlock(w, C, n_heads=heads)
    mod.msa.attn_drop = 0.0
    _randomise(mod, gen)
    mod.eval()
    x = torch.randn(B, C, H, W, generator=gen).requires_grad_()
    y = mod(x)
    arrays = {"x": x, "y": y, **_state(mod), **_grads(mod, x, y, gen)}
    _save(name, arrays, dict(kind="msft_vitblock", B=B, C=C, H=H, W=W, heads=heads, w=w))


def dswt_model(D, name, seed, H, W):
    """whole DSWT network, eval mode (the reference's training forward raises NameError at
    DSwinT_arch.py:483); ReZero gates randomised, otherwise every attention branch is times 0."""
    gen = torch.Generator().manual_seed(seed)
    cfg = dict(block_type="default", n_blocks=1, n_sub_blocks=2, levels=1, window_size=8, dim=24, level_dim=12,
               n_heads=2, n_heads_fuse=2, feature_dim=16, scale_factor=4)
    mod = D.MSFTransformer(**cfg)
    _randomise(mod, gen)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if "rezero" in n:
                p.copy_(torch.rand(p.shape, generator=gen) + 0.5)
            if n.endswith("sparse_discriminator.4.weight"):
                p.mul_(6.0)
            if n.endswith("sparse_discriminator.4.bias"):
                p[1] += 0.6
        for m in mod.modules():
            if isinstance(m, D.DeformableSparseWindowAttention):
                m.num_win_selected = 0.5
    mod.eval()
    x = (torch.randn(1, 3, H, W, generator=gen) * 0.5).clamp(-1, 1).requires_grad_()
    y = mod(x)
    arrays = {"x": x, "y": y, **_state(mod), **_grads(mod, x, y, gen)}
    _save(name, arrays, dict(kind="dswt_model", cfg=cfg, H=H, W=W, num_win_selected=0.5))


def dswt_model_default(D, name, seed, H, W):
    """The default-width DSWT network (configs/config.yaml: 8 x 3 blocks, dim 180, 6 heads, w 16; 14.6 M
    parameters) on one 64x64 LR patch, eval mode: the reference's CPU-runnable case.  The parameters are drawn
    per name by seeded_params.fill_parameters, so only x and y are stored."""
    from seeded_params import fill_parameters
    cfg = dict(block_type="default", n_blocks=8, n_sub_blocks=3, levels=1, window_size=16, dim=180, level_dim=96,
               n_heads=6, n_heads_fuse=4, feature_dim=64, scale_factor=4)
    mod = D.MSFTransformer(**cfg)
    fill_parameters(mod, seed)
    mod.eval()
    gen = torch.Generator().manual_seed(seed)
    x = (torch.randn(1, 3, H, W, generator=gen) * 0.5).clamp(-1, 1)
    picked = []
    for m in mod.modules():
        if isinstance(m, D.DeformableSparseWindowAttention):
            orig = m.cords

            def crd(b, h, ww, offsets, HH, WW, orig=orig):
                out = orig(b, h, ww, offsets, HH, WW)
                picked.append(torch.stack([b, h, ww, out[0], out[1]], 1))
                return out
            m.cords = crd
    with torch.no_grad():
        y = mod(x)
    n_win = torch.tensor([p.shape[0] for p in picked])
    _save(name, {"x": x, "y": y, "aux.n_win": n_win, "aux.windows": torch.cat(picked)},
          dict(kind="dswt_model_default", cfg=cfg, H=H, W=W, seed=seed, n_dsmsa_calls=len(picked)))


def msft_model(M, name, seed, H, W):
    gen = torch.Generator().manual_seed(seed)
    cfg = dict(block_type="default", n_blocks=1, n_sub_blocks=1, levels=2, window_size=8, dim=32, level_dim=16,
               n_heads=2, n_heads_fuse=2, feature_dim=16, scale_factor=4)
    mod = M.MSFTransformer(**cfg)
    _randomise(mod, gen)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if "rezero" in n:
                p.copy_(torch.rand(p.shape, generator=gen) + 0.5)
    for m in mod.modules():
        if isinstance(m, M.WindowedMSA):
            m.attn_drop = 0.0
    mod.eval()
    x = (torch.randn(1, 3, H, W, generator=gen) * 0.5).clamp(-1, 1).requires_grad_()
    y = mod(x)
    arrays = {"x": x, "y": y, **_state(mod), **_grads(mod, x, y, gen)}
    _save(name, arrays, dict(kind="msft_model", cfg=cfg, H=H, W=W))


def main():
    torch.set_num_threads(4)
    D, M = _import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "default-width":   # only the newest vector (the others are unchanged)
        dswt_model_default(D, "model_dswt_default", seed=33, H=64, W=64)
        return
    # DSWT WindowedMSA: w=8 and w=16, head_dim 6 / 8 / 30 (the default config's d)
    dswt_wmsa(D, "dswt_wmsa_w8", B=2, H=16, W=24, C=12, heads=2, w=8, seed=1)
    dswt_wmsa(D, "dswt_wmsa_w16", B=1, H=32, W=32, C=24, heads=3, w=16, seed=2)
    dswt_wmsa(D, "dswt_wmsa_d30", B=1, H=16, W=16, C=60, heads=2, w=16, seed=3)
    # DS-MSA: eval / train (Gumbel), low and high window budgets, ragged H != W
    dswt_dsmsa(D, "dswt_dsmsa_eval_w8", B=2, H=32, W=48, C=12, heads=2, w=8, seed=11, training=False)
    dswt_dsmsa(D, "dswt_dsmsa_train_w8", B=2, H=32, W=32, C=12, heads=2, w=8, seed=12, training=True,
               frac=0.5, tau=0.7)
    dswt_dsmsa(D, "dswt_dsmsa_train_w16", B=1, H=64, W=64, C=24, heads=3, w=16, seed=13, training=True,
               frac=0.5)
    dswt_dsmsa(D, "dswt_dsmsa_full_w8", B=1, H=32, W=32, C=16, heads=2, w=8, seed=14, training=True,
               frac=1.0)
    dswt_dsmsa(D, "dswt_dsmsa_d30_w16", B=1, H=32, W=32, C=60, heads=2, w=16, seed=15, training=False,
               frac=0.75)
    # k == 0: the early return at DSwinT_arch.py:273 (2x2 windows * 0.25 -> k = 1 needs 4; use 1x2)
    dswt_dsmsa(D, "dswt_dsmsa_k0", B=1, H=8, W=16, C=12, heads=2, w=8, seed=16, training=False)
    # MSFT
    msft_wmsa(M, "msft_wmsa_w8", B=2, H=16, W=24, C=32, heads=2, w=8, seed=21)
    msft_wmsa(M, "msft_wmsa_d54", B=1, H=16, W=16, C=108, heads=2, w=8, seed=22)
    msft_patch(M, "msft_patch_merging", "PatchMerging", B=2, c=6, H=16, W=24, scale=2, seed=23)
    msft_patch(M, "msft_patch_unmerging", "PatchUnMerging", B=2, c=24, H=8, W=12, scale=2, seed=24)
    msft_vitblock(M, "msft_vitblock", B=1, C=32, H=16, W=16, heads=2, w=8, seed=25)
    # whole networks (the callers of the path), small configs, H/W that need reflect padding
    dswt_model(D, "model_dswt", seed=31, H=20, W=28)
    msft_model(M, "model_msft", seed=32, H=28, W=20)
    dswt_model_default(D, "model_dswt_default", seed=33, H=64, W=64)


if __name__ == "__main__":
    main()
