"""GPU parity tests, kernel by kernel: the HIP path (through the C ABI) against the CPU oracle on
the same seeded inputs.  Bars: integer decisions (selected windows, gate decisions, corners)
bit-exact; values and gradients within 1e-3 relative error for fp32 and 2e-2 for bf16
(relative error = ||a-b|| / ||b||)."""
import pytest
import torch

from conftest import rel_err
from oracle import dswin_oracle as D

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-3, torch.bfloat16: 2e-2}


def _K():
    from visiontransformer_amd import functional as K
    return K


def _selection_from(b_idx, h1, w1, cell, B, Hw, Ww, gd, sdl, dev):
    """build the kernels' window tables from explicit CPU lists"""
    K = _K()
    n = b_idx.numel()
    cellmap = torch.full((B, Hw * Ww), -1, dtype=torch.int32)
    cellmap[b_idx.long(), cell.long()] = torch.arange(n, dtype=torch.int32)
    i32 = lambda t: t.to(torch.int32).to(dev).contiguous()
    return K.Selection(n=n, win_b=i32(b_idx), win_cell=i32(cell), win_h1=i32(h1), win_w1=i32(w1),
                       cellmap=cellmap.view(B, Hw, Ww).to(dev), scores_delta=sdl, gates_delta=gd,
                       cand_idx=None, cand_pass=None, probs=None)


def _random_windows(gen, B, Hw, Ww, w, frac):
    """random passing windows: distinct cells, corners shifted by up to w/2 and clamped"""
    H, W = Hw * w, Ww * w
    bs, cells, h1s, w1s = [], [], [], []
    for b in range(B):
        n = max(1, int(Hw * Ww * frac))
        perm = torch.randperm(Hw * Ww, generator=gen)[:n]
        for c in perm.tolist():
            r, cc = divmod(c, Ww)
            oh = int(torch.randint(-(w // 2), w // 2 + 1, (1,), generator=gen))
            ow = int(torch.randint(-(w // 2), w // 2 + 1, (1,), generator=gen))
            bs.append(b)
            cells.append(c)
            h1s.append(min(max(r * w + oh, 0), H - w))
            w1s.append(min(max(cc * w + ow, 0), W - w))
    t = lambda v: torch.tensor(v, dtype=torch.int64)
    return t(bs), t(cells), t(h1s), t(w1s)


# ----------------------------------------------------------------------------------------------
# (1) selection
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("B,Hw,Ww,w,frac,tau", [(2, 4, 4, 16, 0.25, 1.0), (3, 4, 6, 8, 0.5, 0.7),
                                                (1, 34, 60, 16, 0.25, 1.0), (2, 8, 8, 8, 1.0, 0.3)])
def test_select(cuda0, dtype, training, B, Hw, Ww, w, frac, tau):
    K = _K()
    gen = torch.Generator().manual_seed(Hw * 100 + Ww + int(training))
    n = Hw * Ww
    sd = (torch.randn(B, 4, Hw, Ww, generator=gen) * 2.0).to(dtype)
    sig = sd[:, 1:].sigmoid()
    noise = D.gumbel_noise((B, n), gen) if training else None
    k = int(n * frac)
    ref = D.select_windows(sd.clone().requires_grad_(), sig.clone().requires_grad_(), w, Hw * w, Ww * w, frac, tau,
                           training, noise)
    sd_g = sd.to(cuda0).requires_grad_()
    sig_g = sig.to(cuda0).requires_grad_()
    sel = K.select_windows(sd_g, sig_g, None if noise is None else noise.to(cuda0), tau, k, w, Hw * w, Ww * w)
    # bit-exact integer decisions.  torch.topk leaves the order of EQUAL probabilities unspecified
    # (bf16 scores over thousands of windows do collide), so with ties among the selected values
    # only the selected probabilities can be compared rank by rank; without ties the indices,
    # gate decisions, compaction order and corners must all be identical.
    cand = sel.cand_idx.cpu().long()
    ref_flat = ref["cand_h"] * Ww + ref["cand_w"]
    pr = ref["probs"].detach()
    assert torch.equal(pr.gather(1, cand).reshape(-1), pr.gather(1, ref_flat.view(B, k)).reshape(-1))
    top = torch.topk(pr, min(k + 1, n), dim=1).values
    has_ties = bool((top[:, 1:] == top[:, :-1]).any())
    assert rel_err(sel.probs, ref["probs"]) < 1e-5
    if has_ties:
        assert dtype == torch.bfloat16, "fp32 test inputs are not expected to tie"
        return
    assert torch.equal(cand.reshape(-1), ref_flat)
    assert torch.equal(sel.cand_pass.cpu().bool().reshape(-1), ref["pass_mask"])
    assert sel.n == ref["b_idx"].numel()
    assert torch.equal(sel.win_b.cpu().long(), ref["b_idx"])
    assert torch.equal(sel.win_cell.cpu().long(), ref["h_idx"] * Ww + ref["w_idx"])
    assert torch.equal(sel.win_h1.cpu().long(), ref["h1"])
    assert torch.equal(sel.win_w1.cpu().long(), ref["w1"])
    # cellmap is the inverse of (win_b, win_cell)
    cm = torch.full((B, n), -1, dtype=torch.int32)
    cm[ref["b_idx"], ref["h_idx"] * Ww + ref["w_idx"]] = torch.arange(sel.n, dtype=torch.int32)
    assert torch.equal(sel.cellmap.cpu().view(B, n), cm)
    assert rel_err(sel.probs, ref["probs"]) < 1e-5
    assert rel_err(sel.scores_delta, ref["scores_delta"]) < 1e-6
    assert rel_err(sel.gates_delta, ref["gates_delta"]) < 1e-6
    if sel.n == 0:
        return
    # straight-through gradients
    r1 = torch.randn(sel.n, generator=gen)
    r2 = torch.randn(sel.n, generator=gen)
    sd_c = sd.clone().requires_grad_()
    sig_c = sig.clone().requires_grad_()
    ref2 = D.select_windows(sd_c, sig_c, w, Hw * w, Ww * w, frac, tau, training, noise)
    ((ref2["scores_delta"] * r1).sum() + (ref2["gates_delta"] * r2).sum()).backward()
    ((sel.scores_delta * r1.to(cuda0)).sum() + (sel.gates_delta * r2.to(cuda0)).sum()).backward()
    tol = TOL[dtype]
    assert rel_err(sd_g.grad, sd_c.grad) < tol
    assert rel_err(sig_g.grad, sig_c.grad) < tol
    assert torch.count_nonzero(sig_g.grad[:, 1:]) == 0  # offsets: no gradient, as in the reference


def test_select_tie_rule(cuda0):
    """equal logits: the lower window index ranks first (torch.topk leaves ties unspecified,
    so exact ties are outside the bit-exact contract; this pins the kernel's own rule)."""
    K = _K()
    sd = torch.zeros(1, 4, 2, 4)
    sd[0, 0] = torch.tensor([[1.0, 3.0, 3.0, 0.0], [3.0, -1.0, 1.0, 1.0]])
    sd[0, 1] = 5.0
    sig = sd[:, 1:].sigmoid()
    sel = K.select_windows(sd.to(cuda0), sig.to(cuda0), None, 1.0, 5, 8, 16, 32)
    assert sel.cand_idx.cpu().view(-1).tolist() == [1, 2, 4, 0, 6]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("training", [False, True])
def test_select_at_the_lowest_temperature_with_wide_logits(cuda0, dtype, training):
    """The end of the reference's annealing (temperature 0.1) with logits spread over +-20: logits / tau spans
    hundreds, most probabilities underflow to exactly 0 in fp32 and DISTINCT logits tie there, where torch.topk's
    order is unspecified.  The kernel ranks the logits themselves.  Whatever the ties: its k windows must be a
    valid top-k of the reference's probabilities (nothing left out beats anything taken), rank by rank the same
    probabilities, and identical windows wherever the probabilities around them are distinct.  The test also
    counts how many images tie inside their top-k + 1 (and requires some, so that it tests what it says)."""
    K = _K()
    B, Hw, Ww, w, frac, tau = 16, 8, 8, 16, 0.25, 0.1
    n, k = Hw * Ww, int(Hw * Ww * frac)
    gen = torch.Generator().manual_seed(77 + int(training))
    sd = (torch.randn(B, 4, Hw, Ww, generator=gen) * torch.tensor([8.0, 2.0, 2.0, 2.0]).view(1, 4, 1, 1)).to(dtype)
    sig = sd[:, 1:].sigmoid()
    noise = D.gumbel_noise((B, n), gen) if training else None
    ref = D.select_windows(sd.clone(), sig.clone(), w, Hw * w, Ww * w, frac, tau, training, noise)
    sel = K.select_windows(sd.to(cuda0), sig.to(cuda0), None if noise is None else noise.to(cuda0), tau, k, w, Hw * w, Ww * w)
    pr = ref["probs"].detach()
    cand = sel.cand_idx.cpu().long().view(B, k)
    ref_flat = (ref["cand_h"] * Ww + ref["cand_w"]).view(B, k)
    taken = pr.gather(1, cand)
    assert torch.equal(taken, pr.gather(1, ref_flat))                       # rank by rank the same probabilities
    left = pr.clone().scatter_(1, cand, -1.0)
    assert bool((left.max(dim=1).values <= taken.min(dim=1).values).all())  # a valid top-k set
    top = torch.topk(pr, k + 1, dim=1).values
    tied_rows = (top[:, 1:] == top[:, :-1]).any(dim=1)
    assert int(tied_rows.sum()) > 0, "no image ties at this temperature: widen the logits"
    # within an image, a rank whose probability differs from both neighbours' is decided: same window
    decided = torch.ones(B, k, dtype=torch.bool)
    decided[:, 1:] &= top[:, 1:k] != top[:, :k - 1]
    decided &= top[:, :k] != top[:, 1:k + 1]
    assert torch.equal(cand[decided], ref_flat[decided])
    assert bool(decided[~tied_rows].all())
    # where the probabilities tie at zero the kernel still follows the logits (plus noise): the larger one first
    z = sd[:, 0].reshape(B, n).float() + (noise if noise is not None else 0.0)
    zt = z.gather(1, cand)
    # (the division by tau is rounded to the activation dtype, as the reference rounds it: logits that meet in one
    # bf16 quotient rank by window index, so the order of the raw logits holds to one ulp of that dtype)
    slack = 1e-4 if dtype == torch.float32 else 2.0 ** -7 * zt[:, :-1].abs() + 1e-4
    assert bool((zt[:, 1:] <= zt[:, :-1] + slack).all())


# ----------------------------------------------------------------------------------------------
# (2) gather and (4) scatter-mean
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Hw,Ww,w,C,frac", [(2, 3, 4, 8, 12, 0.5), (1, 4, 4, 16, 180, 1.0), (2, 2, 5, 16, 24, 1.0)])
def test_gather_fwd_bwd(cuda0, dtype, B, Hw, Ww, w, C, frac):
    K = _K()
    gen = torch.Generator().manual_seed(7 + C)
    H, W = Hw * w, Ww * w
    b_idx, cell, h1, w1 = _random_windows(gen, B, Hw, Ww, w, frac)
    n = b_idx.numel()
    x = torch.randn(B, H, W, C, generator=gen)
    r = torch.randn(n, w * w, C, generator=gen)
    xc = x.clone().requires_grad_()
    ref = D.gather_windows(xc.permute(0, 3, 1, 2).contiguous(), b_idx, h1, w1, w).permute(0, 2, 3, 1).reshape(n, w * w, C)
    (ref * r).sum().backward()
    sel = _selection_from(b_idx, h1, w1, cell, B, Hw, Ww, None, None, cuda0)
    xg = x.to(cuda0).to(dtype).requires_grad_()
    out = K.gather_windows(xg, sel, w)
    (out.float() * r.to(cuda0)).sum().backward()
    tol = TOL[dtype]
    assert rel_err(out, ref) < tol
    assert rel_err(xg.grad, xc.grad) < tol
    if dtype == torch.float32:
        assert (out.cpu() - ref).abs().max() < 1e-5
    # fp32 input, bf16 output (the autocast case): one rounding of the fp32 result
    if dtype == torch.float32:
        out16 = K.gather_windows(x.to(cuda0), sel, w, out_dtype=torch.bfloat16)
        assert torch.equal(out16.cpu(), out.detach().cpu().to(torch.bfloat16))


@pytest.mark.parametrize("after_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Hw,Ww,w,C,frac", [(2, 3, 4, 8, 12, 0.5), (1, 4, 4, 16, 180, 1.0), (2, 2, 5, 16, 24, 1.0),
                                              (1, 1, 1, 8, 8, 1.0)])
def test_scatter_fwd_bwd(cuda0, after_dtype, B, Hw, Ww, w, C, frac):
    K = _K()
    gen = torch.Generator().manual_seed(11 + C)
    H, W = Hw * w, Ww * w
    b_idx, cell, h1, w1 = _random_windows(gen, B, Hw, Ww, w, frac)
    n = b_idx.numel()
    x = torch.randn(B, H, W, C, generator=gen)
    after = torch.randn(n, w * w, C, generator=gen).to(after_dtype)
    gd = torch.rand(n, generator=gen) + 0.5
    sdl = torch.rand(n, generator=gen) + 0.5
    r = torch.randn(B, H, W, C, generator=gen)
    leaves = [t.clone().float().requires_grad_() for t in (x, after, gd, sdl)]
    ref = D.scatter_mean_residual(leaves[0], leaves[1], leaves[2], leaves[3], b_idx, h1, w1, w)
    (ref * r).sum().backward()
    gl = [t.to(cuda0).requires_grad_() for t in (x, after, gd, sdl)]
    sel = _selection_from(b_idx, h1, w1, cell, B, Hw, Ww, gl[2], gl[3], cuda0)
    y = K.scatter_mean_residual(gl[0], gl[1], sel, w)
    (y * r.to(cuda0)).sum().backward()
    tol = TOL[after_dtype]
    assert rel_err(y, ref) < (1e-6 if after_dtype == torch.float32 else tol)
    for got, want, name in zip(gl, leaves, ("x", "after", "gates", "scores")):
        assert rel_err(got.grad, want.grad) < tol, name
    # atomic-free => bitwise reproducible
    y2 = K.scatter_mean_residual(gl[0].detach(), gl[1].detach(), sel, w)
    assert torch.equal(y2, y.detach())


# ----------------------------------------------------------------------------------------------
# (3) fused window attention
# ----------------------------------------------------------------------------------------------
ATTN_CASES = [
    # w, C, heads, layout, grid (B,H,W) or number of packed windows
    (16, 180, 6, "interleaved", (1, 32, 48)),
    (16, 180, 6, "chunked", 5),
    (8, 180, 6, "interleaved", (2, 16, 24)),
    (8, 24, 2, "chunked", 3),
    (16, 48, 3, "chunked", 2),      # head dim 16
    (8, 108, 2, "chunked", (1, 16, 8)),  # head dim 54 -> padded to 64
    (16, 21, 3, "interleaved", 2),  # odd head dim 7
    (16, 108, 2, "chunked", 2),     # 16x16 windows with head dim 54 -> padded to 64
    (16, 96, 3, "interleaved", (1, 16, 32)),  # head dim 32: no padding column left for the row constants
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("bias", [True, False])
@pytest.mark.parametrize("w,C,heads,layout,grid", ATTN_CASES)
def test_window_attention(cuda0, dtype, bias, w, C, heads, layout, grid):
    K = _K()
    from visiontransformer_amd._lib import QKV_CHUNKED, QKV_INTERLEAVED
    if dtype == torch.float32 and w == 16 and C // heads > 32:
        pytest.skip("fp32 operands of 16x16 windows at head dims above 32 do not fit a CU's LDS (DSW_EUNSUPPORTED)")
    gen = torch.Generator().manual_seed(C + w)
    L = w * w
    shape = (grid[0], grid[1], grid[2], 3 * C) if isinstance(grid, tuple) else (grid, L, 3 * C)
    qkv = torch.randn(shape, generator=gen).to(dtype)
    table = (torch.randn((2 * w - 1) ** 2, heads, generator=gen) * 0.5) if bias else None
    r = torch.randn(shape[:-1] + (C,), generator=gen)
    q_ref = qkv.float().clone().requires_grad_()
    t_ref = None if table is None else table.clone().requires_grad_()
    ref = D.window_attention_core(q_ref, t_ref, w, heads, layout, grid=grid if isinstance(grid, tuple) else None)
    ref = ref.reshape(r.shape)
    (ref * r).sum().backward()
    q_g = qkv.to(cuda0).requires_grad_()
    t_g = None if table is None else table.to(cuda0).requires_grad_()
    out = K.window_attention(q_g, t_g, w, heads, QKV_CHUNKED if layout == "chunked" else QKV_INTERLEAVED,
                             grid=grid if isinstance(grid, tuple) else None)
    assert out.dtype == dtype and out.shape == r.shape
    (out.float() * r.to(cuda0)).sum().backward()
    tol = TOL[dtype]
    assert rel_err(out, ref) < tol
    assert rel_err(q_g.grad, q_ref.grad) < tol
    if bias:
        assert rel_err(t_g.grad, t_ref.grad) < tol
    # bitwise reproducible backward (no unordered float atomics)
    q2 = qkv.to(cuda0).requires_grad_()
    t2 = None if table is None else table.to(cuda0).requires_grad_()
    out2 = K.window_attention(q2, t2, w, heads, QKV_CHUNKED if layout == "chunked" else QKV_INTERLEAVED,
                              grid=grid if isinstance(grid, tuple) else None)
    (out2.float() * r.to(cuda0)).sum().backward()
    assert torch.equal(out2, out) and torch.equal(q2.grad, q_g.grad)
    if bias:
        assert torch.equal(t2.grad, t_g.grad)


PROJ_CASES = [
    # w, C, heads, layout, grid (B,H,W) or number of packed windows; head dim <= 32
    (16, 180, 6, "interleaved", (1, 32, 48)),
    (16, 180, 6, "chunked", 5),
    (8, 180, 6, "interleaved", (2, 16, 24)),
    (8, 24, 2, "chunked", 3),
    (16, 48, 3, "chunked", 2),                 # head dim 16
    (16, 96, 3, "interleaved", (1, 16, 32)),   # head dim 32: no padding column for the ones column of V
    (16, 256, 8, "chunked", 1),                # 256 input channels: the weight rows do not fit in Q's and K's LDS space
]


@pytest.mark.parametrize("bias,lin_bias", [(True, True), (False, True), (True, False)])
@pytest.mark.parametrize("w,C,heads,layout,grid", PROJ_CASES)
def test_projected_window_attention(cuda0, bias, lin_bias, w, C, heads, layout, grid):
    """q|k|v projection inside the attention kernel (DSwinT_arch.py:334 + :135-172 / :303) against the oracle's
    attention of the projected rows, and against the two separate kernels it replaces"""
    K = _K()
    from visiontransformer_amd._lib import QKV_CHUNKED, QKV_INTERLEAVED
    gen = torch.Generator().manual_seed(7 * C + w)
    L = w * w
    lead = (grid[0], grid[1], grid[2]) if isinstance(grid, tuple) else (grid, L)
    x = torch.randn(lead + (C,), generator=gen).bfloat16()
    lin = torch.nn.Linear(C, 3 * C, bias=lin_bias)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(3 * C, C, generator=gen) * C ** -0.5)
        if lin_bias:
            lin.bias.copy_(torch.randn(3 * C, generator=gen) * 0.5)
    table = (torch.randn((2 * w - 1) ** 2, heads, generator=gen) * 0.5) if bias else None
    # the oracle on what the kernels compute with: bf16 activations and weights, the projection rounded to bf16
    qkv_ref = (x.float() @ lin.weight.detach().bfloat16().float().t() + (lin.bias.detach() if lin_bias else 0.0)).bfloat16().float()
    ref = D.window_attention_core(qkv_ref, table, w, heads, layout, grid=grid if isinstance(grid, tuple) else None).reshape(lead + (C,))
    lin = lin.to(cuda0)
    code = QKV_CHUNKED if layout == "chunked" else QKV_INTERLEAVED
    g = grid if isinstance(grid, tuple) else None
    t_g = None if table is None else table.to(cuda0)
    xp = K.cast_rows(x.to(cuda0).reshape(-1, C), torch.bfloat16, K._ceil_pad(C))       # zero-padded rows, as the norm kernels leave them
    xv = K._mark_padded(xp)[:, :C].view(lead + (C,))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out = K.projected_window_attention(xv, lin, t_g, w, heads, code, grid=g)
        assert out is not None and out.dtype == torch.bfloat16 and out.shape == lead + (C,)
        sep = K.window_attention(K.linear(xv, lin, pad_out=True), t_g, w, heads, code, grid=g)
        padded = K.projected_window_attention(xv, lin, t_g, w, heads, code, grid=g, pad_out=True)
    assert rel_err(out, ref) < TOL[torch.bfloat16]
    assert rel_err(out, sep) < 4e-3                                   # one bf16 ulp: same operands, same roundings
    assert torch.equal(padded, out)
    if C % 64:
        assert float(padded._base[:, C:].abs().max()) == 0.0
    with torch.autocast("cuda", dtype=torch.bfloat16):                # with autograd on, the fused call declines
        assert K.projected_window_attention(xv, lin, t_g, w, heads, code, grid=g) is None


@pytest.mark.parametrize("bias", [True, False])
@pytest.mark.parametrize("C,heads,layout,grid", [(180, 6, "interleaved", (2, 32, 48)), (180, 6, "chunked", 7), (48, 3, "chunked", 3),
                                                 (108, 2, "interleaved", (1, 32, 16))])
def test_window_attention_backward_kernels_agree(cuda0, monkeypatch, bias, C, heads, layout, grid):
    """16x16 windows in bf16 have two backward kernels: the two-phase one (no dS exchange between waves) and the
    general one (DSW_ATTN_BWD_GENERAL=1).  Same gradients within bf16 rounding of the P / dS operands, and each
    against the fp32 oracle."""
    K = _K()
    from visiontransformer_amd._lib import QKV_CHUNKED, QKV_INTERLEAVED
    w = 16
    gen = torch.Generator().manual_seed(C + len(layout))
    shape = (grid[0], grid[1], grid[2], 3 * C) if isinstance(grid, tuple) else (grid, w * w, 3 * C)
    qkv = torch.randn(shape, generator=gen).to(torch.bfloat16)
    table = (torch.randn((2 * w - 1) ** 2, heads, generator=gen) * 0.5) if bias else None
    r = torch.randn(shape[:-1] + (C,), generator=gen)
    q_ref = qkv.float().clone().requires_grad_()
    t_ref = None if table is None else table.clone().requires_grad_()
    ref = D.window_attention_core(q_ref, t_ref, w, heads, layout, grid=grid if isinstance(grid, tuple) else None).reshape(r.shape)
    (ref * r).sum().backward()
    res = []
    # head dim 30 has a kernel of its own (DSW_ATTN_BWD_W16_OLD=1 keeps it on the one for any even head dim)
    for general, old in (("0", "0"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("DSW_ATTN_BWD_GENERAL", general)
        monkeypatch.setenv("DSW_ATTN_BWD_W16_OLD", old)
        q_g = qkv.to(cuda0).requires_grad_()
        t_g = None if table is None else table.to(cuda0).requires_grad_()
        out = K.window_attention(q_g, t_g, w, heads, QKV_CHUNKED if layout == "chunked" else QKV_INTERLEAVED,
                                 grid=grid if isinstance(grid, tuple) else None)
        (out.float() * r.to(cuda0)).sum().backward()
        assert rel_err(q_g.grad, q_ref.grad) < TOL[torch.bfloat16]
        if bias:
            assert rel_err(t_g.grad, t_ref.grad) < TOL[torch.bfloat16]
        res.append((q_g.grad, None if t_g is None else t_g.grad))
    for other in (1, 2):
        assert rel_err(res[0][0], res[other][0]) < 1e-2
        if bias:
            assert rel_err(res[0][1], res[other][1]) < 1e-2


@pytest.mark.parametrize("bias", [True, False])
@pytest.mark.parametrize("C,heads,layout,grid", [(180, 6, "interleaved", (2, 32, 48)), (180, 6, "chunked", 7), (48, 3, "chunked", 3),
                                                 (84, 3, "interleaved", (1, 16, 32))])
def test_window_attention_forward_kernels_agree(cuda0, monkeypatch, bias, C, heads, layout, grid):
    """16x16 windows in bf16 with an even head dim <= 30 have two forward kernels: the one that keeps Q in registers
    and the softmax reference in the head dim's padding (attn_fwd_w16.hip) and the general one
    (DSW_ATTN_FWD_GENERAL=1).  Same output and log-sum-exp within bf16 rounding, on dense rows and on padded ones."""
    K = _K()
    from visiontransformer_amd._lib import QKV_CHUNKED, QKV_INTERLEAVED
    w = 16
    gen = torch.Generator().manual_seed(3 * C + len(layout))
    lead = (grid[0], grid[1], grid[2]) if isinstance(grid, tuple) else (grid, w * w)
    qkv = (torch.randn(lead + (3 * C,), generator=gen) * 1.5).to(torch.bfloat16)
    table = (torch.randn((2 * w - 1) ** 2, heads, generator=gen) * 0.5) if bias else None
    g = grid if isinstance(grid, tuple) else None
    ref = D.window_attention_core(qkv.float(), table, w, heads, layout, grid=g).reshape(lead + (C,))
    code = QKV_CHUNKED if layout == "chunked" else QKV_INTERLEAVED
    t_g = None if table is None else table.to(cuda0)
    rows = qkv.numel() // (3 * C)
    buf = K._mark_padded(torch.zeros(rows, K._ceil_pad(3 * C), dtype=torch.bfloat16, device=cuda0))
    buf[:, :3 * C] = qkv.to(cuda0).reshape(rows, 3 * C)
    res = []
    for general in ("0", "1"):
        monkeypatch.setenv("DSW_ATTN_FWD_GENERAL", general)
        with torch.no_grad():
            dense = K.window_attention(qkv.to(cuda0), t_g, w, heads, code, grid=g)
            padded = K.window_attention(buf[:, :3 * C].view(lead + (3 * C,)), t_g, w, heads, code, grid=g, pad_out=True)
        assert rel_err(dense, ref) < TOL[torch.bfloat16]
        assert torch.equal(dense, padded)
        if C % 64:
            assert float(padded._base[:, C:].abs().max()) == 0.0
        res.append(dense)
    assert rel_err(res[0], res[1]) < 6e-3


@pytest.mark.parametrize("pattern", ["rising", "falling", "spike", "huge", "negative"])
def test_window_attention_w16_softmax_reference(cuda0, pattern):
    """The forward kernel for 16x16 windows raises its softmax reference only when a key tile exceeds it by 2^8:
    logits that grow from tile to tile, fall, spike in one late key, are huge everywhere or are all very negative
    must give the oracle's softmax, finite, and a backward that agrees with it (log-sum-exp saved by the forward)."""
    K = _K()
    from visiontransformer_amd._lib import QKV_CHUNKED
    w, C, heads = 16, 60, 2
    gen = torch.Generator().manual_seed(11)
    qkv = torch.randn(3, w * w, 3 * C, generator=gen)
    ramp = torch.linspace(0.0, 1.0, w * w).view(1, -1, 1)
    k = qkv[..., C:2 * C]
    if pattern == "rising":
        k *= 1.0 + 14.0 * ramp
    elif pattern == "falling":
        k *= 15.0 - 14.0 * ramp
    elif pattern == "spike":
        k[:, 201] *= 40.0
    elif pattern == "huge":
        qkv[..., :2 * C] *= 9.0
    else:
        qkv[..., :C] = qkv[..., :C].abs() * 6.0
        k.copy_(-k.abs() * 6.0)
    qkv = qkv.to(torch.bfloat16)
    r = torch.randn(3, w * w, C, generator=gen)
    q_ref = qkv.float().clone().requires_grad_()
    ref = D.window_attention_core(q_ref, None, w, heads, "chunked")
    (ref.reshape(r.shape) * r).sum().backward()
    q_g = qkv.to(cuda0).requires_grad_()
    out = K.window_attention(q_g, None, w, heads, QKV_CHUNKED)
    (out.float() * r.to(cuda0)).sum().backward()
    assert torch.isfinite(out).all() and torch.isfinite(q_g.grad).all()
    assert rel_err(out, ref.reshape(r.shape)) < TOL[torch.bfloat16]
    # ("huge": logits of several hundred, rows all but one-hot; the gradient that is left is the difference of nearly
    # equal bf16-rounded terms, in either forward kernel)
    assert rel_err(q_g.grad, q_ref.grad) < (8e-2 if pattern == "huge" else 3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("w,C,heads,layout,grid,bias,p", [
    (8, 96, 6, "chunked", (2, 16, 24), False, 0.1),      # msft level block: head dim 16, SDPA's dropout_p
    (8, 108, 2, "chunked", 3, False, 0.3),               # head dim 54
    (16, 180, 6, "interleaved", (1, 32, 32), True, 0.2), # DSWT WindowedMSA shape with its bias table
    (16, 48, 3, "chunked", 2, True, 0.5),
])
def test_window_attention_dropout(cuda0, dtype, w, C, heads, layout, grid, bias, p):
    """Attention dropout inside the kernels (DSwinT_arch.py:172, msft_arch.py:136) against the oracle given the
    SAME keep mask (oracle.dropout_keep_mask restates the rule of include/dswin_hip.h): output, dqkv and the
    bias-table gradient; a different seed gives a different result; p = 0 is the plain kernel."""
    K = _K()
    from visiontransformer_amd._lib import QKV_CHUNKED, QKV_INTERLEAVED
    gen = torch.Generator().manual_seed(C + w + int(100 * p))
    L = w * w
    shape = (grid[0], grid[1], grid[2], 3 * C) if isinstance(grid, tuple) else (grid, L, 3 * C)
    n_win = grid[0] * (grid[1] // w) * (grid[2] // w) if isinstance(grid, tuple) else grid
    qkv = torch.randn(shape, generator=gen).to(dtype)
    table = (torch.randn((2 * w - 1) ** 2, heads, generator=gen) * 0.5) if bias else None
    r = torch.randn(shape[:-1] + (C,), generator=gen)
    seed = 0x0123456789ABCDEF + w
    keep = D.dropout_keep_mask(n_win, heads, L, p, seed)
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.02
    q_ref = qkv.float().clone().requires_grad_()
    t_ref = None if table is None else table.clone().requires_grad_()
    g_ = grid if isinstance(grid, tuple) else None
    ref = D.window_attention_core(q_ref, t_ref, w, heads, layout, grid=g_, attn_drop=p, keep=keep).reshape(r.shape)
    (ref * r).sum().backward()
    lay = QKV_CHUNKED if layout == "chunked" else QKV_INTERLEAVED
    q_g = qkv.to(cuda0).requires_grad_()
    t_g = None if table is None else table.to(cuda0).requires_grad_()
    out = K.window_attention(q_g, t_g, w, heads, lay, grid=g_, dropout_p=p, dropout_seed=seed)
    (out.float() * r.to(cuda0)).sum().backward()
    tol = TOL[dtype]
    assert rel_err(out, ref) < tol
    assert rel_err(q_g.grad, q_ref.grad) < tol
    if bias:
        assert rel_err(t_g.grad, t_ref.grad) < tol
    with torch.no_grad():
        other = K.window_attention(qkv.to(cuda0), None if table is None else table.to(cuda0), w, heads, lay, grid=g_,
                                   dropout_p=p, dropout_seed=seed + 1)
        same = K.window_attention(qkv.to(cuda0), None if table is None else table.to(cuda0), w, heads, lay, grid=g_,
                                  dropout_p=p, dropout_seed=seed)
        plain = K.window_attention(qkv.to(cuda0), None if table is None else table.to(cuda0), w, heads, lay, grid=g_)
    assert torch.equal(same, out.detach()) and rel_err(other, out) > 0.05 and rel_err(plain, out) > 0.05


def test_window_attention_large_logits(cuda0):
    """softmax stability: logits of a few hundred must not overflow (row max is subtracted)."""
    K = _K()
    from visiontransformer_amd._lib import QKV_CHUNKED
    gen = torch.Generator().manual_seed(5)
    qkv = torch.randn(2, 64, 3 * 32, generator=gen) * 12.0
    ref = D.window_attention_core(qkv, None, 8, 2, "chunked")
    out = K.window_attention(qkv.to(cuda0), None, 8, 2, QKV_CHUNKED)
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < 1e-3


def test_unsupported_arguments_fail_loudly(cuda0):
    K = _K()
    from visiontransformer_amd._lib import QKV_CHUNKED
    with pytest.raises(RuntimeError, match="window size"):
        K.window_attention(torch.zeros(1, 16, 3 * 8, device=cuda0), None, 4, 2, QKV_CHUNKED)
    with pytest.raises(RuntimeError, match="GPU only"):
        K.window_attention(torch.zeros(1, 64, 3 * 8), None, 8, 2, QKV_CHUNKED)


# ----------------------------------------------------------------------------------------------
# NHWC convolutions on the path
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C", [(2, 16, 24, 12), (1, 32, 32, 180), (1, 5, 70, 8)])
def test_dwconv3x3(cuda0, dtype, B, H, W, C):
    K = _K()
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(C)
    x = torch.randn(B, H, W, C, generator=gen)
    wgt = torch.randn(C, 1, 3, 3, generator=gen) * 0.3
    bias = torch.randn(C, generator=gen) * 0.1
    r = torch.randn(B, H, W, C, generator=gen)
    leaves = [t.clone().requires_grad_() for t in (x, wgt, bias)]
    ref = F.conv2d(leaves[0].permute(0, 3, 1, 2), leaves[1], leaves[2], 1, 1, groups=C).permute(0, 2, 3, 1)
    (ref * r).sum().backward()
    gl = [x.to(cuda0).to(dtype).requires_grad_(), wgt.to(cuda0).requires_grad_(), bias.to(cuda0).requires_grad_()]
    y = K.dwconv3x3_nhwc(gl[0], gl[1], gl[2])
    (y.float() * r.to(cuda0)).sum().backward()
    tol = TOL[dtype]
    assert rel_err(y, ref) < tol
    for got, want, name in zip(gl, leaves, ("x", "weight", "bias")):
        assert rel_err(got.grad, want.grad) < tol, name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C,w", [(2, 16, 24, 12, 8), (1, 64, 64, 180, 16), (2, 32, 16, 24, 16)])
def test_sparse_discriminator(cuda0, dtype, B, H, W, C, w):
    K = _K()
    gen = torch.Generator().manual_seed(C + w)
    x = torch.randn(B, H, W, C, generator=gen)
    convs = torch.nn.Sequential(torch.nn.Conv2d(C, 4, 3, 1, 1), torch.nn.SiLU(), torch.nn.Conv2d(4, 4, w, stride=w),
                                torch.nn.SiLU(), torch.nn.Conv2d(4, 4, 3, 1, 1))
    with torch.no_grad():
        for p in convs.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * (0.3 if p.dim() == 1 else (2.0 / p[0].numel()) ** 0.5))
    P = {"sparse_discriminator." + k: v.detach().clone().requires_grad_() for k, v in convs.state_dict().items()}
    xc = x.clone().requires_grad_()
    ref = D.sparse_discriminator(P, xc.permute(0, 3, 1, 2).contiguous(), w)
    r = torch.randn(ref.shape, generator=gen)
    (ref * r).sum().backward()
    convs = convs.to(cuda0)
    xg = x.to(cuda0).to(dtype).requires_grad_()
    sd = K.sparse_discriminator_nhwc(xg, (convs[0], convs[2], convs[4]), w)
    assert sd.shape == ref.shape
    (sd.float() * r.to(cuda0)).sum().backward()
    tol = TOL[dtype]
    assert rel_err(sd, ref) < tol
    assert rel_err(xg.grad, xc.grad) < tol
    for k, p in convs.state_dict(keep_vars=True).items():
        assert rel_err(p.grad, P["sparse_discriminator." + k].grad) < tol, k


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(1, 4), (131, 180), (4096 + 77, 540), (1000, 2048)])
def test_colsum(cuda0, rows, cols, dtype):
    """Column sums (Linear bias gradients) against torch, and bitwise reproducible."""
    from visiontransformer_amd import functional as K
    g = torch.Generator(device=cuda0).manual_seed(rows * 7 + cols)
    x = torch.randn(rows, cols, device=cuda0, generator=g).to(dtype)
    got = K.colsum(x)
    ref = x.double().sum(0)
    assert got.dtype == torch.float32 and got.shape == (cols,)
    assert rel_err(got, ref.float()) < 1e-5
    assert torch.equal(got, K.colsum(x))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_select_in_kernel_sigmoid(cuda0, dtype):
    """sig = None (sigmoids and their backward inside the kernels) against the explicit
    sd[:, 1:].sigmoid() tensor: same windows bit for bit, same gradient of sd."""
    from visiontransformer_amd import functional as K
    g = torch.Generator(device=cuda0).manual_seed(11)
    B, Hw, Ww, w, k = 5, 6, 4, 8, 7
    sd0 = (torch.randn(B, 4, Hw, Ww, device=cuda0, generator=g) * 2).to(dtype)
    noise = -torch.empty(B, Hw * Ww, device=cuda0).exponential_(generator=g).log()
    outs = []
    for fused in (False, True):
        sd = sd0.clone().requires_grad_()
        sig = None if fused else sd[:, 1:].sigmoid()
        sel = K.select_windows(sd, sig, noise, 0.7, k, w, Hw * w, Ww * w)
        assert sel.n > 0
        wg = torch.linspace(0.5, 1.5, sel.n, device=cuda0)
        ((sel.scores_delta * wg).sum() + (sel.gates_delta * wg.flip(0)).sum()).backward()
        outs.append((sel, sd.grad))
    (a, ga), (b, gb) = outs
    for f in ("win_b", "win_cell", "win_h1", "win_w1", "cellmap", "cand_idx", "cand_pass"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    assert torch.equal(a.scores_delta, b.scores_delta) and torch.equal(a.gates_delta, b.gates_delta)
    assert rel_err(gb.float(), ga.float()) < (1e-5 if dtype == torch.float32 else 2e-2)


def _ceil64(n):
    return (n + 63) // 64 * 64


def _weight_ops(w, dev):
    from visiontransformer_amd import _lib
    N, K = w.shape
    wp = torch.empty(_ceil64(N), _ceil64(K), device=dev, dtype=torch.bfloat16)
    wpt = torch.empty(_ceil64(K), _ceil64(N), device=dev, dtype=torch.bfloat16)
    _lib.call("dsw_weight_prep", _lib.ptr(w), _lib.dtype_code(w.dtype), N, K, _lib.ptr(wp), _lib.ptr(wpt), _lib.stream(dev))
    return wp, wpt


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,a_dtype,c_dtype,pad_a,pad_c,copy", [
    (300, 180, 540, torch.float32, torch.bfloat16, False, True, True),    # qkv from the fp32 stream
    (300, 96, 288, torch.bfloat16, torch.bfloat16, True, True, False),    # padded bf16 rows in, K slab of 128
    (257, 64, 64, torch.float32, torch.float32, False, False, True),      # ragged last row block, K slab of 64
    (1000, 544, 180, torch.bfloat16, torch.float32, True, False, False),  # 3 slabs: accumulators persist
    (1000, 256, 320, torch.bfloat16, torch.bfloat16, True, True, False),  # 2 slabs, two block columns
    (515, 320, 100, torch.float32, torch.bfloat16, False, False, True),   # 3 slabs of 128, the last half empty; dense ragged output
    (700, 432, 216, torch.bfloat16, torch.bfloat16, True, True, False),   # msft MLP: 4 slabs of 128 (448), four persistent chunks
    (700, 648, 216, torch.float32, torch.float32, False, False, True),    # 6 slabs of 128 (704); the copy's rows end at 704
    (900, 216, 648, torch.bfloat16, torch.bfloat16, True, True, False),   # 2 slabs, 11 chunks in passes of four
    (128 * 40 + 1, 180, 180, torch.bfloat16, torch.bfloat16, True, True, False),
    (128 * 9 + 77, 180, 180, torch.bfloat16, torch.float32, False, False, True),     # dense bf16 rows of 360 bytes: 8-byte loads
    (128 * 300, 180, 540, torch.float32, torch.bfloat16, False, True, True),
])
def test_linear_kernel(cuda0, M, K, N, a_dtype, c_dtype, pad_a, pad_c, copy):
    """dsw_linear / dsw_weight_prep through the C ABI against a torch fp32 product of the same
    bf16-rounded operands: 1e-2 of the largest value for bf16 results (their own rounding), 1e-5 * K
    for fp32 results; padding columns of the result and of the bf16 copy of A exactly zero, the copy
    bit-exact."""
    from visiontransformer_amd import _lib
    bf = torch.bfloat16
    g = torch.Generator(device=cuda0).manual_seed(M + K + N)
    w = torch.randn(N, K, device=cuda0, generator=g) * 0.1
    bias = torch.randn(N, device=cuda0, generator=g)
    wp, wpt = _weight_ops(w, cuda0)
    assert torch.equal(wp[:N, :K], w.to(bf)) and torch.equal(wpt, wp.t())
    assert wp[N:].abs().max().item() == 0 if wp.shape[0] > N else True
    lda = _ceil64(K) if pad_a else K
    a = torch.zeros(M, lda, device=cuda0, dtype=a_dtype)
    a[:, :K] = torch.randn(M, K, device=cuda0, generator=g).to(a_dtype)
    ncols = _ceil64(N) if pad_c else N
    c = torch.full((M, ncols), float("nan"), device=cuda0, dtype=c_dtype)
    a_copy = torch.full((M, _ceil64(K)), float("nan"), device=cuda0, dtype=bf) if copy else None
    Kk = lda if (a_dtype == bf and pad_a) else K
    _lib.call("dsw_linear", _lib.ptr(a), _lib.dtype_code(a_dtype), lda, M, Kk, _lib.ptr(wp), wp.shape[0], wp.shape[1], _lib.ptr(bias), N,
              _lib.ptr(c), _lib.dtype_code(c_dtype), ncols, ncols, 0, _lib.ptr(a_copy) if copy else None, a_copy.shape[1] if copy else 0,
              None, 0, _lib.stream(cuda0))
    ref = a[:, :K].to(bf).float() @ w.to(bf).float().t() + bias
    tol = 1e-2 if c_dtype == bf else 1e-5 * K
    assert ((c[:, :N].float() - ref).abs().max() / ref.abs().max()).item() < tol
    if ncols > N:
        assert c[:, N:].abs().max().item() == 0
    if copy:
        assert torch.equal(a_copy[:, :K], a[:, :K].to(bf))
        if a_copy.shape[1] > K:
            assert a_copy[:, K:].abs().max().item() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,C,padded_in,pad_out", [(515, 432, True, True), (1000, 192, False, True), (77, 100, True, False)])
def test_gelu_rows(cuda0, dtype, M, C, padded_in, pad_out):
    """nn.GELU() of msft's MLP (msft_arch.py:25, the erf form) on dense rows and on views of zero-padded rows, forward
    and backward, against torch's own gelu of the same inputs evaluated in fp32 (1e-6 for fp32; one bf16 rounding for
    bf16); the padding columns of the result and of the gradient are zero."""
    from visiontransformer_amd import functional as K
    gen = torch.Generator().manual_seed(M + C)
    ld = _ceil64(C) if padded_in else C
    buf = torch.zeros(M, ld)
    buf[:, :C] = torch.randn(M, C, generator=gen) * 2.0
    base = buf.to(dtype).to(cuda0)
    x = base[:, :C].detach().requires_grad_()
    dy = torch.randn(M, C, generator=gen).to(dtype).to(cuda0)
    y = K.gelu(x, pad_out=pad_out)
    assert y.shape == (M, C) and y.dtype == dtype
    if pad_out and _ceil64(C) > C:
        assert y.stride(0) == _ceil64(C)
        whole = torch.as_strided(y, (M, _ceil64(C)), (_ceil64(C), 1), y.storage_offset())
        assert whole[:, C:].abs().max().item() == 0
    gx, = torch.autograd.grad(y, x, dy)
    xr = x.detach().float().requires_grad_()
    yr = torch.nn.functional.gelu(xr)
    gr, = torch.autograd.grad(yr, xr, dy.float())
    tol = 1e-6 if dtype == torch.float32 else 2 ** -8
    assert ((y.float() - yr).abs() <= tol * (1 + yr.abs())).all()
    assert ((gx.float() - gr).abs() <= (1e-5 if dtype == torch.float32 else 2 ** -7) * (1 + gr.abs())).all()


@pytest.mark.gpu
@pytest.mark.parametrize("T,C,N", [(1000, 180, 540), (515, 96, 288), (4099, 192, 64), (300, 60, 180)])
def test_linear_with_the_rmsnorm_inside(cuda0, T, C, N):
    """lin(norm(x)) for nn.RMSNorm + nn.Linear (DSwinT_arch.py:355 with :334) in one kernel, inference under bf16
    autocast: against the fp32 modules within the bf16 bar, and against the two-kernel path (norm to bf16, then the
    projection) within a few bf16 roundings; the same for the gated first projection of the MLP (:357 with :69)."""
    from visiontransformer_amd import functional as K
    torch.manual_seed(T + C)
    norm = torch.nn.RMSNorm(C).to(cuda0)
    lin = torch.nn.Linear(C, N).to(cuda0)
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.3 * torch.randn(C, device=cuda0))
    x = torch.randn(3, T, C, device=cuda0) * torch.rand(3, T, 1, device=cuda0) * 4     # rows of very different norms
    ref = lin(norm(x))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = K.linear_prenorm(x, norm, lin, pad_out=True)
        y2 = K.linear(K.rms_norm(x, norm, for_projection=True), lin, pad_out=True)
        assert K.linear_prenorm(x.double(), norm, lin) is None
    assert y is not None and y.dtype == torch.bfloat16 and y.shape == ref.shape
    assert rel_err(y, ref) < 1e-2 and rel_err(y, y2) < 6e-3
    if N % 64:
        whole = torch.as_strided(y, (3 * T, _ceil64(N)), (_ceil64(N), 1), y.storage_offset())
        assert whole[:, N:].abs().max().item() == 0
    H = 2 * C                                   # the MLP's hidden width (ffn_scale 2): 360, 192, 384, 120
    if (H + 31) // 32 * 32 == _ceil64(H):
        gate, cont = torch.nn.Linear(C, H).to(cuda0), torch.nn.Linear(C, H).to(cuda0)
        iw, ib = K.interleave_gate_content(gate.weight, cont.weight, gate.bias, cont.bias)
        xn = norm(x)
        ref_g = torch.nn.functional.silu(gate(xn)) * cont(xn)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            g = K.linear_gated_silu(x, iw, ib, H, prenorm=norm)
        assert g is not None and g.shape == ref_g.shape and rel_err(g, ref_g) < 1e-2


@pytest.mark.gpu
def test_weight_prep_batch_and_the_kept_operands(cuda0):
    """dsw_weight_prep_batch against dsw_weight_prep, weight by weight (more weights than one launch holds, both
    dtypes), and the bookkeeping above it: a kept operand pair is reused while its parameter is unchanged, and the
    first projection after an in-place update (an optimizer step) refreshes the pairs of every modified parameter
    - 1x1 convolution weights taken as matrices included - in one call, as new tensors with the new values."""
    import ctypes
    from visiontransformer_amd import _lib, functional as K
    gen = torch.Generator().manual_seed(5)
    shapes = [(180, 180), (540, 180), (96, 288), (30, 44), (216, 648), (64, 64), (7, 500)] * 8      # 56 > 48 per launch
    ws = [torch.randn(n, k, generator=gen).to(torch.bfloat16 if i % 3 == 0 else torch.float32).to(cuda0) for i, (n, k) in enumerate(shapes)]
    singles, batch = [], []
    for w in ws:
        n, k = w.shape
        a = torch.full((_ceil64(n), _ceil64(k)), float("nan"), device=cuda0, dtype=torch.bfloat16)
        b = torch.full((_ceil64(k), _ceil64(n)), float("nan"), device=cuda0, dtype=torch.bfloat16)
        _lib.call("dsw_weight_prep", _lib.ptr(w), _lib.dtype_code(w.dtype), n, k, _lib.ptr(a), _lib.ptr(b), _lib.stream(cuda0))
        singles.append((a, b))
        batch.append((torch.full_like(a, float("nan")), torch.full_like(b, float("nan"))))
    cnt = len(ws)
    vp, ip = ctypes.c_void_p * cnt, ctypes.c_int * cnt
    tabs = (vp(*[w.data_ptr() for w in ws]), ip(*[_lib.dtype_code(w.dtype) for w in ws]), ip(*[w.shape[0] for w in ws]),
            ip(*[w.shape[1] for w in ws]), vp(*[a.data_ptr() for a, _ in batch]), vp(*[b.data_ptr() for _, b in batch]))
    _lib.call("dsw_weight_prep_batch", cnt, *[ctypes.cast(t, ctypes.c_void_p) for t in tabs], _lib.stream(cuda0))
    for (a, b), (a2, b2), w in zip(singles, batch, ws):
        assert torch.equal(a, a2) and torch.equal(b, b2)
        assert torch.equal(a[:w.shape[0], :w.shape[1]], w.to(torch.bfloat16)) and torch.equal(a.t(), b)

    lins = [torch.nn.Linear(180, 540).to(cuda0), torch.nn.Linear(96, 96, bias=False).to(cuda0)]
    conv = torch.nn.Conv2d(192, 96, 1).to(cuda0)
    mats = [lins[0].weight, lins[1].weight, conv.weight.view(96, 192)]
    first = [K._weight_operands(m) for m in mats]
    again = [K._weight_operands(conv.weight.view(96, 192) if i == 2 else m) for i, m in enumerate(mats)]
    for (a, b), (a2, b2) in zip(first, again):
        assert a2 is a and b2 is b                                 # reused, the fresh view of the convolution's weight too
    with torch.no_grad():
        for m in (lins[0].weight, conv.weight):
            m.mul_(0.5)                                            # what an optimizer step does to the version counters
    calls = []
    real_call = K.call

    def spy(name, *args):
        calls.append(name)
        return real_call(name, *args)

    K.call = spy
    try:
        after = [K._weight_operands(m) for m in (lins[0].weight, lins[1].weight, conv.weight.view(96, 192))]
    finally:
        K.call = real_call
    assert calls == ["dsw_weight_prep_batch"]                      # one call for both modified parameters
    assert after[1][0] is first[1][0]                              # the untouched one is still the kept pair
    for i, w in ((0, lins[0].weight), (2, conv.weight.view(96, 192))):
        assert after[i][0] is not first[i][0] and after[i][1] is not first[i][1]
        assert torch.equal(after[i][0][:w.shape[0], :w.shape[1]], w.detach().to(torch.bfloat16))
        assert torch.equal(after[i][0].t(), after[i][1])
        assert torch.equal(first[i][0][:w.shape[0], :w.shape[1]].float(), (w.detach() * 2).to(torch.bfloat16).float())   # the old pair is untouched


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N", [(300, 180, 540), (1000, 96, 100), (4096 + 24, 192, 64), (128 * 300 + 5, 180, 180),
                                   (2000, 360, 180), (3000, 216, 648), (700, 432, 216)])
def test_linear_wgrad_kernel(cuda0, M, K, N):
    """dsw_linear_wgrad (weight and bias gradient of a projection) against a float64 product of
    the same bf16 operands, and bitwise reproducible."""
    from visiontransformer_amd import _lib
    bf = torch.bfloat16
    g = torch.Generator(device=cuda0).manual_seed(M + K + N)
    Kp, Np = _ceil64(K), _ceil64(N)
    x = torch.zeros(M, Kp, device=cuda0, dtype=bf)
    x[:, :K] = torch.randn(M, K, device=cuda0, generator=g).to(bf)
    dy = torch.zeros(M, Np, device=cuda0, dtype=bf)
    dy[:, :N] = torch.randn(M, N, device=cuda0, generator=g).to(bf)
    outs = []
    for _ in range(2):
        dw = torch.full((N, K), float("nan"), device=cuda0)
        db = torch.full((N,), float("nan"), device=cuda0) if K < Kp else None
        ws = torch.empty(_lib.load().dsw_linear_wgrad_workspace(M, N, K), device=cuda0)
        _lib.call("dsw_linear_wgrad", _lib.ptr(dy), Np, _lib.ptr(x), Kp, M, N, K, _lib.ptr(dw), _lib.ptr(db) if db is not None else None,
                  _lib.ptr(ws), None, 0, _lib.stream(cuda0))
        outs.append((dw, db))
    dw, db = outs[0]
    ref = dy[:, :N].double().t() @ x[:, :K].double()
    assert rel_err(dw, ref.float()) < 1e-5
    assert torch.equal(dw, outs[1][0])
    if db is not None:
        assert rel_err(db, dy[:, :N].double().sum(0).float()) < 1e-5
        assert torch.equal(db, outs[1][1])
    else:
        with pytest.raises(RuntimeError, match="padding column"):
            _lib.call("dsw_linear_wgrad", _lib.ptr(dy), Np, _lib.ptr(x), Kp, M, N, K, _lib.ptr(dw), _lib.ptr(dw), _lib.ptr(ws),
                      None, 0, _lib.stream(cuda0))


@pytest.mark.gpu
def test_linear_function_against_torch(cuda0):
    """The projection Function under bf16 autocast (own MFMA kernels) against F.linear under the
    same autocast: values and all three gradients within 2e-2, for an fp32 input (cast fused),
    a padded bf16 input, and an input whose width leaves no padding column (bias gradient by column sums)."""
    import torch.nn.functional as F
    from visiontransformer_amd import functional as K
    torch.manual_seed(5)
    for Kd, N, T in ((180, 540, 128 * 9 + 3), (192, 64, 2048 * 8), (96, 100, 777)):
        lin = torch.nn.Linear(Kd, N).to(cuda0)
        x = torch.randn(T, Kd, device=cuda0, requires_grad=True)
        dy = torch.randn(T, N, device=cuda0)
        res = []
        for own in (False, True):
            lin.zero_grad(set_to_none=True)
            x.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = K.linear(x, lin, pad_out=True) if own else F.linear(x, lin.weight, lin.bias)
            (y.float() * dy).sum().backward()
            res.append((y.float(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()))
        for got, ref in zip(res[1], res[0]):
            assert rel_err(got, ref) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("M,C", [(1, 4), (77, 180), (4099, 192), (1000, 256)])
@pytest.mark.parametrize("mode", ["fp32", "bf16_padded"])
def test_rms_norm(cuda0, M, C, mode):
    """K.rms_norm against nn.RMSNorm: fp32 output and gradients to 1e-5; the bf16 padded form
    (under autocast, for a projection) equal to the fp32 result rounded to bf16, padding columns zero."""
    from visiontransformer_amd import functional as K
    torch.manual_seed(M + C)
    norm = torch.nn.RMSNorm(C).to(cuda0)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
    x = (torch.randn(M, C, device=cuda0) * 3).requires_grad_()
    dy = torch.randn(M, C, device=cuda0)
    ref = norm(x)
    gx, gw = torch.autograd.grad(ref, (x, norm.weight), dy)
    if mode == "fp32":
        got = K.rms_norm(x, norm)
        assert got.dtype == torch.float32 and rel_err(got, ref) < 1e-5
        hx, hw = torch.autograd.grad(got, (x, norm.weight), dy)
        assert rel_err(hx, gx) < 1e-5 and rel_err(hw, gw) < 1e-5
    else:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got = K.rms_norm(x, norm, for_projection=True)
        assert got.dtype == torch.bfloat16 and got.shape == (M, C)
        assert rel_err(got.float(), ref.to(torch.bfloat16).float()) < 1e-2
        ldo = (C + 63) // 64 * 64
        if ldo != C:
            buf = got._base
            assert buf.shape == (M, ldo) and buf[:, C:].abs().max().item() == 0
        hx, hw = torch.autograd.grad(got, (x, norm.weight), dy.to(torch.bfloat16))
        assert rel_err(hx, gx) < 2e-2 and rel_err(hw, gw) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,H,padded", [(5, 4, False), (333, 360, True), (1000, 64, False)])
def test_gated_silu(cuda0, dtype, M, H, padded):
    """silu(gate) * content on [gate | content] rows (dense, or a view of rows padded to 64) against torch,
    forward and backward; padding columns of the results are zero."""
    import torch.nn.functional as F
    from visiontransformer_amd import functional as K
    torch.manual_seed(M + H)
    ld = (2 * H + 63) // 64 * 64 if padded else 2 * H
    buf = torch.zeros(M, ld, device=cuda0, dtype=dtype)
    buf[:, :2 * H] = torch.randn(M, 2 * H, device=cuda0).to(dtype)
    h = buf[:, :2 * H].detach().requires_grad_()
    dy = torch.randn(M, H, device=cuda0).to(dtype)
    hr = h.detach().float().requires_grad_()
    ref = F.silu(hr[:, :H]) * hr[:, H:]
    gr, = torch.autograd.grad(ref, hr, dy.float())
    got = K.gated_silu(h, pad_out=True)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert got.shape == (M, H) and got.dtype == dtype and rel_err(got.float(), ref) < tol
    if got._base is not None and got._base.shape[1] > H:
        assert got._base[:, H:].abs().max().item() == 0
    gg, = torch.autograd.grad(got, h, dy)
    assert rel_err(gg.float(), gr) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("bdtype,padded", [(torch.float32, False), (torch.bfloat16, False), (torch.bfloat16, True)])
def test_rezero_residual(cuda0, bdtype, padded):
    """x + r * branch in one kernel against torch.addcmul, with the gradients of all three inputs."""
    from visiontransformer_amd import functional as K
    torch.manual_seed(7)
    B, H, W, C = 2, 9, 11, 180
    x = torch.randn(B, H, W, C, device=cuda0, requires_grad=True)
    r = torch.nn.Parameter(torch.tensor([0.37], device=cuda0))
    ld = 192 if padded else C
    buf = torch.zeros(B * H * W, ld, device=cuda0, dtype=bdtype)
    buf[:, :C] = torch.randn(B * H * W, C, device=cuda0).to(bdtype)
    br = buf[:, :C].view(B, H, W, C).detach().requires_grad_()
    dy = torch.randn(B, H, W, C, device=cuda0)
    ref = torch.addcmul(x, r, br)
    gx, gr, gb = torch.autograd.grad(ref, (x, r, br), dy)
    got = K.rezero_residual(x, r, br)
    hx, hr, hb = torch.autograd.grad(got, (x, r, br), dy)
    tol = 1e-5 if bdtype == torch.float32 else 1e-2
    assert rel_err(got, ref) < 1e-6 and rel_err(hx, gx) < 1e-6
    assert rel_err(hr, gr) < 1e-4 and rel_err(hb.float(), gb.float()) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("bdtype,padded", [(torch.float32, False), (torch.bfloat16, True)])
@pytest.mark.parametrize("p", [0.1, 0.6, 1.0])
def test_rezero_residual_with_dropout(cuda0, bdtype, padded, p):
    """x + r * dropout(branch) with the dropout applied inside the residual kernel: the keep decisions are the
    oracle's function of (seed, position) - no mask tensor exists - or an explicit mask; values and the three
    gradients match the torch expression on that mask."""
    from visiontransformer_amd import functional as K
    torch.manual_seed(11)
    B, H, W, C = 2, 9, 11, 180
    x = torch.randn(B, H, W, C, device=cuda0, requires_grad=True)
    r = torch.nn.Parameter(torch.tensor([0.37], device=cuda0))
    ld = 192 if padded else C
    buf = torch.zeros(B * H * W, ld, device=cuda0, dtype=bdtype)
    buf[:, :C] = torch.randn(B * H * W, C, device=cuda0).to(bdtype)
    br = buf[:, :C].view(B, H, W, C).detach().requires_grad_()
    dy = torch.randn(B, H, W, C, device=cuda0)
    drop = torch.nn.Dropout(p).train()
    keep = 1.0 / (1.0 - p) if p < 1.0 else 0.0
    tol = 1e-5 if bdtype == torch.float32 else 1e-2
    seed = 0x1234_5678_9ABC_DEF1
    if p < 1.0:
        masks = [(D.elementwise_keep_mask(B * H * W * C, p, seed).view(B, H, W, C).to(cuda0).to(torch.uint8), dict(dropout_seed=seed))]
    else:
        masks = [(torch.zeros(B, H, W, C, dtype=torch.uint8, device=cuda0), {})]
    explicit = torch.empty(B, H, W, C, dtype=torch.uint8, device=cuda0).bernoulli_(1.0 - p)
    masks.append((explicit, dict(dropout_mask=explicit)))
    for mask, kw in masks:
        assert p == 1.0 or abs(float(mask.float().mean()) - (1 - p)) < 0.02
        ref = x + r * (br.float() * mask * keep)
        gx, gr, gb = torch.autograd.grad(ref, (x, r, br), dy)
        got = K.rezero_residual(x, r, br, dropout=drop, **kw)
        hx, hr, hb = torch.autograd.grad(got, (x, r, br), dy)
        assert rel_err(got, ref) < 1e-6 and rel_err(hx, gx) < 1e-6
        assert rel_err(hr, gr) < 1e-4 and rel_err(hb.float(), gb.float()) < tol
        assert torch.equal(hb == 0, (mask == 0) | (dy == 0) if p < 1.0 else torch.ones_like(mask, dtype=torch.bool))
    if p < 1.0:     # the seed comes from torch's CPU generator when none is given: reproducible under manual_seed
        torch.manual_seed(5)
        a = K.rezero_residual(x, r, br, dropout=drop)
        torch.manual_seed(5)
        b = K.rezero_residual(x, r, br, dropout=drop)
        c = K.rezero_residual(x, r, br, dropout=drop)
        assert torch.equal(a, b) and not torch.equal(a, c)
    # evaluation mode: no mask, the plain residual
    assert torch.equal(K.rezero_residual(x, r, br, dropout=drop.eval()), K.rezero_residual(x, r, br))


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Ci,Co,xdtype", [(1, 5, 7, 12, 8, torch.float32), (2, 33, 20, 180, 180, torch.float32),
                                                (1, 64, 64, 192, 64, torch.bfloat16), (3, 16, 16, 64, 256, torch.float32),
                                                (2, 6, 64, 64, 64, torch.bfloat16), (1, 3, 128, 64, 8, torch.float32),
                                                (3, 5, 64, 64, 256, torch.bfloat16), (2, 1, 192, 64, 4, torch.float32),
                                                (2, 4, 64, 16, 4, torch.bfloat16), (1, 7, 128, 36, 72, torch.float32)])
def test_conv3x3_nhwc(cuda0, B, H, W, Ci, Co, xdtype):
    """The NHWC 3x3 convolution (inference, bf16 autocast) against F.conv2d on the same bf16-rounded
    operands: borders, image boundaries inside the batch, channel counts off the 64 grid; 64 input channels on
    widths that are multiples of 64 (the weight gradient's three-row-strip kernel), one to six image rows."""
    import torch.nn.functional as F
    from visiontransformer_amd import functional as K
    torch.manual_seed(B + H + Ci)
    conv = torch.nn.Conv2d(Ci, Co, 3, 1, 1).to(cuda0)
    x = torch.randn(B, H, W, Ci, device=cuda0).to(xdtype)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        got = K.conv3x3_nhwc(x, conv)
    assert got is not None and got.shape == (B, H, W, Co) and got.dtype == torch.bfloat16
    bf = torch.bfloat16
    ref = F.conv2d(x.to(bf).float().permute(0, 3, 1, 2), conv.weight.to(bf).float(), conv.bias.float(), padding=1).permute(0, 2, 3, 1)
    assert ((got.float() - ref).abs().max() / ref.abs().max()).item() < 1e-2
    # training: input, weight and bias gradients against autograd through F.conv2d on the same rounded operands
    xg = x.detach().clone().requires_grad_()
    dy = torch.randn(B, H, W, Co, device=cuda0)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = K.conv3x3_nhwc(xg, conv)
    conv.zero_grad(set_to_none=True)
    y.backward(dy.to(bf))
    xr = x.to(bf).float().requires_grad_()
    wr = conv.weight.detach().to(bf).float().requires_grad_()
    br = conv.bias.detach().float().requires_grad_()
    F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=1).permute(0, 2, 3, 1).backward(dy.to(bf).float())
    assert rel_err(xg.grad.float(), xr.grad) < 2e-2
    assert rel_err(conv.weight.grad, wr.grad) < 2e-2 and rel_err(conv.bias.grad, br.grad) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("M,C", [(3, 8), (77, 96), (4099, 216), (1000, 256)])
@pytest.mark.parametrize("mode", ["fp32", "bf16_padded"])
def test_layer_norm(cuda0, M, C, mode):
    """K.layer_norm against nn.LayerNorm, as test_rms_norm."""
    from visiontransformer_amd import functional as K
    torch.manual_seed(M + C)
    norm = torch.nn.LayerNorm(C).to(cuda0)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
    x = (torch.randn(M, C, device=cuda0) * 3 + 1).requires_grad_()
    dy = torch.randn(M, C, device=cuda0)
    ref = norm(x)
    gx, gw, gb = torch.autograd.grad(ref, (x, norm.weight, norm.bias), dy)
    if mode == "fp32":
        got = K.layer_norm(x, norm)
        assert got.dtype == torch.float32 and rel_err(got, ref) < 1e-5
        hx, hw, hb = torch.autograd.grad(got, (x, norm.weight, norm.bias), dy)
        assert rel_err(hx, gx) < 1e-4 and rel_err(hw, gw) < 1e-5 and rel_err(hb, gb) < 1e-5
    else:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got = K.layer_norm(x, norm, for_projection=True)
        assert got.dtype == torch.bfloat16 and got.shape == (M, C)
        assert rel_err(got.float(), ref.to(torch.bfloat16).float()) < 1e-2
        ldo = (C + 63) // 64 * 64
        if ldo != C:
            assert got._base.shape == (M, ldo) and got._base[:, C:].abs().max().item() == 0
        hx, hw, hb = torch.autograd.grad(got, (x, norm.weight, norm.bias), dy.to(torch.bfloat16))
        assert rel_err(hx, gx) < 2e-2 and rel_err(hw, gw) < 2e-2 and rel_err(hb, gb) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("T,Kd,H,padded_in", [(1000, 180, 360, True), (1000, 180, 360, False), (257, 64, 128, False), (4096 + 3, 192, 40 + 64 * 5, True)])
def test_linear_gated_silu(cuda0, T, Kd, H, padded_in):
    """fc1 with the gated SiLU in its epilogue (DSwinT_arch.py:69) against the fp32 formula on the bf16-rounded
    operands, and bit for bit against the projection kernel followed by the stand-alone gate kernel"""
    K = _K()
    gen = torch.Generator().manual_seed(T + H)
    x = torch.randn(T, Kd, generator=gen)
    wg, wc = torch.randn(H, Kd, generator=gen) * Kd ** -0.5, torch.randn(H, Kd, generator=gen) * Kd ** -0.5
    bg, bc = torch.randn(H, generator=gen) * 0.3, torch.randn(H, generator=gen) * 0.3
    xb, wgb, wcb = x.bfloat16().float(), wg.bfloat16().float(), wc.bfloat16().float()
    g = (xb @ wgb.t() + bg).bfloat16().float()
    c = (xb @ wcb.t() + bc).bfloat16().float()
    ref = torch.nn.functional.silu(g) * c
    dev = cuda0
    if padded_in:
        xin = K._mark_padded(K.cast_rows(x.to(dev), torch.bfloat16, K._ceil_pad(Kd)))[:, :Kd]
    else:
        xin = x.to(dev)                                          # the fp32 stream, rounded on the way in
    iw, ib = K.interleave_gate_content(wg.to(dev), wc.to(dev), bg.to(dev), bc.to(dev))

    class Stacked:
        weight, bias = torch.cat([wg, wc]).to(dev), torch.cat([bg, bc]).to(dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out = K.linear_gated_silu(xin, iw, ib, H)
        assert out is not None and out.shape == (T, H) and out.dtype == torch.bfloat16
        sep = K.gated_silu(K.linear(xin, Stacked, pad_out=True), pad_out=True)
    assert rel_err(out, ref) < 1e-2
    assert torch.equal(out, sep)
    if H % 64:
        assert float(out._base[:, H:].abs().max()) == 0.0
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert K.linear_gated_silu(xin, iw, ib, H) is None       # with autograd on the fused call declines


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,w,C,heads", [(torch.bfloat16, 16, 180, 6), (torch.bfloat16, 8, 24, 2), (torch.float32, 16, 48, 3)])
def test_switched_off_windows_cost_nothing_and_give_zeros(cuda0, dtype, w, C, heads):
    """n_live: a counter on the device; window slots at and beyond it are switched off (DS-MSA keeps the number of
    windows that passed the gate on the device, DSwinT_arch.py:280-292).  Attention and projections: the live part
    equals a run on the live windows alone bit for bit, the rest is zero, forward and backward."""
    K = _K()
    from visiontransformer_amd._lib import QKV_CHUNKED
    gen = torch.Generator().manual_seed(w + C)
    n, live, L = 5, 2, w * w
    qkv = torch.randn(n, L, 3 * C, generator=gen).to(dtype).to(cuda0)
    table = (torch.randn((2 * w - 1) ** 2, heads, generator=gen) * 0.5).to(cuda0)
    r = torch.randn(n, L, C, generator=gen).to(cuda0)
    n_live = torch.tensor([live], dtype=torch.int32, device=cuda0)

    def run(q, cnt):
        q = q.clone().requires_grad_()
        t = table.clone().requires_grad_()
        o = K.window_attention(q, t, w, heads, QKV_CHUNKED, n_live=cnt)
        (o.float() * r[:q.shape[0]]).sum().backward()
        return o.detach(), q.grad, t.grad
    o_all, dq_all, dt_all = run(qkv, n_live)
    o_ref, dq_ref, dt_ref = run(qkv[:live], None)
    assert torch.equal(o_all[:live], o_ref) and torch.equal(dq_all[:live], dq_ref) and torch.equal(dt_all, dt_ref)
    assert float(o_all[live:].abs().max()) == 0.0 and float(dq_all[live:].abs().max()) == 0.0
    zero = torch.zeros(1, dtype=torch.int32, device=cuda0)                       # nothing live: all zeros
    o0, dq0, dt0 = run(qkv, zero)
    assert float(o0.abs().max()) == 0.0 and float(dq0.abs().max()) == 0.0 and float(dt0.abs().max()) == 0.0
    if dtype != torch.bfloat16:
        return
    # projections under bf16 autocast: rows of switched-off units are not computed, forward, input and weight gradient
    lin = torch.nn.Linear(C, 3 * C).to(cuda0)
    x = torch.randn(n, L, C, generator=gen).to(cuda0)
    g = torch.randn(n, L, 3 * C, generator=gen).to(cuda0)
    g[live:] = 0.0                                                               # what reaches switched-off rows in DS-MSA

    def run_lin(xx, gg, cnt):
        lin.zero_grad(set_to_none=True)
        xx = xx.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = K.linear(xx, lin, pad_out=True, live=None if cnt is None else (cnt, L))
        (y.float() * gg).sum().backward()
        return y.detach(), xx.grad, lin.weight.grad.clone(), lin.bias.grad.clone()
    y_all, dx_all, dw_all, db_all = run_lin(x, g, n_live)
    y_ref, dx_ref, dw_ref, db_ref = run_lin(x[:live], g[:live], None)
    assert torch.equal(y_all[:live], y_ref) and torch.equal(dx_all[:live], dx_ref)
    assert float(y_all[live:].abs().max()) == 0.0 and float(dx_all[live:].abs().max()) == 0.0
    assert rel_err(dw_all, dw_ref) < 1e-6 and rel_err(db_all, db_ref) < 1e-6    # other split of the row range: other summation order
    # and the fused projection + attention kernel
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        xp = K._mark_padded(K.cast_rows(x.reshape(-1, C), torch.bfloat16, K._ceil_pad(C)))[:, :C].view(n, L, C)
        z = K.projected_window_attention(xp, lin, table, w, heads, QKV_CHUNKED, n_live=n_live)
        z_ref = K.projected_window_attention(xp[:live], lin, table, w, heads, QKV_CHUNKED)
    if z is not None:
        assert torch.equal(z[:live], z_ref) and float(z[live:].abs().max()) == 0.0
