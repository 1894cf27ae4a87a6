"""The CPU oracle against the golden vectors the reference itself produced
(tests/golden/make_golden.py).  fp32, CPU: outputs must agree to 1e-5
relative, integer decisions exactly."""
import pytest
import torch

from conftest import golden_names, load_golden, params_of, rel_err
from oracle import dswin_oracle as D
from oracle import msft_oracle as M

TOL = 1e-5


def _check_grads(g, x, params, y):
    (y * g["r"]).sum().backward()
    assert rel_err(x.grad, g["dx"]) < TOL
    for k, p in params.items():
        if not p.requires_grad:
            continue
        ref = g["grad." + k]
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        assert rel_err(got, ref) < TOL, k


def _leafify(P):
    return {k: (v.clone().requires_grad_() if v.is_floating_point() else v) for k, v in P.items()}


@pytest.mark.parametrize("name", golden_names("dswt_wmsa"))
def test_dswt_wmsa(name):
    g, m = load_golden(name)
    P = _leafify(params_of(g))
    x = g["x"].clone().requires_grad_()
    y = D.wmsa_forward(P, x, m["w"], m["heads"])
    assert rel_err(y, g["y"]) < TOL
    _check_grads(g, x, P, y)


@pytest.mark.parametrize("name", golden_names("dswt_dsmsa"))
def test_dswt_dsmsa(name):
    g, m = load_golden(name)
    P = _leafify(params_of(g))
    x = g["x"].clone().requires_grad_()
    aux = {}
    y = D.dsmsa_forward(P, x, m["w"], m["heads"], training=m["training"], noise=g["noise"],
                        temperature=m["temperature"], num_win_selected=m["num_win_selected"], aux=aux)
    assert aux["cand_b"].numel() == m["n_candidates"]
    for k in ("cand_b", "cand_h", "cand_w", "b_idx", "h_idx", "w_idx", "h1", "w1"):
        if "aux." + k in g:
            assert torch.equal(aux[k], g["aux." + k]), k
    assert rel_err(y, g["y"]) < TOL
    _check_grads(g, x, P, y)
    # offsets never receive gradient in the reference (DSwinT_arch.py:232-233)
    gw = g["grad.sparse_discriminator.4.weight"]
    assert torch.count_nonzero(gw[2:]) == 0


def test_roi_align_general_equals_specialised():
    """the general torchvision-style restatement and the integer-window
    specialisation agree, including windows clamped against the border."""
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 24, 32, generator=gen)
    w = 8
    b = torch.tensor([0, 1, 1, 0])
    h1 = torch.tensor([0, 16, 7, 3])
    w1 = torch.tensor([24, 0, 11, 24])
    boxes = torch.stack([b, w1, h1, w1 + w, h1 + w], 1).float()
    a = D.roi_align(x, boxes, (w, w))
    s = D.gather_windows(x, b, h1, w1, w)
    assert torch.allclose(a, s, atol=1e-6)
    # known answers: interior = 2x2 box mean, last row/col = 2x1 mean, corner = the pixel
    n = 0  # window 0 at h1=0,w1=24 touches the right border (W=32)
    assert torch.allclose(s[n, :, 2, 3], 0.25 * (x[0, :, 2, 27] + x[0, :, 2, 28] + x[0, :, 3, 27] + x[0, :, 3, 28]), atol=1e-6)
    assert torch.allclose(s[n, :, 2, 7], 0.5 * (x[0, :, 2, 31] + x[0, :, 3, 31]), atol=1e-6)
    n = 1  # window 1 at h1=16 touches the bottom border (H=24)
    assert torch.allclose(s[n, :, 7, 2], 0.5 * (x[1, :, 23, 2] + x[1, :, 23, 3]), atol=1e-6)


def test_scatter_mean_residual_matches_dsmsa_block():
    """the NHWC formulation used as the kernel-level checker equals the
    NCHW block it restates, with overlapping windows."""
    gen = torch.Generator().manual_seed(3)
    B, H, W, C, w = 2, 24, 32, 6, 8
    x = torch.randn(B, H, W, C, generator=gen)
    b = torch.tensor([0, 0, 0, 1, 1])
    h1 = torch.tensor([0, 4, 4, 16, 12])
    w1 = torch.tensor([0, 4, 10, 24, 20])
    after = torch.randn(5, w * w, C, generator=gen)
    gd = torch.rand(5, generator=gen) + 0.5
    sdl = torch.rand(5, generator=gen) + 0.5
    y = D.scatter_mean_residual(x, after, gd, sdl, b, h1, w1, w)
    xn = x.permute(0, 3, 1, 2).contiguous()
    before = D.gather_windows(xn, b, h1, w1, w)
    aft = after.view(-1, w, w, C).permute(0, 3, 1, 2)
    ref = xn + D.integrate(xn.shape, (aft - before) * gd.view(-1, 1, 1, 1) * sdl.view(-1, 1, 1, 1), b, h1, w1, w)
    assert torch.allclose(y, ref.permute(0, 2, 3, 1), atol=1e-6)


@pytest.mark.parametrize("name", golden_names("msft_wmsa"))
def test_msft_wmsa(name):
    g, m = load_golden(name)
    P = _leafify(params_of(g))
    x = g["x"].clone().requires_grad_()
    y = M.wmsa_forward(P, x, m["H"], m["W"], m["w"], m["heads"])
    assert rel_err(y, g["y"]) < TOL
    _check_grads(g, x, P, y)


@pytest.mark.parametrize("name", ["msft_patch_merging", "msft_patch_unmerging"])
def test_msft_patch(name):
    g, m = load_golden(name)
    P = _leafify(params_of(g))
    x = g["x"].clone().requires_grad_()
    fn = M.patch_merging if m["kind"].endswith("PatchMerging") else M.patch_unmerging
    y = fn(P, x, m["scale"])
    assert rel_err(y, g["y"]) < TOL
    _check_grads(g, x, P, y)


def test_msft_vitblock():
    g, m = load_golden("msft_vitblock")
    P = _leafify(params_of(g))
    x = g["x"].clone().requires_grad_()
    y = M.vit_block(P, x, m["w"], m["heads"])
    assert rel_err(y, g["y"]) < TOL
    _check_grads(g, x, P, y)


def test_dsmsa_helper_methods_follow_the_oracle():
    """candidate_indexes / cords / integrate of the drop-in class (DSwinT_arch.py:201-253; plain torch ops, run here
    on the CPU) against the oracle's restatements, which the golden vectors pin"""
    from visiontransformer_amd.dswin import DeformableSparseWindowAttention
    torch.manual_seed(4)
    w, C, B, H, W = 8, 12, 2, 32, 48
    mod = DeformableSparseWindowAttention(w, C, 2).eval()
    mod.num_win_selected = 0.5
    sd = torch.randn(B, 4, H // w, W // w)
    x = torch.randn(B, C, H, W)
    b, h, ww, sdl = mod.candidate_indexes(x, sd[:, :1])
    rb, rh, rw, rs, _ = D.candidate_indexes(sd[:, :1], 0.5, 1.0, False)
    assert torch.equal(b, rb) and torch.equal(h, rh) and torch.equal(ww, rw) and torch.equal(sdl, rs)
    off = sd[:, 2:].sigmoid()
    got = mod.cords(b, h, ww, off, H, W)
    ref = D.cords(rb, rh, rw, off, w, H, W)
    assert all(torch.equal(a, r) for a, r in zip(got, ref))
    delta = torch.randn(b.numel(), C, w, w)
    assert torch.equal(mod.integrate(x, delta, b, got[0], got[1]), D.integrate(x.shape, delta, rb, ref[0], ref[1], w))
