"""The trainer's optimizer (toolkit/init.py:133, stepped at toolkit/model_helper2.py:125): the numpy restatement
against torch.optim.Adam itself, the drop-in class on CPU parameters, and the one-pass kernel against both.

Tolerances.  The update is fp32 elementwise arithmetic.  The kernel against the restatement, and the drop-in
class against torch.optim.Adam on the GPU (the very computation the reference's trainer runs): bit-exact.  The
restatement against torch.optim.Adam on the CPU, whose kernels round a few steps differently (contraction, true
division): |difference| <= 2e-6 * (|value| + floor) per element after six steps, floor = the learning rate for a
parameter, the largest gradient for the first moment and its square for the second."""
import numpy as np
import pytest
import torch

from oracle import optim_oracle
from visiontransformer_amd import optim

HYPER = dict(lr=2e-4, beta1=0.9, beta2=0.99, eps=1e-8)      # the reference's betas; lr of its train config scale


def _tensors(sizes, seed, offsets=True):
    gen = torch.Generator().manual_seed(seed)
    out = []
    for i, n in enumerate(sizes):
        off = i % 3 if offsets else 0       # element offsets 0, 1, 2: bases that are not 16-byte aligned
        out.append((torch.randn(n + off, generator=gen) * (0.02 + 0.3 * (i % 5)))[off:])
    return out


def _close(got, ref, floor, what):
    """|got - ref| <= 2e-6 * (|ref| + floor) per element; floor = the size of the terms that were added to make the
    value (the learning rate for a parameter, the largest gradient for a moment)"""
    tol = 2e-6 * (ref.abs() + floor)
    bad = (got - ref).abs() > tol
    assert not bad.any(), "%s: %d of %d elements differ by more than 2e-6 * (|x| + %g); largest %g" % (
        what, int(bad.sum()), bad.numel(), floor, float((got - ref).abs().max()))


def _oracle_run(p0, grads, steps):
    p = [x.numpy().copy() for x in p0]
    m = [np.zeros_like(x) for x in p]
    v = [np.zeros_like(x) for x in p]
    for s in range(steps):
        for i in range(len(p)):
            p[i], m[i], v[i] = optim_oracle.adam_step(p[i], grads[s][i].numpy(), m[i], v[i], step=s + 1,
                                                      **HYPER)
    return p, m, v


def test_oracle_against_torch_adam_on_the_cpu():
    sizes = [1, 5, 64, 1031, 4097]
    p0 = _tensors(sizes, 1, offsets=False)
    grads = [_tensors(sizes, 10 + s, offsets=False) for s in range(6)]
    params = [torch.nn.Parameter(x.clone()) for x in p0]
    opt = torch.optim.Adam(params, lr=HYPER["lr"], betas=(HYPER["beta1"], HYPER["beta2"]), eps=HYPER["eps"])
    for s in range(6):
        for q, g in zip(params, grads[s]):
            q.grad = g.clone()
        opt.step()
    p, m, v = _oracle_run(p0, grads, 6)
    for i, q in enumerate(params):
        _close(torch.from_numpy(p[i]), q.detach(), HYPER["lr"], "param %d" % i)
        st, gmax = opt.state[q], max(float(g[i].abs().max()) for g in grads)
        _close(torch.from_numpy(m[i]), st["exp_avg"], gmax, "exp_avg %d" % i)
        _close(torch.from_numpy(v[i]), st["exp_avg_sq"], gmax * gmax, "exp_avg_sq %d" % i)


def test_drop_in_class_on_cpu_parameters_and_checkpoints():
    """CPU parameters take torch's own update: the same bits as torch.optim.Adam, and state_dicts that load
    into each other in the middle of a run"""
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.GELU(), torch.nn.Linear(33, 3))
    b = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.GELU(), torch.nn.Linear(33, 3))
    b.load_state_dict(a.state_dict())
    oa = optim.Adam(a.parameters(), lr=1e-3, betas=(0.9, 0.99))
    ob = torch.optim.Adam(b.parameters(), lr=1e-3, betas=(0.9, 0.99))
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(oa, T_max=10, eta_min=5e-6)      # toolkit/init.py:136
    sched_b = torch.optim.lr_scheduler.CosineAnnealingLR(ob, T_max=10, eta_min=5e-6)
    for s in range(4):
        x = torch.randn(5, 7)
        for net, o in ((a, oa), (b, ob)):
            o.zero_grad(set_to_none=True)
            net(x).square().mean().backward()
            torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)                      # toolkit/model_helper2.py:124
            o.step()
        sched.step()
        sched_b.step()
        if s == 1:      # swap the optimizer states through their state_dicts
            sa, sb = oa.state_dict(), ob.state_dict()
            oa.load_state_dict(sb)
            ob.load_state_dict(sa)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
    assert set(oa.state_dict()["state"][0]) == set(ob.state_dict()["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(oa.state_dict()["state"][0]["step"]) == 4.0


def test_unsupported_options_are_refused():
    w = [torch.nn.Parameter(torch.zeros(3))]
    for kw in (dict(weight_decay=0.1), dict(amsgrad=True), dict(maximize=True), dict(fused=True), dict(capturable=True)):
        with pytest.raises(ValueError):
            optim.Adam(w, **kw)
    o = optim.Adam(w)
    o.param_groups[0]["weight_decay"] = 0.1
    w[0].grad = torch.ones(3)
    with pytest.raises(RuntimeError):
        o.step()
    # beta1 <= 0.5: torch's lerp takes its other form there, which the kernel does not follow
    o = optim.Adam(w, betas=(0.5, 0.99))
    with pytest.raises(RuntimeError):
        o.step()


SIZES = [0, 1, 3, 4, 5, 2047, 2048, 2049, 70001] + [17 + 13 * i for i in range(140)]     # more than two launches' worth


@pytest.mark.gpu
def test_kernel_against_the_oracle_bit_for_bit(cuda0):
    p0 = _tensors(SIZES, 2)
    grads = [_tensors(SIZES, 20 + s) for s in range(3)]
    want_p, want_m, want_v = _oracle_run(p0, grads, 3)

    def on_gpu(ts):
        return [torch.empty(t.numel() + i % 3, device=cuda0)[i % 3:].copy_(t) for i, t in enumerate(ts)]

    p, m, v = on_gpu(p0), on_gpu([torch.zeros_like(t) for t in p0]), on_gpu([torch.zeros_like(t) for t in p0])
    versions = [t._version for t in p]
    for s in range(3):
        optim.adam_update_(p, on_gpu(grads[s]), m, v, step=s + 1, **HYPER)
    assert all(t._version > ver for t, ver in zip(p, versions) if t.numel())
    for i in range(len(SIZES)):
        for got, want, name in ((p[i], want_p[i], "p"), (m[i], want_m[i], "m"), (v[i], want_v[i], "v")):
            same = torch.equal(got.cpu(), torch.from_numpy(want))
            assert same, "%s of tensor %d (%d elements)" % (name, i, SIZES[i])


@pytest.mark.gpu
def test_drop_in_class_against_torch_adam_on_the_gpu(cuda0):
    """a small DSWT network trained side by side with torch.optim.Adam (multi-tensor, as the reference runs it) from
    the same gradients: parameters and both moments after five steps; one parameter without a gradient keeps its
    values and its step count"""
    from visiontransformer_amd import models
    torch.manual_seed(0)
    cfg = dict(models.DSWT_DEFAULT)
    cfg.update(n_blocks=1, n_sub_blocks=1, dim=48, n_heads=2, window_size=8, feature_dim=16)
    a = models.DSWTransformer(**cfg).to(cuda0)
    b = models.DSWTransformer(**cfg).to(cuda0)
    b.load_state_dict(a.state_dict())
    pa, pb = list(a.parameters()), list(b.parameters())
    oa = optim.Adam(pa, lr=2e-4, betas=(0.9, 0.99))
    ob = torch.optim.Adam(pb, lr=2e-4, betas=(0.9, 0.99), foreach=True)
    gen = torch.Generator(device=cuda0).manual_seed(5)
    frozen = 3
    for s in range(5):
        for i, (x, y) in enumerate(zip(pa, pb)):
            if i == frozen and s < 2:
                x.grad = y.grad = None
                continue
            g = torch.randn(x.shape, device=cuda0, generator=gen) * (1e-3 if i % 2 else 1.0)
            x.grad, y.grad = g, g.clone()
        oa.step()
        ob.step()
        for i, (x, y) in enumerate(zip(pa, pb)):
            same = torch.equal(x, y)
            assert same, "step %d, parameter %d %s: largest difference %g" % (s + 1, i, tuple(x.shape), float((x - y).abs().max()))
        if s == 3:      # swap the optimizer states through their state_dicts, as a checkpoint would
            sa, sb = oa.state_dict(), ob.state_dict()
            oa.load_state_dict(sb)
            ob.load_state_dict(sa)
    assert float(oa.state[pa[0]]["step"]) == float(ob.state[pb[0]]["step"]) == 5.0
    assert float(oa.state[pa[frozen]]["step"]) == float(ob.state[pb[frozen]]["step"]) == 3.0
    for i, (x, y) in enumerate(zip(pa, pb)):
        same = torch.equal(x, y)
        assert same, "parameter %d: largest difference %g" % (i, float((x - y).abs().max()))
        for k in ("exp_avg", "exp_avg_sq"):
            same = torch.equal(oa.state[x][k], ob.state[y][k])
            assert same, "%s of parameter %d" % (k, i)


@pytest.mark.gpu
def test_channels_last_parameters_and_gradients_in_other_layouts(cuda0):
    """the update is elementwise: a dense permuted layout goes through the kernel as it lies, and a gradient laid out
    differently from its parameter is brought to the parameter's layout first (values against the restatement)"""
    gen = torch.Generator().manual_seed(4)
    w0 = torch.randn(8, 5, 3, 3, generator=gen)
    b0 = torch.randn(40, generator=gen)
    w = torch.nn.Parameter(w0.to(cuda0).contiguous(memory_format=torch.channels_last))
    b = torch.nn.Parameter(b0.to(cuda0))
    opt = optim.Adam([w, b], lr=HYPER["lr"], betas=(HYPER["beta1"], HYPER["beta2"]), eps=HYPER["eps"])
    want = [(w0.numpy(), np.zeros(w0.shape, np.float32), np.zeros(w0.shape, np.float32)),
            (b0.numpy(), np.zeros(40, np.float32), np.zeros(40, np.float32))]
    for s in range(3):
        gw, gb = torch.randn(8, 5, 3, 3, generator=gen), torch.randn(80, generator=gen)[::2]
        w.grad = gw.to(cuda0) if s != 1 else gw.to(cuda0).contiguous(memory_format=torch.channels_last)
        b.grad = gb.to(cuda0)                                     # s == 0, 2: row-major gradient of a channels-last weight
        assert s == 1 or w.grad.stride() != w.stride()
        opt.step()
        want = [optim_oracle.adam_step(p, g.numpy(), m, v, step=s + 1, **HYPER) for (p, m, v), g in zip(want, (gw, gb))]
    assert opt._plans[0].fast and not opt._plans[0].slow          # both went through the kernel
    assert w.is_contiguous(memory_format=torch.channels_last)
    for q, (p, m, v) in zip((w, b), want):
        for got, ref, name in ((q.detach(), p, "p"), (opt.state[q]["exp_avg"], m, "m"), (opt.state[q]["exp_avg_sq"], v, "v")):
            same = torch.equal(got.cpu().contiguous(), torch.from_numpy(ref))
            assert same, name
