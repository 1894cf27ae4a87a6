"""Parameter EMA (toolkit/param_ema.py): the oracle and the host class against vectors written by the
reference's own ParameterEMA (tests/golden/make_golden_ema.py), the fused kernel against the oracle."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import ema_oracle
from visiontransformer_amd.ema import ParameterEMA, ema_update_

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ema_reference.npz")


def _golden():
    z = np.load(GOLDEN, allow_pickle=False)
    n = int(z["n_params"])
    cur = [[torch.from_numpy(z["cur.%d.%d" % (s, i)]) for i in range(n)] for s in range(4)]
    ema = [[torch.from_numpy(z["ema.%d.%d" % (s, i)]) for i in range(n)] for s in range(4)]
    return cur, ema


def _toy():
    return torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.Linear(7, 1031), torch.nn.Linear(1031, 3, bias=False),
                               torch.nn.LayerNorm(9))


def test_oracle_matches_the_reference_bit_for_bit():
    cur, ema = _golden()
    state = dict(step=0, start_step=1, beta=0.995)
    e = [c + 1.0 for c in cur[0]]
    for s in range(4):
        e = ema_oracle.step_ema(state, e, cur[s])
        for got, ref in zip(e, ema[s]):
            assert torch.equal(got, ref)
    assert not torch.equal(ema[1][1], cur[1][1])   # the later steps average, they do not copy


def _drive(device):
    cur, ema = _golden()
    model, ema_model = _toy().to(device), _toy().to(device).eval().requires_grad_(False)
    helper = ParameterEMA(start_step=1, beta=0.995)
    for s in range(4):
        with torch.no_grad():
            for p, c in zip(model.parameters(), cur[s]):
                p.copy_(c)
        helper.step_ema(ema_model, model, new_ema=True)
        for p, ref in zip(ema_model.parameters(), ema[s]):
            assert torch.equal(p.detach().cpu(), ref), "step %d" % s
    assert helper.step == 4


def test_host_class_on_cpu_tensors():
    _drive(torch.device("cpu"))


@pytest.mark.gpu
def test_host_class_on_the_gpu_matches_the_reference_bit_for_bit(cuda0):
    _drive(cuda0)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_kernel_against_the_oracle(cuda0, dtype):
    """more tensors than one launch takes, empty and ragged sizes, bases that are not 16-byte aligned"""
    gen = torch.Generator().manual_seed(9)
    sizes = [0, 1, 3, 4, 5, 4095, 4096, 4097, 70001] + [17 + 13 * i for i in range(150)]
    ema_c, cur_c = [], []
    for i, n in enumerate(sizes):
        off = i % 3                      # element offsets 0, 1, 2 into a larger buffer
        ema_c.append(torch.randn(n + off, generator=gen).to(dtype)[off:])
        cur_c.append(torch.randn(n + off, generator=gen).to(dtype)[off:])
    ref = [ema_oracle.update_average(e, c, 0.995) for e, c in zip(ema_c, cur_c)]
    ema_g = [torch.empty(e.numel() + i % 3, dtype=dtype, device=cuda0)[i % 3:].copy_(e) for i, e in enumerate(ema_c)]
    cur_g = [torch.empty(c.numel() + i % 3, dtype=dtype, device=cuda0)[i % 3:].copy_(c) for i, c in enumerate(cur_c)]
    ema_update_(ema_g, cur_g, 0.995)
    for i, (g, r) in enumerate(zip(ema_g, ref)):
        assert torch.equal(g.cpu(), r), "tensor %d (%d elements)" % (i, r.numel())
    # a second step on the result (in place), as training does
    ref2 = [ema_oracle.update_average(r, c, 0.995) for r, c in zip(ref, cur_c)]
    ema_update_(ema_g, cur_g, 0.995)
    for g, r in zip(ema_g, ref2):
        assert torch.equal(g.cpu(), r)


def test_parameter_lists_follow_replaced_parameters():
    """the flat parameter lists are kept between steps; a Parameter object replaced in a module is seen"""
    model, ema_model = _toy(), _toy().eval().requires_grad_(False)
    helper = ParameterEMA(start_step=0, beta=0.5)
    helper.step_ema(ema_model, model)
    model[1].weight = torch.nn.Parameter(torch.full_like(model[1].weight, 3.0))
    with torch.no_grad():
        ema_model[1].weight.fill_(1.0)
    helper.step_ema(ema_model, model)
    assert torch.equal(ema_model[1].weight, torch.full_like(ema_model[1].weight, 2.0))


@pytest.mark.gpu
def test_ema_update_is_seen_by_the_prepared_weight_caches(cuda0):
    """Without autograd the projections, 3x3 convolutions and the stacked fc1 keep their prepared bf16 operands
    per parameter until the parameter's version counter moves.  The fused update writes through raw pointers:
    it has to move that counter, or a second evaluation of the EMA model (toolkit/model_helper2.py sample_model,
    use_ema_model=True) would run on the weights of the first."""
    from visiontransformer_amd import models
    torch.manual_seed(0)
    cfg = dict(models.DSWT_DEFAULT)
    cfg.update(n_blocks=1, n_sub_blocks=1, dim=48, n_heads=2, window_size=8, feature_dim=16)
    model = models.DSWTransformer(**cfg)
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "rezero" in n:
                p.copy_(torch.empty_like(p).uniform_(0.5, 1.5, generator=gen))
    model = model.to(cuda0)
    ema_model = copy.deepcopy(model).eval().requires_grad_(False)
    x = torch.rand(1, 3, 32, 32, device=cuda0, generator=torch.Generator(device=cuda0).manual_seed(1)) * 2 - 1
    helper = ParameterEMA(start_step=0, beta=0.5)

    def evaluate(m):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return m(x).float()

    y0 = evaluate(ema_model)                      # fills the caches
    versions = [p._version for p in ema_model.parameters()]
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn(p.shape, device=cuda0, generator=torch.Generator(device=cuda0).manual_seed(2)) * 0.05)
    helper.step_ema(ema_model, model)             # step 0 >= start_step: averages through the kernel
    assert all(p._version > v for p, v in zip(ema_model.parameters(), versions))
    y1 = evaluate(ema_model)
    cold = copy.deepcopy(ema_model)               # new Parameter objects: nothing cached for them
    y_cold = evaluate(cold)
    assert not torch.equal(y1, y0)
    assert torch.equal(y1, y_cold)
