"""HIP-graph capture and replay of the DSWT inference forward (the static path has no host read): two graphs of
different input shapes captured one after the other, replayed in both orders, replayed after an in-place weight
update, and an eager call after the first graph is destroyed."""
import gc

import pytest
import torch

from conftest import load_golden, params_of, rel_err

pytestmark = pytest.mark.gpu


def test_graph_capture_replay_recapture(cuda0):
    from visiontransformer_amd import dswin, models
    g, m = load_golden("model_dswt")
    net = models.DSWTransformer(**m["cfg"])
    net.load_state_dict(params_of(g), strict=True)
    net = net.to(cuda0).eval()
    for sub in net.modules():
        if isinstance(sub, dswin.DeformableSparseWindowAttention):
            sub.num_win_selected = m["num_win_selected"]
    gen = torch.Generator(device=cuda0).manual_seed(1)
    xs = [torch.rand(1, 3, 32, 32, device=cuda0, generator=gen) * 2 - 1, torch.rand(1, 3, 48, 32, device=cuda0, generator=gen) * 2 - 1]

    def eager(x):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return net(x).float().clone()

    ref = [eager(x) for x in xs]
    assert torch.equal(ref[0], eager(xs[0]))           # the eager forward itself is reproducible
    graphs, outs, ins = [], [], []
    for x in xs:
        xin = x.clone()
        gr = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            with torch.cuda.graph(gr):
                y = net(xin)
        graphs.append(gr), outs.append(y), ins.append(xin)
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(y.float(), ref[len(graphs) - 1]), "replay right after capture %d" % len(graphs)
    for i in (0, 1, 0):                                 # replays in both orders after the second capture
        ins[i].copy_(xs[i])
        graphs[i].replay()
        torch.cuda.synchronize()
        assert torch.equal(outs[i].float(), ref[i]), "replay of graph %d after both captures" % i
    # new input values through the same graph
    x_new = torch.rand(1, 3, 32, 32, device=cuda0, generator=gen) * 2 - 1
    ins[0].copy_(x_new)
    graphs[0].replay()
    torch.cuda.synchronize()
    assert torch.equal(outs[0].float(), eager(x_new))
    # an in-place parameter update is seen by the replay: the graph prepares its operands from the live weights
    with torch.no_grad():
        net.layers[0].layers[0].wmsa.out_proj.weight.mul_(1.5)
        net.layers[0].layers[0].mlp1.fc2.weight.mul_(0.5)
    graphs[0].replay()
    torch.cuda.synchronize()
    new_ref = eager(x_new)
    assert rel_err(new_ref, ref[0]) > 1e-3
    assert torch.equal(outs[0].float(), new_ref)
    # the first graph goes away; nothing the eager path keeps may have lived in its memory pool
    del graphs[0], outs[0]
    gc.collect()
    torch.cuda.empty_cache()
    filler = torch.randn(64 << 20, device=cuda0)       # reuse whatever the pool gave back
    assert torch.equal(eager(x_new), new_ref)
    graphs[0].replay()
    torch.cuda.synchronize()
    del filler


def test_tiled_inference_replayed_from_graphs(cuda0):
    """parallel.GraphedForward around the network inside tiled_super_resolve: one capture per tile shape, every
    later tile of that shape a replay; bit-identical to the eager tiles"""
    from visiontransformer_amd import dswin, models, parallel
    g, m = load_golden("model_dswt")
    net = models.DSWTransformer(**m["cfg"])
    net.load_state_dict(params_of(g), strict=True)
    net = net.to(cuda0).eval()
    gen = torch.Generator(device=cuda0).manual_seed(3)
    lr = torch.rand(1, 3, 80, 96, device=cuda0, generator=gen) * 2 - 1
    graphed = parallel.GraphedForward(net)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        eager = parallel.tiled_super_resolve(net, lr, 4, tile=32, halo=8)
        replayed = parallel.tiled_super_resolve(graphed, lr, 4, tile=32, halo=8)
        again = parallel.tiled_super_resolve(graphed, lr, 4, tile=32, halo=8)
    assert len(graphed._graphs) <= 4 and len(parallel.plan_tiles(80, 96, 32, 8)) == 9
    assert torch.equal(replayed, eager) and torch.equal(again, eager)


def _attention_pair(device, C=60, heads=2, win=16):
    from visiontransformer_amd.dswin import DeformableSparseWindowAttention, WindowedMSA

    class Pair(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.wmsa = WindowedMSA(win, C, heads)
            self.sdwa = DeformableSparseWindowAttention(win, C, heads)

        def forward(self, x):
            x = x + self.wmsa(x)
            return x + self.sdwa(x)

    torch.manual_seed(5)
    m = Pair()
    with torch.no_grad():
        for mod in (m.wmsa, m.sdwa):
            mod.pe.relative_position_bias_table.normal_(0, 0.02)
        m.sdwa.sparse_discriminator[4].bias[1] += 1.0
    return m.to(device).train()


def test_graphed_training_step_against_the_eager_step(cuda0):
    """GraphedTrainStep on the WindowedMSA + DS-MSA residual pair in training mode under bf16 autocast: the replayed
    step leaves the loss, the input gradient and every parameter gradient of the eager step bit for bit (every kernel
    of the path is deterministic), follows new input values written into the captured input, and sees an in-place
    parameter update.  The host synchronises before every comparison, which is the condition under which a plain
    .mean() inside the graph returned a partial sum (parallel.graph_safe_sum)."""
    from visiontransformer_amd.parallel import GraphedTrainStep, graph_safe_mean
    pair = _attention_pair(cuda0)
    gen = torch.Generator(device=cuda0).manual_seed(3)
    B, HW, C = 2, 64, 60
    x = torch.randn(B, HW, HW, C, device=cuda0, generator=gen).requires_grad_()
    pair.sdwa.gumbel_noise = -torch.empty(B, (HW // 16) ** 2, device=cuda0).exponential_(generator=gen).log()
    loss_fn = lambda y: graph_safe_mean(y.float().square())

    def eager():
        pair.zero_grad(set_to_none=True)
        x.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = loss_fn(pair(x))
        loss.backward()
        return loss.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in pair.named_parameters() if p.grad is not None}

    ref = eager()
    step = GraphedTrainStep(pair, loss_fn)

    def check(ref, what):
        loss = step(x)
        torch.cuda.synchronize()
        assert torch.equal(loss, ref[0]), what
        assert torch.equal(x.grad, ref[1]), what
        got = {n: p.grad for n, p in pair.named_parameters() if p.grad is not None}
        assert got.keys() == ref[2].keys(), what
        for n in got:
            assert torch.equal(got[n], ref[2][n]), (what, n)

    check(ref, "first replay")
    check(ref, "second replay: gradients overwritten, not accumulated")
    with torch.no_grad():                       # new values in the captured input
        x.copy_(torch.randn(B, HW, HW, C, device=cuda0, generator=gen))
    ref2 = eager()
    assert not torch.equal(ref2[1], ref[1])
    check(ref2, "new input values")
    with torch.no_grad():                       # an optimizer's in-place update
        pair.wmsa.qkv.weight.mul_(1.25)
        pair.sdwa.out_proj.weight.mul_(0.75)
    ref3 = eager()
    check(ref3, "after an in-place parameter update")
    assert len(step._graphs) == 1
