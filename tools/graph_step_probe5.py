import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_graphs import _attention_pair
from visiontransformer_amd.parallel import GraphedTrainStep
mode = sys.argv[1]
cuda0 = torch.device("cuda:0")
pair = _attention_pair(cuda0)
gen = torch.Generator(device=cuda0).manual_seed(3)
B, HW, C = 2, 64, 60
x = torch.randn(B, HW, HW, C, device=cuda0, generator=gen).requires_grad_()
pair.sdwa.gumbel_noise = -torch.empty(B, (HW // 16) ** 2, device=cuda0).exponential_(generator=gen).log()
loss_fn = (lambda y: y.float().square().view(-1, 1024).sum(1).sum() / y.numel()) if os.environ.get("TWOSTAGE") else (lambda y: y.float().square().mean())
def eager():
    pair.zero_grad(set_to_none=True)
    x.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = loss_fn(pair(x))
    loss.backward()
    return loss.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in pair.named_parameters() if p.grad is not None}
ref = eager()
step = GraphedTrainStep(pair, loss_fn)
def check(what):
    loss = step(x)
    if mode == "devsync":
        torch.cuda.synchronize()
    elif mode == "streamsync":
        torch.cuda.current_stream().synchronize()
    ok = [bool(torch.equal(loss, ref[0])), bool(torch.equal(x.grad, ref[1]))]
    if mode != "nopgrad":
        got = {n: p.grad for n, p in pair.named_parameters() if p.grad is not None}
        bad = [n for n in got if not torch.equal(got[n], ref[2][n])]
    else:
        bad = "-"
    print(what, ok, bad, flush=True)
for i in range(4):
    check("replay %d" % i)
