"""Does a multi-block torch reduction replay correctly from a HIP graph when the host synchronises between replays?"""
import torch
dev = torch.device("cuda:0")
a = torch.randn(2 * 64 * 64 * 60, device=dev)
for name, fn in (("sum", lambda t: t.sum()), ("sqmean", lambda t: t.float().square().mean()), ("abs-max", lambda t: t.abs().max())):
    ref = fn(a).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(a)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn(a)
    res = []
    for i in range(6):
        g.replay()
        torch.cuda.synchronize()
        res.append(bool(torch.equal(out, ref)))
        # eager work between replays: another multi-block reduction
        torch.equal(a, a)
    print(name, res, float(out), float(ref), flush=True)
