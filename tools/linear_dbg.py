import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiontransformer_amd import _lib
sys.argv = [sys.argv[0]]
dev = torch.device("cuda:0"); _lib.load()
import importlib.util
spec = importlib.util.spec_from_file_location("lb", os.path.join(os.path.dirname(__file__), "linear_bench.py"))
src = open(spec.origin).read().split("ok = True")[0]
exec(src)
for M in (32768, 65536, 131072):
    K, N = 180, 180
    w = torch.randn(N, K, device=dev) * 0.1
    wp, wpt = prep(w)
    a = torch.randn(M, K, device=dev)
    c = torch.empty(M, 192, device=dev, dtype=bf)
    ac = torch.full((M, 192), float("nan"), device=dev, dtype=bf)
    run(a, K, wp, None, N, c, 192, ac)
    torch.cuda.synchronize()
    bad = (ac[:, :K].float() != a.to(bf).float()).any(1)
    idx = bad.nonzero().flatten()
    print(M, "bad rows", idx.numel(), idx[:10].tolist(), idx[-5:].tolist())
    if idx.numel():
        r = idx[0].item()
        print(" row", r, "got", ac[r, :6].tolist(), "want", a[r, :6].to(bf).tolist())
        cols = (ac[r, :K].float() != a[r].to(bf).float()).nonzero().flatten()
        print(" bad cols", cols[:20].tolist())
        print(" got ", ac[r, cols[:8]].tolist())
        print(" want", a[r, cols[:8]].to(bf).tolist(), a[r, cols[:8]].tolist())
        print(" c row err", (c[r, :N].float() - (a[r].to(bf).float() @ w.to(bf).float().t())).abs().max().item())
