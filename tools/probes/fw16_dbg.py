import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visiontransformer_amd import functional as K
from visiontransformer_amd._lib import QKV_INTERLEAVED
dev = torch.device("cuda:0")
torch.manual_seed(0)
w, C, heads = 16, 180, 6
qkv = torch.randn(1, 32, 48, 3 * C, device=dev).bfloat16()
tab = torch.randn(961, heads, device=dev) * 0.5
with torch.no_grad():
    a = K.window_attention(qkv, tab, w, heads, QKV_INTERLEAVED, grid=(1, 32, 48))
    b = K.window_attention(qkv, tab, w, heads, QKV_INTERLEAVED, grid=(1, 32, 48))
    os.environ["DSW_ATTN_FWD_GENERAL"] = "1"
    c = K.window_attention(qkv, tab, w, heads, QKV_INTERLEAVED, grid=(1, 32, 48))
for name, x, y in (("a-b", a, b), ("a-c", a, c), ("b-c", b, c)):
    dlt = (x.float() - y.float()).abs()
    bad = (dlt > 0).nonzero() if name == 'a-b' else (dlt > 2e-2).nonzero()
    print(name, "max", float(dlt.max()), "nbad", bad.shape[0])
    if bad.shape[0]:
        print(" cols", sorted(set((bad[:, 3] % 30).tolist())), "heads", sorted(set((bad[:, 3] // 30).tolist())))
        print(" n per col", torch.bincount(bad[:, 3] % 30, minlength=30).tolist())
        print(" rows", sorted(set(bad[:, 1].tolist()))[:40], "xs", sorted(set(bad[:, 2].tolist()))[:50])
