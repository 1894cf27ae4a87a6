#!/usr/bin/env python3
"""dsw_linear against torch (values) and against the GEMM library (time) on the path's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiontransformer_amd import _lib
import bench

dev = torch.device("cuda:0")
_lib.load()
bf, f32 = torch.bfloat16, torch.float32


def ceil64(n):
    return (n + 63) // 64 * 64


def prep(w):
    N, K = w.shape
    wp = torch.empty(ceil64(N), ceil64(K), device=dev, dtype=bf)
    wpt = torch.empty(ceil64(K), ceil64(N), device=dev, dtype=bf)
    _lib.call("dsw_weight_prep", _lib.ptr(w), _lib.dtype_code(w.dtype), N, K, _lib.ptr(wp), _lib.ptr(wpt), _lib.stream(dev))
    return wp, wpt


def run(a, K, wp, bias, N, c, ncols, a_copy=None):
    _lib.call("dsw_linear", _lib.ptr(a), _lib.dtype_code(a.dtype), a.stride(0), a.shape[0], K, _lib.ptr(wp), wp.shape[0], wp.stride(0),
              _lib.ptr(bias) if bias is not None else None, N, _lib.ptr(c), _lib.dtype_code(c.dtype), c.stride(0), ncols, 0,
              _lib.ptr(a_copy) if a_copy is not None else None, a_copy.stride(0) if a_copy is not None else 0, None, 0, _lib.stream(dev))


def case(M, K, N, a_dt, c_dt, pad_a, pad_c, copy, dgrad=False):
    torch.manual_seed(0)
    w = torch.randn(N, K, device=dev) * 0.1
    bias = torch.randn(N, device=dev)
    wp, wpt = prep(w)
    lda = ceil64(K) if pad_a else K
    a_full = torch.zeros(M, lda, device=dev, dtype=a_dt)
    a_full[:, :K] = torch.randn(M, K, device=dev).to(a_dt)
    ncols = ceil64(N) if pad_c else N
    c = torch.full((M, ncols), float("nan"), device=dev, dtype=c_dt)
    a_copy = torch.full((M, ceil64(K)), float("nan"), device=dev, dtype=bf) if copy else None
    Kk = lda if a_dt == bf else K   # bf16 rows are taken with their zero padding
    run(a_full, Kk, wp, bias, N, c, ncols, a_copy)
    ref = a_full[:, :K].to(bf).float() @ w.to(bf).float().t() + bias
    err = ((c[:, :N].float() - ref).abs().max() / ref.abs().max()).item()
    ok = err < (1e-2 if c_dt == bf else 1e-5 * K)
    if not (ncols == N or bool((c[:, N:] == 0).all())):
        ok = False; print("   padding columns of c not zero")
    if copy and not torch.equal(a_copy[:, :K], a_full[:, :K].to(bf)):
        ok = False; print("   a_copy differs", (a_copy[:, :K].float() - a_full[:, :K].to(bf).float()).abs().max().item())
    if copy and not bool((a_copy[:, K:] == 0).all()):
        ok = False; print("   a_copy padding not zero")
    # library time on the padded operands vs this kernel
    a_lib = torch.zeros(M, ceil64(K), device=dev, dtype=bf)
    bp = torch.zeros(ncols, device=dev, dtype=bf)
    t_lib = bench.time_calls(lambda: torch.addmm(bp, a_lib, wp[:ncols].t()), 20)
    t_me = bench.time_calls(lambda: run(a_full, Kk, wp, bias, N, c, ncols, a_copy), 20)
    byt = M * (K * a_full.element_size() + ncols * c.element_size() + (ceil64(K) * 2 if copy else 0))
    print("M=%6d K=%3d N=%3d a=%s c=%s pad_a=%d pad_c=%d copy=%d : err %.2e %s  lib %6.1f us  this %6.1f us  (%.0f GB/s)" % (
        M, K, N, str(a_dt)[6:10], str(c_dt)[6:10], pad_a, pad_c, copy, err, "ok" if ok else "WRONG", t_lib * 1e6, t_me * 1e6,
        byt / t_me / 1e9), flush=True)
    return ok


def wcase(M, K, N):
    torch.manual_seed(1)
    Kp, Np = ceil64(K), ceil64(N)
    x = torch.zeros(M, Kp, device=dev, dtype=bf); x[:, :K] = torch.randn(M, K, device=dev).to(bf)
    dy = torch.zeros(M, Np, device=dev, dtype=bf); dy[:, :N] = torch.randn(M, N, device=dev).to(bf)
    dw = torch.full((N, K), float("nan"), device=dev); db = torch.full((N,), float("nan"), device=dev) if K < Kp else None
    ws = torch.empty(_lib.load().dsw_linear_wgrad_workspace(M, N, K), device=dev)
    f = lambda: _lib.call("dsw_linear_wgrad", _lib.ptr(dy), Np, _lib.ptr(x), Kp, M, N, K, _lib.ptr(dw), _lib.ptr(db) if db is not None else None,
                          _lib.ptr(ws), None, 0, _lib.stream(dev))
    f()
    ref = dy[:, :N].double().t() @ x[:, :K].double()
    err = ((dw.double() - ref).abs().max() / ref.abs().max()).item()
    ok = err < 1e-5
    if db is not None:
        rb = dy[:, :N].double().sum(0)
        eb = ((db.double() - rb).abs().max() / rb.abs().max()).item()
        ok = ok and eb < 1e-5
    else:
        eb = float("nan")
    S = M // 2048 if M % 2048 == 0 and M >= 16384 else 1
    lib = (lambda: torch.bmm(dy.view(S, M // S, Np).transpose(1, 2), x.view(S, M // S, Kp)).sum(0, dtype=torch.float32))
    t_lib = bench.time_calls(lib, 20); t_me = bench.time_calls(f, 20)
    print("wgrad M=%6d K=%3d N=%3d : err %.1e / %.1e %s  lib(bmm+sum) %6.1f us  this %6.1f us (%.0f GB/s)" % (
        M, K, N, err, eb, "ok" if ok else "WRONG", t_lib * 1e6, t_me * 1e6, M * (Kp + Np) * 2 / t_me / 1e9), flush=True)
    return ok


ok = True
if os.environ.get('WONLY'):
    for M in (131072, 32768):
        wcase(M, 180, 540); wcase(M, 180, 180)
    sys.exit(0)
if os.environ.get('MSFT'):   # the projections of workload C's fuse block (dim 216) and level block (dim 96)
    for (M, dims) in ((131072, (216, 648, 432)), (32768, (96, 288, 192))):
        C, C3, Hd = dims
        for (K, N) in ((C, C3), (C, C), (C, Hd), (Hd, C), (C3, C)):
            case(M, K, N, bf, bf, 1, 1, 0)
        case(M, Hd, C, f32, bf, 0, 1, 1)   # a foreign fp32 input of the second MLP projection, with the bf16 copy
        case(M, C3, C, bf, f32, 1, 0, 0)
        for (K, N) in ((C, C3), (C, C), (C, Hd), (Hd, C)):
            wcase(M, K, N)
    sys.exit(0)
ok &= wcase(300, 180, 540)
ok &= wcase(1000, 96, 100)
ok &= wcase(4099 if False else 4096 + 24, 192, 64)
for M in (131072, 32768):
    ok &= wcase(M, 180, 540)
    ok &= wcase(M, 180, 180)
# small odd cases first
ok &= case(300, 180, 540, f32, bf, 0, 1, 1)
ok &= case(300, 96, 288, bf, bf, 1, 1, 0)
ok &= case(257, 64, 64, f32, f32, 0, 0, 1)
ok &= case(1000, 544, 180, bf, f32, 1, 0, 0)   # 3 slabs of 192 (Kp = 576), persistent accumulators
ok &= case(1000, 256, 320, bf, bf, 1, 1, 0)    # 2 slabs of 128, two block columns
ok &= case(515, 320, 100, f32, bf, 0, 0, 1)    # 5 slabs of 64
# the path's shapes
for M in (131072, 32768):
    ok &= case(M, 180, 540, f32, bf, 0, 1, 1)   # W-MSA qkv from the fp32 stream
    ok &= case(M, 180, 540, bf, bf, 1, 1, 0)    # DS-MSA in_proj from gathered windows
    ok &= case(M, 180, 180, bf, bf, 1, 1, 0)    # out_proj
    ok &= case(M, 540, 180, bf, f32, 1, 0, 0)   # qkv input gradient into the fp32 stream
    ok &= case(M, 540, 180, bf, bf, 1, 1, 0)    # in_proj input gradient
    ok &= case(M, 180, 180, f32, bf, 0, 1, 1)   # out_proj input gradient from a foreign fp32 dy
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
