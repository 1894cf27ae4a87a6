#!/usr/bin/env python3
"""Benchmark of the DS-MSA / windowed-MSA hot path on MI355X.

Workload ("config B": the DS-MSA forward + backward case at batch 32): one DSWT attention pair - WindowedMSA followed
by DeformableSparseWindowAttention, default model config (C=180, 6 heads, 16x16 windows) - forward
and backward under bf16 autocast on a batch of 32 LR patches of 64x64 (activations [32,64,64,180]
fp32, as the reference's ViTBlock feeds them).  A step = one forward + backward of that pair.
value = patches/s over all ranks.  Synthetic inputs, random-init weights (gates biased open so
that the selected windows pass), explicit Gumbel noise.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, the pair in a data-parallel wrapper over RCCL (one flat gradient all-reduce per step)
(backend "nccl"), per-GPU batch fixed (weak scaling).

The JSON line also carries
  roofline     - the dominant kernel (fused window-attention backward at the WMSA shape),
                 algorithmic FLOPs / its measured launch time against the dense bf16 MFMA peak;
  kernels      - the same measurement for every kernel of the path (GB/s or TFLOP/s);
  sr_inference - the other half of the headline metric: x4 super-resolution of a 960x540 frame by the whole
                 DSWT network, tiles sharded over the ranks (strong scaling), output megapixels / s;
  cpu_baseline - the CPU oracle (a PyTorch restatement of the reference) timed on this host's
                 cores on a bounded sample of the same workload (N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16

B_PER_GPU, HW, C, HEADS, WIN = 32, 64, 180, 6, 16
# the kernel `roofline` is about, as rocprofv3 names it: window-attention backward at the WindowedMSA shape
DOMINANT_KERNEL = "win_attn_bwd_w16_d30_kernel<true>"


def build_pair(device, seed=0):
    from visiontransformer_amd.dswin import DeformableSparseWindowAttention, WindowedMSA
    torch.manual_seed(seed)

    class Pair(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.wmsa = WindowedMSA(WIN, C, HEADS)
            self.sdwa = DeformableSparseWindowAttention(WIN, C, HEADS)

        def forward(self, x):
            # residual form of the reference's ViTBlock (DSwinT_arch.py:355-356): the fp32 stream feeds both
            # attentions; x + branch(x) with the addition inside the branch's closing kernel (BENCH_UNFUSED_RESIDUAL=1:
            # as two torch additions)
            if os.environ.get("BENCH_UNFUSED_RESIDUAL") == "1":
                x = x + self.wmsa(x)
                return x + self.sdwa(x)
            x = self.wmsa(x, residual=(x, None))
            return self.sdwa(x, residual=(x, None))

    m = Pair()
    with torch.no_grad():
        for mod in (m.wmsa, m.sdwa):
            mod.pe.relative_position_bias_table.normal_(0, 0.02)
        m.sdwa.sparse_discriminator[4].bias[1] += 1.0   # "dense" run: gates open
        m.sdwa.sparse_discriminator[4].weight.mul_(4.0)  # spread offsets over the window
    return m.to(device).train()


def time_calls(fn, iters, warm=3, rounds=1):
    """seconds per call: events on the launch stream around `iters` calls; with rounds > 1 the median of that many
    such measurements (the first one after a change of kernel runs several per cent slower than the ones behind it)"""
    for _ in range(warm):
        fn()
    res = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e-3 / iters)
    res.sort()
    return res[len(res) // 2]


def kernel_rooflines(device, iters=20, warm=3, k=4, attention=True):
    """Per-kernel achieved rate at config B shapes, measured with events on the launch stream."""
    torch.manual_seed(1)
    out = {}
    if attention:
        _attention_rooflines(device, iters, warm, out)
    # DS-MSA side: k windows per image (config B: 4), all passing, random corners
    x = torch.randn(B_PER_GPU, HW, HW, C, device=device)
    _window_rooflines(device, iters, warm, k, x, out)
    _projection_rooflines(device, iters, warm, x, out)
    _mlp_roofline(device, iters, warm, x, out)
    for v in out.values():
        if "achieved" in v:
            v["frac"] = v["achieved"] / v["peak"]
    return out


def _attention_rooflines(device, iters, warm, out):
    from visiontransformer_amd import functional as K
    from visiontransformer_amd._lib import QKV_INTERLEAVED
    L = WIN * WIN
    T = B_PER_GPU * HW * HW
    bf = torch.bfloat16
    table = torch.randn((2 * WIN - 1) ** 2, HEADS, device=device) * 0.02
    # windowed MSA shape: all 512 windows, image grid addressing, on the row pitches the modules hand the kernels
    # (functional.linear(..., pad_out=True) leaves q|k|v rows of 576 elements, the attention's output and the
    # gradient coming back from out_proj rows of 192)
    P3, P1 = K._ceil_pad(3 * C), K._ceil_pad(C)
    qbuf = K._mark_padded(torch.zeros(T, P3, device=device, dtype=bf))
    qbuf[:, :3 * C] = torch.randn(T, 3 * C, device=device).to(bf)
    qkv = qbuf[:, :3 * C].view(B_PER_GPU, HW, HW, 3 * C).requires_grad_()
    tab = table.clone().requires_grad_()
    o = K.window_attention(qkv, tab, WIN, HEADS, QKV_INTERLEAVED, grid=(B_PER_GPU, HW, HW), pad_out=True)
    gbuf = torch.zeros(T, P1, device=device, dtype=bf)
    gbuf[:, :C] = torch.randn(T, C, device=device).to(bf)
    do = gbuf[:, :C].view(B_PER_GPU, HW, HW, C)
    with torch.no_grad():
        t_f = time_calls(lambda: K.window_attention(qkv, table, WIN, HEADS, QKV_INTERLEAVED, grid=(B_PER_GPU, HW, HW), pad_out=True),
                         iters, max(warm, 10), rounds=5 if iters > 2 else 1)
    t_b = time_calls(lambda: torch.autograd.grad(o, (qkv, tab), do, retain_graph=True), iters, max(warm, 10) if iters > 2 else warm,
                     rounds=5 if iters > 2 else 1)
    flops_f, flops_b = 4.0 * T * L * C, 8.0 * T * L * C
    # algorithmic HBM bytes (bf16): forward reads qkv, writes out (+ lse); backward reads qkv, out, dout
    # (+ lse), writes dqkv.  At head dim 30 both passes need more time on the HBM roofline than on the
    # MFMA one (128 flop/byte against the machine's 312), so
    # HBM is the roofline that bounds them; the MFMA rate is reported next to it.
    by_f = T * (3 * C + C) * 2 + T * HEADS * 4
    by_b = T * (3 * C + C + C + 3 * C) * 2 + T * HEADS * 4
    for name, t, fl, by in (("win_attn_fwd", t_f, flops_f, by_f), ("win_attn_bwd", t_b, flops_b, by_b)):
        out[name] = dict(bound="hbm", achieved=by / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", us=t * 1e6, algorithmic=by,
                         flops=fl, mfma_tflops=fl / t / 1e12, mfma_frac=fl / t / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                         row_pitch="q|k|v %d, out / dout %d elements (what the modules use)" % (P3, P1))
    out["win_attn_bwd"]["note"] = "attention backward kernel + the fixed-order sum of the bias table's partials"
    # inference: q|k|v projection inside the attention kernel (q|k|v never written).  Reads the padded bf16 rows of
    # x, writes out; the projection's 2*T*C*3C flops join the attention's (the kernel itself multiplies the padded
    # 192 x 3*32*heads operands).  Its intensity, (4*L*C + 6*C*C) / (2*C*2) = 346 flop/byte, is above the machine's
    # 312: the matrix cores are the roofline that bounds it.
    Cp = (C + 63) // 64 * 64
    xp = K._mark_padded(torch.zeros(T, Cp, device=device, dtype=bf))
    xp[:, :C] = torch.randn(T, C, device=device).to(bf)
    xv = xp[:, :C].view(B_PER_GPU, HW, HW, C)
    lin = torch.nn.Linear(C, 3 * C).to(device)
    with torch.no_grad(), torch.autocast("cuda", dtype=bf):
        fused = lambda: K.projected_window_attention(xv, lin, table, WIN, HEADS, QKV_INTERLEAVED, grid=(B_PER_GPU, HW, HW))
        if fused() is not None:
            t_p = time_calls(fused, iters, warm)
            t_2 = time_calls(lambda: K.window_attention(K.linear(xv, lin, pad_out=True), table, WIN, HEADS, QKV_INTERLEAVED,
                                                        grid=(B_PER_GPU, HW, HW)), iters, warm)
            fl_p = flops_f + 2.0 * T * C * 3 * C
            by_p = T * (C + C) * 2
            out["win_attn_proj_fwd"] = dict(bound="mfma", achieved=fl_p / t_p / 1e12, peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                                            us=t_p * 1e6, flops=fl_p, algorithmic=by_p, hbm_frac=by_p / t_p / 1e9 / HBM_PEAK_GBS,
                                            separate_kernels_us=t_2 * 1e6,
                                            note="inference forward; same bits as dsw_linear + dsw_win_attn_fwd")


def _window_rooflines(device, iters, warm, k, x, out):
    """Selection, gather and scatter kernels, called through the C ABI with preallocated buffers
    (the autograd wrappers cost more host time per call than these kernels run)."""
    from visiontransformer_amd import functional as K
    from visiontransformer_amd._lib import call, dtype_code, ptr, stream
    L = WIN * WIN
    bf = torch.bfloat16
    B, H, W = B_PER_GPU, HW, HW
    sd = torch.randn(B, 4, H // WIN, W // WIN, device=device)
    sd[:, 1] = 4.0
    sig = sd[:, 1:].sigmoid()
    sel = K.select_windows(sd, sig, None, 1.0, k, WIN, H, W)
    n = sel.n
    t_s = time_calls(lambda: K.select_windows(sd, sig, None, 1.0, k, WIN, H, W), iters, warm)
    out["select"] = dict(bound="latency", us=t_s * 1e6, note="two kernels + the host read of the window count (the modules use the variant that leaves the count on the device)")
    st = stream(device)
    f32c, bfc = dtype_code(torch.float32), dtype_code(bf)
    g = torch.empty(n, L, C, device=device, dtype=bf)
    gather_f = lambda: call("dsw_gather_fwd", ptr(x), f32c, ptr(sel.win_b), ptr(sel.win_h1), ptr(sel.win_w1), n, B, H, W, C,
                            WIN, ptr(g), bfc, C, st)
    t_g = time_calls(gather_f, iters, warm)
    dg = torch.randn(n, L, C, device=device).to(bf)
    dx = torch.empty_like(x)
    gather_b = lambda: call("dsw_gather_bwd", ptr(dg), bfc, ptr(sel.cellmap), ptr(sel.win_h1), ptr(sel.win_w1), n, B, H, W, C,
                            WIN, C, ptr(dx), f32c, 0, st)
    t_gb = time_calls(gather_b, iters, warm)
    # the form the training step runs: the windows' gradients added into the input-gradient map that the other
    # branches of the module share (only the pixels a window covers are read and written)
    dx.zero_()
    gather_ba = lambda: call("dsw_gather_bwd", ptr(dg), bfc, ptr(sel.cellmap), ptr(sel.win_h1), ptr(sel.win_w1), n, B, H, W, C,
                             WIN, C, ptr(dx), f32c, 1, st)
    t_gba = time_calls(gather_ba, iters, warm)
    by_gba = n * C * (L * 2 + (WIN + 1) ** 2 * 4 * 2)
    by_g = n * C * ((WIN + 1) ** 2 * 4 + L * 2)
    by_gb = n * C * L * 2 + B * H * W * C * 4
    out["gather_fwd"] = dict(bound="hbm", achieved=by_g / t_g / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", us=t_g * 1e6,
                             algorithmic=by_g)
    out["gather_bwd"] = dict(bound="hbm", achieved=by_gb / t_gb / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", us=t_gb * 1e6,
                             algorithmic=by_gb)
    out["gather_bwd_accumulate"] = dict(bound="hbm", achieved=by_gba / t_gba / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", us=t_gba * 1e6,
                                        algorithmic=by_gba, note="adds into an existing map; bytes = the windows' gradients + read and write of the "
                                        "(w+1)^2 pixels each window covers")
    after = torch.randn(n, L, C, device=device).to(bf)
    y = torch.empty_like(x)
    gd, sdl = sel.gates_delta.float().contiguous(), sel.scores_delta.float().contiguous()
    scatter_f = lambda: call("dsw_scatter_mean_fwd", ptr(x), f32c, ptr(after), bfc, ptr(gd), ptr(sdl), ptr(sel.cellmap),
                             ptr(sel.win_h1), ptr(sel.win_w1), n, B, H, W, C, WIN, C, ptr(y), f32c, st)
    t_sc = time_calls(scatter_f, iters, warm)
    dy = torch.randn_like(x)
    d_after = torch.empty_like(after)
    dgs = torch.empty(2, n, device=device)
    partial = torch.empty(2 * n * WIN, device=device)
    scatter_b = lambda: call("dsw_scatter_mean_bwd", ptr(dy), f32c, ptr(x), f32c, ptr(after), bfc, ptr(gd), ptr(sdl),
                             ptr(sel.cellmap), ptr(sel.win_b), ptr(sel.win_h1), ptr(sel.win_w1), n, B, H, W, C, WIN, C,
                             ptr(dx), f32c, 0, ptr(d_after), bfc, C, ptr(dgs[0]), ptr(dgs[1]), ptr(partial), None, st)
    t_scb = time_calls(scatter_b, iters, warm)
    by_sc = n * C * L * 2 + 2 * B * H * W * C * 4
    by_scb = 2 * B * H * W * C * 4 + n * C * L * (2 + 2) + n * C * (WIN + 1) ** 2 * 4
    out["scatter_mean_fwd"] = dict(bound="hbm", achieved=by_sc / t_sc / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                   us=t_sc * 1e6, algorithmic=by_sc)
    out["scatter_mean_bwd"] = dict(bound="hbm", achieved=by_scb / t_scb / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                   us=t_scb * 1e6, algorithmic=by_scb, note="three kernels: dx, window side, reduce")


def _projection_rooflines(device, iters, warm, x, out):
    """The projection kernels at the windowed-MSA shapes (all tokens), through the C ABI: qkv from the
    fp32 stream (with the bf16 copy for the weight gradient), its input gradient back into an fp32
    buffer, and its weight + bias gradient."""
    from visiontransformer_amd._lib import call, dtype_code, load, ptr, stream
    bf = torch.bfloat16
    M, K, N = x.numel() // C, C, 3 * C
    Kp, Np = (K + 63) // 64 * 64, (N + 63) // 64 * 64
    st = stream(device)
    f32c, bfc = dtype_code(torch.float32), dtype_code(bf)
    w = torch.randn(N, K, device=device) * 0.05
    bias = torch.randn(N, device=device)
    w_p = torch.empty(Np, Kp, device=device, dtype=bf)
    w_pt = torch.empty(Kp, Np, device=device, dtype=bf)
    call("dsw_weight_prep", ptr(w), f32c, N, K, ptr(w_p), ptr(w_pt), st)
    x2 = x.view(M, K)
    y = torch.empty(M, Np, device=device, dtype=bf)
    xk = torch.empty(M, Kp, device=device, dtype=bf)
    fwd = lambda: call("dsw_linear", ptr(x2), f32c, K, M, K, ptr(w_p), Np, Kp, ptr(bias), N, ptr(y), bfc, Np, Np, 0, ptr(xk), Kp, None, 0, st)
    t_f = time_calls(fwd, iters, warm)
    dx = torch.empty(M, K, device=device)
    dgrad = lambda: call("dsw_linear", ptr(y), bfc, Np, M, Np, ptr(w_pt), Kp, Np, None, 0, ptr(dx), f32c, K, K, 0, None, 0, None, 0, st)
    t_d = time_calls(dgrad, iters, warm)
    dw, db = torch.empty(N, K, device=device), torch.empty(N, device=device)
    ws = torch.empty(load().dsw_linear_wgrad_workspace(M, N, K), device=device)
    wgrad = lambda: call("dsw_linear_wgrad", ptr(y), Np, ptr(xk), Kp, M, N, K, ptr(dw), ptr(db), ptr(ws), None, 0, st)
    t_w = time_calls(wgrad, iters, warm)
    by_f = M * (K * 4 + Np * 2 + Kp * 2)
    by_d = M * (Np * 2 + K * 4)
    by_w = M * (Np * 2 + Kp * 2)
    for name, t, by, note in (("linear_qkv_fwd", t_f, by_f, "fp32 in, bf16 out + bf16 copy of the input"),
                              ("linear_qkv_dgrad", t_d, by_d, "bf16 in, fp32 out"),
                              ("linear_qkv_wgrad", t_w, by_w, "two kernels: split-K product, fixed-order sum")):
        out[name] = dict(bound="hbm", achieved=by / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", us=t * 1e6, algorithmic=by, note=note)


def _mlp_roofline(device, iters, warm, x, out):
    """The gated MLP of a DSWT block with its norm and its ReZero residual as one kernel (inference; the whole branch of
    DSwinT_arch.py:357) through the C ABI, at the bench's token count: x read once, y written once."""
    from visiontransformer_amd import functional as K
    from visiontransformer_amd._lib import call, ptr, stream
    M, Hd = x.numel() // C, 2 * C
    g = torch.Generator(device="cpu").manual_seed(3)
    mk = lambda *shape: (torch.randn(*shape, generator=g) * 0.05).to(device)
    ops = K.mlp_gated_operands(mk(Hd, C), mk(Hd, C), mk(Hd), mk(Hd), mk(C, Hd), mk(C), 1.0 + mk(C))
    w1c, b1c, w2c, b2, nch = ops
    x2 = x.view(M, C)
    y = torch.empty_like(x2)
    r = torch.full((1,), 0.2, device=device)
    eps = ctypes.c_float(1e-6)
    run = lambda: call("dsw_mlp_gated_residual", ptr(x2), C, M, C, eps, ptr(w1c), ptr(b1c), ptr(w2c), ptr(b2), nch, ptr(r), ptr(y), C,
                       stream(device))
    t = time_calls(run, iters, warm)
    by = M * C * 8
    fl = 2.0 * M * (192 * 2 * 32 * nch + 32 * nch * 192)      # the MFMA work as issued: K and C padded to 192, hidden to 32 * chunks
    out["mlp_gated_residual"] = dict(bound="hbm", achieved=by / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", us=t * 1e6, algorithmic=by,
                                     flops=fl, mfma_frac=fl / t / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                     note="inference: rms norm + fc1 (gate, content) + gated SiLU + fc2 + ReZero residual in one kernel; "
                                          "the two-kernel form moves 2.6 x the bytes")


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes over
    `bench.py --pmc-target` (FETCH_SIZE doubled as gfx950 requires, plus WRITE_SIZE; see
    tools/pmc_summary.py).  Counters cannot be read from inside this process, so the figure is
    the one measured by those passes at the same shapes; None when the summary is absent."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
    try:
        with open(path) as f:
            return json.load(f)[kernel]["traffic"]
    except (OSError, KeyError, ValueError):
        return None   # no pass on record for this kernel name: the field stays empty rather than stale


def cpu_baseline(sample_batch=8, budget_s=15.0):
    """The CPU oracle on this host's cores: the same residual pair, fp32, forward + backward, on a
    bounded sample (a quarter of the batch, repeated until about budget_s seconds have passed)
    so that the default run stays within minutes."""
    from oracle import dswin_oracle as D
    cores = os.cpu_count() or 1
    threads = min(cores, 16)
    torch.set_num_threads(threads)
    m = build_pair(torch.device("cpu"))
    Pw = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.wmsa.state_dict().items()}
    Ps = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.sdwa.state_dict().items()}
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(sample_batch, HW, HW, C, generator=gen).requires_grad_()
    noise = D.gumbel_noise((sample_batch, (HW // WIN) ** 2), gen)

    def step():
        y = x + D.wmsa_forward(Pw, x, WIN, HEADS)
        y = y + D.dsmsa_forward(Ps, y, WIN, HEADS, training=True, noise=noise)
        y.sum().backward()

    step()
    iters, t0 = 0, time.perf_counter()
    while iters < 2 or time.perf_counter() - t0 < budget_s:
        step()
        iters += 1
    dt = (time.perf_counter() - t0) / iters
    return dict(value=sample_batch / dt, unit="patches/s", cores=threads, kind="port",
                sample="CPU oracle (PyTorch restatement of DSwinT_arch.py WindowedMSA + DS-MSA, residual pair), fp32, "
                       "fwd+bwd, batch %d of 64x64x180, %d iters, %.2f s/iter" % (sample_batch, iters, dt))


def train_network(device, rank, world, grouped, steps=5, warm=3):
    """Whole-network x4 training steps, what toolkit/init.py:131 + toolkit/model_helper2.py:105-131 run: the DSWT
    default network and MSFT at the headline configuration (dim 192, 2 levels, 8x8 windows, attention dropout 0.1),
    batch 32 of 64x64 LR patches per GPU, bf16 autocast, L1 loss, the package's Adam; at N > 1 under
    DistributedDataParallel as parallel.data_parallel builds it (RCCL, 25 MB buckets overlapping the backward pass,
    gradients as views of the buckets).  patches/s over all ranks, time = max over ranks, barriers both sides."""
    import gc
    import torch.distributed as dist
    from visiontransformer_amd import models, optim, parallel
    res = {}
    for name, build in (("dswt_default", lambda: models.DSWTransformer(**models.DSWT_DEFAULT)),
                        ("msft_config_c", lambda: models.MSFTransformer(**models.MSFT_CONFIG_C))):
        try:
            torch.manual_seed(0)
            net = build().to(device).train()
            n_par = sum(p.numel() for p in net.parameters())
            model = parallel.data_parallel(net, device) if grouped else net
            opt = optim.Adam(net.parameters(), lr=2e-4, betas=(0.9, 0.99))
            gen = torch.Generator(device=device).manual_seed(200 + rank)
            x = torch.rand(B_PER_GPU, 3, HW, HW, device=device, generator=gen) * 2 - 1
            target = torch.rand(B_PER_GPU, 3, 4 * HW, 4 * HW, device=device, generator=gen) * 2 - 1

            def step():
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = model(x)
                torch.nn.functional.l1_loss(y.float(), target).backward()
                opt.step()
                opt.zero_grad(set_to_none=True)

            for _ in range(warm):
                step()
            torch.cuda.synchronize()
            gc.collect()
            if grouped:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            if grouped:
                dist.barrier()
            dt = time.perf_counter() - t0
            if grouped:
                t = torch.tensor([dt], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            res[name] = {"metric": "training patches/sec (x4 SR, fwd+bwd+Adam)", "value": world * B_PER_GPU * steps / dt,
                         "unit": "patches/s", "n_gpus": world, "ms_per_step": dt / steps * 1e3, "scaling": "weak", "dtype": "bf16",
                         "data": "synthetic", "parameters": n_par, "gradient_bytes_all_reduced_per_step": 4 * n_par if grouped else 0,
                         "parallelism": ("dp%d: DistributedDataParallel over RCCL, gradient_as_bucket_view" % world) if grouped else "one GPU",
                         "config": {"workload": "%s, batch %d x 64x64 LR per GPU" % (name, B_PER_GPU)}}
            del model, net, opt, x, target
        except Exception as e:  # reported in place of the number
            res[name] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
        gc.collect()
        torch.cuda.empty_cache()
    return res


def dswt_flops_per_frame(cfg, H, W, pass_share=0.25):
    """multiply-add flops (x2) of one forward of the DSWT network on an H x W LR frame (a multiple of the window):
    projections, attention products, gated MLP, the 3x3 convolutions of the trunk and of the x4 reconstruction tail;
    DS-MSA attends pass_share of the windows at most (num_win_selected, DSwinT_arch.py:179)."""
    Cc, w, fd = cfg["dim"], cfg["window_size"], cfg["feature_dim"]
    L, T = w * w, H * W
    attn = 8 * Cc * Cc + 4 * L * Cc                      # q|k|v + out projections, q k^T and p v
    dsmsa = pass_share * attn + 2 * 9 * Cc * 4 + 2 * 9 * Cc   # + sparse discriminator's first convolution, out_conv
    mlp = 12 * Cc * Cc                                   # gate + content (C -> 2C each), fc2 (2C -> C)
    block = attn + dsmsa + mlp
    trunk = cfg["n_blocks"] * (cfg["n_sub_blocks"] * block + 2 * 9 * Cc * Cc) + 2 * 9 * Cc * Cc   # out_layer per block, feature_transition
    stem = 2 * 9 * 3 * Cc
    tail = 2 * 9 * (Cc * fd + fd * fd) + 2 * 9 * fd * 4 * fd * (1 + 4) + 16 * 2 * 9 * (fd * (fd // 2) + (fd // 2) * 3)
    return float(T) * (trunk + stem + tail)


SR_LR = (540, 960)  # x4 -> 3840 x 2160


def sr_inference(device, rank, world, grouped=False, iters=5):
    """x4 super-resolution of one 960x540 frame by the DSWT default network (eval, bf16 autocast), the frame
    cut into the balanced tile plan for `world` ranks (one rank: the whole frame as one tile), every
    rank computing its tiles and one all-reduce assembling the image: output megapixels / s over
    the whole job (strong scaling: the frame is fixed).  Time = max over ranks, barriers both sides."""
    import torch.distributed as dist
    from visiontransformer_amd.models import DSWT_DEFAULT, dswt_from_config
    from visiontransformer_amd.parallel import GraphedForward, balanced_grid, plan_tiles, tiled_super_resolve
    torch.manual_seed(0)
    model = dswt_from_config(DSWT_DEFAULT).to(device).eval()
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if "rezero" in n_:
                p_.fill_(0.2)    # the ReZero gates are zero at init: every block would add nothing to the stream and the tile check below would see the two ends only
    gen = torch.Generator(device=device).manual_seed(7)
    lr = torch.rand(1, 3, *SR_LR, device=device, generator=gen)
    # at 2+ ranks the forward of a tile is captured once per tile shape (untimed, like building the model) and
    # replayed: the tiles are small enough there for ~1000 launches per forward to be bound by the host otherwise.
    # One rank takes the whole frame as one tile: its kernels are long, the host runs ahead of them, and the eager
    # launches are the faster form (64.1 ms against 65.9 ms replayed, tools/infer_eager_vs_graph.py)
    runner = GraphedForward(model) if world > 1 else model

    def run():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return tiled_super_resolve(runner, lr, 4, tile="balanced", halo=32, rank=rank, world=world, reduce=grouped)

    out = run()
    ok = bool(torch.isfinite(out).all())
    # one 256 x 256 tile of the frame through the path timed below (fused kernels; replayed from a graph at 2+ ranks) against the
    # same tile through the module-by-module path: separate projection and attention kernels, separate gate and
    # residual passes, stock convolutions at the two ends, DS-MSA reading its pass count on the host
    tile_err = None
    if rank == 0:
        from visiontransformer_amd import functional as K, models as M_
        from visiontransformer_amd.dswin import DeformableSparseWindowAttention
        patch = lr[:, :, 128:384, 320:576].contiguous()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            fused = runner(patch).float().clone()
            saved = (K.FUSE_PROJECTION, K.FUSE_GATED_SILU, M_.NHWC_ENDS, M_.DSMSA_BF16_INPUT)
            saved_fold = K.FOLD_RMSNORM
            sd = [m_ for m_ in model.modules() if isinstance(m_, DeformableSparseWindowAttention)]
            try:
                K.FUSE_PROJECTION = K.FUSE_GATED_SILU = M_.NHWC_ENDS = M_.DSMSA_BF16_INPUT = K.FOLD_RMSNORM = False
                for m_ in sd:
                    m_.static_inference = False
                plain = model(patch).float()
            finally:
                K.FUSE_PROJECTION, K.FUSE_GATED_SILU, M_.NHWC_ENDS, M_.DSMSA_BF16_INPUT = saved
                K.FOLD_RMSNORM = saved_fold
                for m_ in sd:
                    m_.static_inference = True
        tile_err = float((fused - plain).norm() / plain.norm())
        del fused, plain
    run()   # untimed: the tile check's other shapes have been through the caching allocator since the first frame
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    dt = (time.perf_counter() - t0) / iters
    if grouped:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    grid = balanced_grid(SR_LR[0], SR_LR[1], world)
    # network-level roofline: the frame padded to whole windows as one tile (the one-rank plan); tiles with halo compute more
    Hp, Wp = -(-SR_LR[0] // 16) * 16, -(-SR_LR[1] // 16) * 16
    flops = dswt_flops_per_frame(DSWT_DEFAULT, Hp, Wp)
    return {"metric": "x4 SR output megapixels/sec (fwd)", "value": out.shape[2] * out.shape[3] / 1e6 / dt, "unit": "MP/s",
            "n_gpus": world, "s_per_image": dt, "scaling": "strong", "dtype": "bf16", "data": "synthetic", "finite": ok,
            "tile_check": {"rel_err_fused_vs_module_by_module": tile_err,
                           "tile": "256x256 LR at (128, 320) of the frame; both in bf16, ReZero gates at 0.2 (zero at init), window decisions may differ between the two"},
            "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS, "flops_per_frame": flops,
                         "note": "multiply-add flops of one forward on the %dx%d padded frame (dswt_flops_per_frame) / time per frame" % (Hp, Wp)},
            "config": {"workload": "D: DSWT default (14.6 M params), 960x540 LR -> 3840x2160, %d tile(s) of %dx%d LR + 32 px halo "
                                   "clipped at the frame border, tiles round-robin over %d rank(s), one all-reduce of the image"
                                   % (len(plan_tiles(SR_LR[0], SR_LR[1], grid, 32, True)), grid[0], grid[1], world)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernels", action="store_true")
    ap.add_argument("--no-inference", action="store_true")
    ap.add_argument("--no-networks", action="store_true", help="skip the whole-network training steps (train_network)")
    ap.add_argument("--pmc-target", action="store_true",
                    help="launch every kernel of the path twice at the bench shapes and exit (the "
                         "program rocprofv3 --pmc passes run; see tools/pmc_summary.py)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_REHEARSAL=1 at N = 1: the N > 1 code path on one GPU - a one-rank RCCL process group, the pair under
    # DistributedDataParallel (gradient hooks, bucket rebuild after the first step, all-reduce launches on RCCL's
    # stream), the barriers, the tile all-reduce of the inference half.  Not a measurement: the line it prints says so.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1" and world == 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    import torch.distributed as dist
    grouped = world > 1 or rehearsal
    if grouped:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group(backend="nccl", device_id=device, rank=0, world_size=1)
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    from visiontransformer_amd import _lib
    _lib.load()
    if args.pmc_target:
        kernel_rooflines(device, iters=2, warm=0)
        torch.cuda.synchronize()
        return
    model = build_pair(device)
    if grouped:
        # 0.4 MB of gradients: one flat buffer, one all-reduce queued for the end of the backward pass
        # (BENCH_DDP=1: torch's DistributedDataParallel instead, gradients as views of its buckets, no broadcast of
        # the buffers - index tables fixed at construction)
        if os.environ.get("BENCH_DDP") == "1":
            model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], broadcast_buffers=False,
                                                              gradient_as_bucket_view=True)
        else:
            from visiontransformer_amd.parallel import FlatDataParallel
            model = FlatDataParallel(model)
    pair = model.module if grouped else model
    gen = torch.Generator(device=device).manual_seed(100 + rank)
    x = torch.randn(B_PER_GPU, HW, HW, C, device=device, generator=gen).requires_grad_()
    pair.sdwa.gumbel_noise = -torch.empty(B_PER_GPU, (HW // WIN) ** 2, device=device).exponential_(generator=gen).log()

    from visiontransformer_amd.parallel import GraphedTrainStep, graph_safe_sum

    def eager_step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = model(x)
        graph_safe_sum(y.float()).backward()
        model.zero_grad(set_to_none=True)
        x.grad = None

    # Process warm-up before the W warmup steps: code objects load on first launch, the caching allocator grows
    # to its steady size, and CPython runs its one full (generation-2) garbage collection of the freshly
    # imported torch heap about ten steps in - a 40-70 ms pause (tools/warm_probe.py) that, now that no step
    # waits on the GPU, would otherwise land inside a 20-step timed region and double its average.
    import gc
    for _ in range(15):
        eager_step()
    torch.cuda.synchronize()
    # The step as a training loop on this machine runs it: forward, loss and backward captured once as a HIP graph
    # and replayed (parallel.GraphedTrainStep; the path has no host read).  Issued launch by launch the step costs
    # the host as long as it costs the GPU (tools/host_step_probe.py).  The same kernels on the same data either
    # way; BENCH_EAGER=1 times the launch-by-launch form, which is also the fallback when the capture fails.
    step_mode = "hip graph replay (forward + loss + backward captured once)"
    step = eager_step
    if os.environ.get("BENCH_EAGER") == "1":
        step_mode = "eager (BENCH_EAGER=1)"
    elif grouped and os.environ.get("BENCH_DDP") == "1":
        step_mode = "eager (DistributedDataParallel's reducer works from autograd hooks)"
    else:
        try:
            # N > 1: the graph holds the local step; the one all-reduce of the flat gradient buffer follows the replay
            graphed = GraphedTrainStep(pair, lambda y: graph_safe_sum(y.float()))
            if grouped:
                def step():
                    graphed(x)
                    model.reduce_gradients()
                step_mode += ", then one RCCL all-reduce of the flat gradient buffer"
            else:
                step = lambda: graphed(x)
            step()
            torch.cuda.synchronize()
        except Exception as e:   # reported in the line; the eager step is the same work
            step_mode = "eager (graph capture failed: %s: %s)" % (type(e).__name__, str(e)[:200])
            step = eager_step
            model.zero_grad(set_to_none=True)
            x.grad = None
    gc.collect()
    gc.freeze()
    for _ in range(args.warmup):
        step()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    dt = time.perf_counter() - t0
    if grouped:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ls = pair.sdwa.last_selection
    n_pass = 0 if ls is None else (int(ls.n_dev.item()) if ls.n_dev is not None else ls.n)   # read after the timed region
    # second half of the headline metric, outside the timed training region: all ranks take part
    sr = None
    if not args.no_inference:
        try:
            sr = sr_inference(device, rank, world, grouped)
        except Exception as e:  # the training line above stays valid; the failure is reported in its place
            sr = {"metric": "x4 SR output megapixels/sec (fwd)", "value": None, "error": "%s: %s" % (type(e).__name__, e)}

    # whole-network training steps (the workloads a multi-GPU run is about: 58 MB of gradients all-reduced per step
    # for the DSWT default network), outside the timed region of the line's own value: all ranks take part
    tn = None
    if not args.no_networks:
        tn = train_network(device, rank, world, grouped)

    if rank == 0:
        ms = dt / args.steps * 1e3
        line = {
            "metric": "training patches/sec (DSWT WindowedMSA + DS-MSA attention pair, fwd+bwd)",
            "value": world * B_PER_GPU * args.steps / dt,
            "unit": "patches/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "step_mode": step_mode,
            "config": {"workload": "B: DSWT WindowedMSA+DS-MSA fwd+bwd, batch 32 x 64x64 LR patches, C=180, 6 heads, "
                                   "16x16 windows, bf16 autocast, k=4 windows/image (num_win_selected 0.25), "
                                   "%d of %d candidate windows passed the gate" % (n_pass, B_PER_GPU * 4),
                       "global_batch": world * B_PER_GPU, "parallelism": "dp%d" % world},
        }
        if rehearsal:
            line["rehearsal"] = "one-rank RCCL group: exercises the N > 1 code path (DDP, barriers, tile all-reduce), not a measurement"
        if not args.no_kernels:
            ks = kernel_rooflines(device)
            dom = ks["win_attn_bwd"]
            traffic = pmc_traffic(DOMINANT_KERNEL)
            line["roofline"] = {"bound": "hbm", "kernel": DOMINANT_KERNEL, "achieved": dom["achieved"],
                                "peak": dom["peak"], "unit": "GB/s", "frac": dom["frac"], "mfma_frac": dom["mfma_frac"],
                                "traffic": traffic,
                                "traffic_source": None if traffic is None else
                                "committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --pmc-target` "
                                "(profiles/pmc_traffic_latest.json), not measured in this run",
                                "us_per_launch": dom["us"]}
            line["kernels"] = ks
        if sr is not None:
            line["sr_inference"] = sr
        if tn is not None:
            line["train_network"] = tn
        if world == 1 and not rehearsal and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
