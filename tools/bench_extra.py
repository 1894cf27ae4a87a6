#!/usr/bin/env python3
"""Measurements beyond the driver's bench line (one MI355X, synthetic data):

  D  DSWT default model, x4 SR of one 960x540 LR image (-> 3840x2160), eval mode, bf16 autocast:
     output megapixels / s with the balanced tile plan (one tile on one GPU), with tiles of
     256x256 LR + 32 px halo, and the busiest rank's share of the 2 / 4 / 8 rank plans.
  E  config B with num_win_selected 0.5 / 1.0 (256 / 512 windows of 16x16, gates open, random
     offsets): gather / scatter kernels against the HBM roofline.
  C  MSFT (dim 192, levels 2, w 8), x4 training step (fwd + bwd, bf16 autocast) on 64x64 patches,
     batch 8: patches / s.

  T  DSWT default network, x4 training step on 64x64 patches (T_BATCH, default 32): patches / s.
  M  parameter EMA of the DSWT default network: fused multi-tensor kernel against the per-parameter loop.

  python tools/bench_extra.py [D] [E] [C] [T] [M]   -> one JSON line per workload
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def sync_time(fn, iters, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def workload_d(dev):
    from visiontransformer_amd.models import DSWT_DEFAULT, dswt_from_config
    from visiontransformer_amd.parallel import balanced_grid, plan_tiles, tiled_super_resolve
    torch.manual_seed(0)
    model = dswt_from_config(DSWT_DEFAULT).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(0)
    lr = (torch.randn(1, 3, 540, 960, device=dev, generator=g) * 0.5).clamp_(-1, 1)

    def run(tile, rank=0, world=1, **kw):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return tiled_super_resolve(model, lr, 4, tile=tile, halo=32, rank=rank, world=world, reduce=False, **kw)

    out = run("balanced")
    assert out.shape == (1, 3, 2160, 3840) and torch.isfinite(out).all()
    mp = out.shape[2] * out.shape[3] / 1e6
    dt = sync_time(lambda: run("balanced"), 3)
    dt256 = sync_time(lambda: run(256), 3)
    # the busiest rank's share of the balanced plan for 2 / 4 / 8 ranks, timed on this one GPU: the
    # per-image time of an N-GPU run before its one final reduction
    shares = {}
    for w in (2, 4, 8):
        grid = balanced_grid(540, 960, w)
        shares[str(w)] = {"core": list(grid), "tiles": len(plan_tiles(540, 960, grid, 32, True)),
                          "rank0_s": sync_time(lambda: run("balanced", 0, w), 5, warm=3)}
    grid = balanced_grid(540, 960, 1)
    return {"metric": "x4 SR output megapixels/sec (fwd)", "value": mp / dt, "unit": "MP/s",
            "n_gpus": 1, "s_per_image": dt, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "D: DSWT default (14.6 M params), 960x540 LR -> 3840x2160, balanced plan: %d tile(s) of %dx%d LR, "
                                   "halo 32 clipped at the image border" % (len(plan_tiles(540, 960, grid, 32, True)), grid[0], grid[1])},
            "tiles_256_halo_32": {"tiles": len(plan_tiles(540, 960, 256, 32)), "s_per_image": dt256, "MP/s": mp / dt256},
            "rank0_share_by_world": shares}


def workload_e(dev):
    out = {}
    for frac, k in ((0.5, 8), (1.0, 16)):
        ks = bench.kernel_rooflines(dev, iters=20, warm=3, k=k, attention=False)
        out["num_win_selected=%.1f (%d windows)" % (frac, bench.B_PER_GPU * k)] = {
            n: {"us": round(v["us"], 1), "GB/s": round(v.get("achieved", 0.0), 1), "frac_of_8TB/s": round(v.get("frac", 0.0), 3)}
            for n, v in ks.items()}
    return {"metric": "gather / scatter kernels vs HBM roofline", "config": {"workload": "E: B=32, 64x64, C=180, w=16, gates open"},
            "kernels": out}


def workload_c(dev):
    from visiontransformer_amd.models import MSFT_CONFIG_C, MSFTransformer
    torch.manual_seed(0)
    cfg = dict(MSFT_CONFIG_C)
    model = MSFTransformer(**cfg).to(dev).train()
    for m in model.modules():
        if hasattr(m, "attn_drop"):
            m.attn_drop = 0.0
    B = int(os.environ.get("C_BATCH", "32"))
    x = torch.randn(B, 3, 64, 64, device=dev)
    target = torch.randn(B, 3, 256, 256, device=dev)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = model(x)
        # an L1 loss as in training: a dense gradient (the gradient of a plain mean() is a stride-0
        # expansion, which sends MIOpen to its naive kernels)
        torch.nn.functional.l1_loss(y.float(), target).backward()
        model.zero_grad(set_to_none=True)

    dt = sync_time(step, 5, warm=2)
    return {"metric": "training patches/sec (MSFT x4, fwd+bwd)", "value": B / dt, "unit": "patches/s", "n_gpus": 1,
            "ms_per_step": dt * 1e3, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "C: MSFT dim 192, levels 2, w 8, 8 blocks x 3, batch %d of 64x64 LR" % B}}


def workload_t(dev):
    """DSWT default network, x4 training step (fwd + bwd, bf16 autocast, L1 loss) on 64x64 LR patches"""
    from visiontransformer_amd.models import DSWT_DEFAULT, DSWTransformer
    torch.manual_seed(0)
    model = DSWTransformer(**DSWT_DEFAULT).to(dev).train()
    B = int(os.environ.get("T_BATCH", "32"))
    x = torch.rand(B, 3, 64, 64, device=dev)
    target = torch.rand(B, 3, 256, 256, device=dev)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = model(x)
        torch.nn.functional.l1_loss(y.float(), target).backward()
        model.zero_grad(set_to_none=True)

    import gc
    from visiontransformer_amd.dswin import DeformableSparseWindowAttention
    sdwa = [m for m in model.modules() if isinstance(m, DeformableSparseWindowAttention)]
    sync_time(step, 8, warm=4)
    gc.collect()
    gc.freeze()
    dt = sync_time(step, 10, warm=2)
    # share of the B*k window slots whose gate is open (the device-side count; read here, after the timing)
    live = sum(int(m.last_selection.n_dev) for m in sdwa) / max(1, sum(m.last_selection.n for m in sdwa))
    # the variant that reads the pass count on the host and processes only the passing windows
    for m in sdwa:
        m.static_training = False
    dt_host = sync_time(step, 10, warm=3)
    return {"metric": "training patches/sec (DSWT x4, fwd+bwd)", "value": B / dt, "unit": "patches/s", "n_gpus": 1,
            "ms_per_step": dt * 1e3, "dtype": "bf16", "data": "synthetic",
            "open_gate_share": live, "ms_per_step_with_host_read_of_the_pass_count": dt_host * 1e3,
            "config": {"workload": "T: DSWT default (8 blocks x 3, dim 180, w 16), batch %d of 64x64 LR" % B}}


def workload_m(dev):
    """parameter EMA of the DSWT default network (14.6 M fp32 parameters in ~600 tensors): the fused
    multi-tensor kernel against the reference's per-parameter loop (toolkit/param_ema.py:9-17)"""
    import copy
    from visiontransformer_amd.ema import ParameterEMA
    from visiontransformer_amd.models import DSWT_DEFAULT, DSWTransformer
    torch.manual_seed(0)
    model = DSWTransformer(**DSWT_DEFAULT).to(dev)
    ema_model = copy.deepcopy(model).eval().requires_grad_(False)
    helper = ParameterEMA(start_step=0)
    n_t = len(list(model.parameters()))
    numel = sum(p.numel() for p in model.parameters())

    def loop():   # the reference's expression, parameter by parameter
        for c, e in zip(model.parameters(), ema_model.parameters()):
            e.data = e.data * helper.beta + (1 - helper.beta) * c.data

    t_fused = sync_time(lambda: helper.update_model_average(ema_model, model), 20, warm=3)
    t_loop = sync_time(loop, 5, warm=2)
    # device time of the fused launches alone (the host builds three pointer tables per call)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(10):
        helper.update_model_average(ema_model, model)
    ev1.record()
    torch.cuda.synchronize()
    gbs = 3 * numel * 4 / 1e9
    # device time of the kernel launches of one call (the call itself is bound by the host's table building)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            helper.update_model_average(ema_model, model)
        torch.cuda.synchronize()
    dev_us = sum(e.self_device_time_total for e in prof.key_averages() if "ema_update_kernel" in e.key) / 5
    launches = sum(e.count for e in prof.key_averages() if "ema_update_kernel" in e.key) // 5
    return {"metric": "parameter EMA step", "tensors": n_t, "parameters": numel, "fused_ms": t_fused * 1e3, "reference_loop_ms": t_loop * 1e3,
            "speedup": t_loop / t_fused, "algorithmic_GB": gbs, "GB/s_wall": gbs / t_fused,
            "config": {"workload": "M: EMA of DSWT default, fp32, beta 0.995"}, "wall_ms_per_call_back_to_back": ev0.elapsed_time(ev1) / 10,
            "kernel_launches_per_call": launches, "kernel_us_per_call": dev_us,
            "roofline": {"bound": "hbm", "achieved": gbs / (dev_us * 1e-6) if dev_us else None, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / (dev_us * 1e-6) / bench.HBM_PEAK_GBS if dev_us else None}}


def main():
    dev = torch.device("cuda:0")
    from visiontransformer_amd import _lib
    _lib.load()
    which = [a.upper() for a in sys.argv[1:]] or ["D", "E", "C"]
    for w in which:
        print(json.dumps({"D": workload_d, "E": workload_e, "C": workload_c, "T": workload_t, "M": workload_m}[w](dev)), flush=True)


if __name__ == "__main__":
    main()
