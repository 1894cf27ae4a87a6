#!/usr/bin/env python3
"""Two ranks of bench.py's training pair under DistributedDataParallel on ONE GPU with gloo collectives on device
tensors.  A diagnostic, not a check: on this image gloo's device-tensor path stalls with two processes on one GPU
(tools/ddp_plain_probe.py with HEAVY=1 stalls with stock modules in its first buffer broadcast), and here it stalls in
the second backward unless host timing is perturbed.  BENCH_REHEARSAL=1 python bench.py is the usable rehearsal.
  PAIR=wmsa|sdwa|both  which half of the pair runs
  TRACE=1              print every entry-point call of the second and later steps (TRACE_SYNC=0: record an event
                       per call instead of waiting for the GPU)
  WATCH=1              eight seconds into the second step, say which recorded events the GPU has finished
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/ddp_rehearsal.py"""
import faulthandler
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as dist

import bench


def say(msg):
    print("[rank %s] %6.1f s  %s" % (os.environ.get("RANK"), time.time() - T0, msg), flush=True)


T0 = time.time()
faulthandler.dump_traceback_later(25, repeat=False, file=sys.stderr, exit=True)
rank = int(os.environ["RANK"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="gloo")
say("process group up")
from visiontransformer_amd import _lib
_lib.load()
model = bench.build_pair(dev)
which = os.environ.get("PAIR", "both")   # wmsa / sdwa / both: which half of the pair runs
if which != "both":
    half = getattr(model, which)
    model = torch.nn.Sequential()
    model.add_module(which, half)
    if os.environ.get("STATIC") == "0" and hasattr(half, "static_training"):
        half.static_training = False
say("pair built")

TRACE = {"on": False}
EVENTS = []


def watchdog():
    """which of the launches recorded so far has the GPU finished, eight seconds into a stalled step"""
    import threading

    def look():
        time.sleep(float(os.environ.get("WATCH_AFTER", "8")))
        say("  watchdog: current stream idle: %s" % torch.cuda.current_stream().query())
        for i, (name, ev) in enumerate(list(EVENTS)):
            say("  event %3d %-28s %s" % (i, name, "done" if ev.query() else "PENDING"))
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            t = torch.ones(4, device=dev).add_(1)
        say("  watchdog: side stream launched")
        side.synchronize()
        say("  watchdog: side stream finished, %s" % t.tolist())
    threading.Thread(target=look, daemon=True).start()


class Tracer(dict):
    """stands in for the table of entry points: every call says its name, runs, waits for the GPU, says so"""

    def get(self, name, default=None):
        fn = getattr(_lib.load(), name)

        def traced(*args):
            if not TRACE["on"]:
                return fn(*args)
            say("  -> " + name)
            rc = fn(*args)
            if os.environ.get("TRACE_SYNC", "1") == "1":
                torch.cuda.synchronize()
            else:
                ev = torch.cuda.Event()
                ev.record()
                EVENTS.append((name, ev))
            say("  <- " + name)
            return rc
        return traced


if os.environ.get("WATCH") == "1" and os.environ.get("WATCH_HOOKS", "1") == "1":
    for pname, prm in model.named_parameters():
        def mark(t, pname=pname):
            if TRACE["on"]:
                ev = torch.cuda.Event()
                ev.record()
                EVENTS.append(("grad " + pname, ev))
        prm.register_post_accumulate_grad_hook(mark)
if os.environ.get("TRACE") == "1":
    _lib._entry = Tracer()
    for pname, prm in model.named_parameters():
        def ready(t, pname=pname):
            if TRACE["on"]:
                say("  grad ready: " + pname)
        prm.register_post_accumulate_grad_hook(ready)
ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
say("DDP wrapped")
gen = torch.Generator(device=dev).manual_seed(100 + rank)
x = torch.randn(bench.B_PER_GPU, bench.HW, bench.HW, bench.C, device=dev, generator=gen).requires_grad_()
for step in range(3):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = ddp(x)
    say("step %d forward done" % step)
    TRACE["on"] = step >= 1
    if step == 1 and (os.environ.get("TRACE") == "1" or os.environ.get("WATCH") == "1"):
        watchdog()
    y.float().sum().backward()
    torch.cuda.synchronize()
    say("step %d backward done" % step)
    g = torch.cat([p.grad.flatten() for p in model.parameters()])
    gs = [torch.zeros_like(g) for _ in range(dist.get_world_size())]
    dist.all_gather(gs, g)
    say("step %d gradients equal across ranks: %s" % (step, all(torch.equal(gs[0], t) for t in gs)))
    ddp.zero_grad(set_to_none=True)
    x.grad = None
dist.barrier()
say("done")
dist.destroy_process_group()
