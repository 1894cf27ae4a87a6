"""Multi-GPU use of the path on one 8 x MI355X node: one process per GPU, torch.distributed.

The path shards in exactly two ways, both with no collective inside the data path:
  * training - data parallel: independent samples per rank, one gradient all-reduce per step
    (DistributedDataParallel; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU);
    the reference wires gloo + DDP at toolkit/init.py:111,131;
  * large-image inference - independent spatial tiles with a halo, dealt round-robin to the
    ranks; every rank writes its own output regions and one final reduction assembles the image
    (the reference has no tiling: toolkit/model_helper2.py:167-191 runs whole images).
Tile-local results differ from whole-image inference by design (DS-MSA's top-k is global per
image and the receptive field grows with depth), so parity is defined per tile.
"""
import os
from dataclasses import dataclass
from typing import List

import torch
import torch.distributed as dist
import torch.nn.functional as F


def init_distributed(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*.
    Returns (rank, world, device).  Single process: no group is created."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if use_cuda else "gloo")
        if backend == "nccl":
            dist.init_process_group(backend=backend, device_id=device)
        else:
            dist.init_process_group(backend=backend)
    return rank, world, device


def data_parallel(model, device, find_unused_parameters=False, broadcast_buffers=True, flat=False):
    """Wrap for data-parallel training when a process group exists (gradient all-reduce in buckets,
    overlapped with backward).  The kernel-backed modules give every parameter a gradient on every
    step: DS-MSA's default path (static_training) never returns early - when no window passes the gate
    its attention parameters receive zeros - so the reducer needs no search for unused parameters
    (the reference's DDP call, toolkit/init.py:131, does not ask for one either).  Only a map with fewer
    than 1 / num_win_selected windows (k == 0, DSwinT_arch.py:273) or static_training = False leaves
    parameters without a gradient: pass find_unused_parameters=True for those.  Gradients live in the reducer's
    buckets (gradient_as_bucket_view): no copy into and out of them around the all-reduce - 0.24 ms of a 1.8 ms
    step of the attention pair.  broadcast_buffers=False skips the per-forward broadcast of the buffers when
    they are the index tables fixed at construction (the whole DSWT also counts its iterations in a buffer:
    the same on every rank when the ranks step together).  flat=True: FlatDataParallel, for modules whose gradients
    fit one short all-reduce."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    if flat:
        return FlatDataParallel(model)
    ids = [device.index] if device.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, find_unused_parameters=find_unused_parameters,
                                                     broadcast_buffers=broadcast_buffers, gradient_as_bucket_view=True)


class FlatDataParallel(torch.nn.Module):
    """Data-parallel wrapper for a module whose gradients fit one collective (the attention pair: 0.4 MB).

    Every gradient lives in one flat buffer per dtype.  The module's output carries a hook that, when the backward
    pass reaches it, queues one callback for the end of that pass; the callback gathers the gradients that are not
    in the buffer yet with ONE multi-tensor copy, runs ONE all-reduce (average) over the buffer and leaves each
    parameter's .grad as a view of it.  Against DistributedDataParallel's reducer - a hook per parameter on the
    autograd thread, a copy per parameter into the bucket after zero_grad(set_to_none=True) - that is three
    launches per step instead of two per parameter; nothing overlaps with the backward pass, which is the right
    trade only while the all-reduce is short (use data_parallel(..., flat=False) for the whole networks).
    A parameter without a gradient in some pass contributes zeros, so early returns need no search for unused
    parameters.  The module must return one tensor.  `module` keeps the reference's attribute name for the
    wrapped model (toolkit/init.py:131 wraps with DistributedDataParallel)."""

    def __init__(self, module, process_group=None):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self._sync = True
        self._armed_task = None   # id of the backward pass (graph task) whose end-of-pass callback is queued
        params = [p for p in module.parameters() if p.requires_grad]
        with torch.no_grad():       # same start on every rank, as DistributedDataParallel's constructor does
            for t in list(module.parameters()) + list(module.buffers()):
                if t.is_contiguous():
                    dist.broadcast(t, 0, group=process_group)
                else:                  # e.g. channels-last convolution weights: collectives take dense tensors
                    dense = t.contiguous()
                    dist.broadcast(dense, 0, group=process_group)
                    t.copy_(dense)
        self._sets = []            # per (dtype, device): (flat buffer, parameters, views)
        by_kind = {}
        for p in params:
            by_kind.setdefault((p.dtype, p.device), []).append(p)
        for (dtype, device), ps in by_kind.items():
            # every view starts on a 128-byte boundary of the buffer
            step = max(1, 128 // torch.empty((), dtype=dtype).element_size())
            offs, total = [], 0
            for p in ps:
                offs.append(total)
                total += -(-p.numel() // step) * step
            flat = torch.zeros(total, dtype=dtype, device=device)
            views = [flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, ps)]
            self._sets.append((flat, ps, views))
        backend = dist.get_backend(process_group)
        self._avg = dist.ReduceOp.AVG if backend == "nccl" else None

    def no_sync(self):
        """context: backward passes inside accumulate locally (gradient accumulation), no all-reduce"""
        outer = self

        class _NoSync:
            def __enter__(self):
                self.old, outer._sync = outer._sync, False

            def __exit__(self, *exc):
                outer._sync = self.old
        return _NoSync()

    def forward(self, *args, **kwargs):
        out = self.module(*args, **kwargs)
        if torch.is_grad_enabled() and self._sync:
            if not isinstance(out, torch.Tensor):
                raise TypeError("FlatDataParallel needs a module that returns one tensor, got %s" % type(out).__name__)
            if out.requires_grad:
                out.register_hook(self._arm)
        return out

    def _arm(self, grad):
        # One callback per backward pass.  The state is the id of the pass, not a flag: a pass that raises after this
        # hook has fired never runs its callback, and a flag left set by it would keep every later pass from
        # queueing one (gradients no longer averaged, no error).
        task = torch._C._current_graph_task_id()
        if self._armed_task != task:
            self._armed_task = task
            torch.autograd.Variable._execution_engine.queue_callback(self._reduce)
        return None

    def reduce_gradients(self):
        """the all-reduce of a step whose backward pass did not run under this wrapper's hook: after a replay of
        GraphedTrainStep(self.module, ...), whose captured backward leaves the local gradients in p.grad"""
        self._reduce()

    @torch.no_grad()
    def _reduce(self):
        self._armed_task = None
        for flat, ps, views in self._sets:
            dst, src = [], []
            for p, v in zip(ps, views):
                g = p.grad
                if g is None:
                    v.zero_()
                elif g.data_ptr() != v.data_ptr() or g.stride() != v.stride():
                    dst.append(v)
                    src.append(g)
            if dst:
                torch._foreach_copy_(dst, src)
            if self._avg is not None:
                dist.all_reduce(flat, op=self._avg, group=self.group)
            else:
                dist.all_reduce(flat, group=self.group)
                flat.div_(self.world)
            for p, v in zip(ps, views):
                p.grad = v


@dataclass
class Tile:
    y0: int  # core region in the LR image, [y0, y1) x [x0, x1)
    y1: int
    x0: int
    x1: int
    py0: int  # padded region (core + halo), may reach outside the image: reflect-padded
    py1: int
    px0: int
    px1: int


def plan_tiles(H, W, tile=256, halo=32, clip=False) -> List[Tile]:
    """Row-major list of tiles covering an H x W image; every core is at most tile x tile
    (tile: an int or (rows, columns)).  clip=True keeps the halo inside the image, so a tile at the
    border sees the border as whole-image inference does instead of a reflected halo."""
    th, tw = (tile, tile) if isinstance(tile, int) else tile
    tiles = []
    for y0 in range(0, H, th):
        for x0 in range(0, W, tw):
            y1, x1 = min(y0 + th, H), min(x0 + tw, W)
            py0, py1, px0, px1 = y0 - halo, y1 + halo, x0 - halo, x1 + halo
            if clip:
                py0, py1, px0, px1 = max(py0, 0), min(py1, H), max(px0, 0), min(px1, W)
            tiles.append(Tile(y0, y1, x0, x1, py0, py1, px0, px1))
    return tiles


def balanced_grid(H, W, world=1, max_side=1024, halo=32, align=16):
    """Core size (rows, columns) of the cheapest tiling of an H x W image whose tile count is a
    multiple of `world` (every rank gets the same number of tiles) with padded tiles (core + halo
    clipped at the image border) of at most max_side a side.  Cost = padded pixels computed by the
    busiest rank; cores are multiples of `align` (the window size) so that only the last row and
    column of tiles need the model's own padding.  world = 1 and a small enough image give a
    single tile: whole-image inference, as toolkit/model_helper2.py:167-191 runs it."""
    up = lambda v: -(-v // align) * align
    best = None
    for ny in range(1, -(-H // align) + 1):
        th = up(-(-H // ny))
        if -(-H // th) != ny:
            continue
        for nx in range(1, -(-W // align) + 1):
            tw = up(-(-W // nx))
            if -(-W // tw) != nx or (ny * nx) % world:
                continue
            tiles = plan_tiles(H, W, (th, tw), halo, clip=True)
            if max(max(t.py1 - t.py0, t.px1 - t.px0) for t in tiles) > max_side:
                continue
            load = [0] * world
            for i, t in enumerate(tiles):
                load[i % world] += up(t.py1 - t.py0) * up(t.px1 - t.px0)
            cost = (max(load), ny * nx)
            if best is None or cost < best[0]:
                best = (cost, (th, tw))
    if best is None:
        raise ValueError("no tiling of %dx%d into a multiple of %d tiles of at most %d a side" % (H, W, world, max_side))
    return best[1]


def _reflect_crop(img, py0, py1, px0, px1):
    """img[:, :, py0:py1, px0:px1] where the window may stick out of the image: outside pixels
    are reflected (like the reference's padding(), DSwinT_arch.py:463)."""
    _, _, H, W = img.shape
    top, left = max(0, -py0), max(0, -px0)
    bot, right = max(0, py1 - H), max(0, px1 - W)
    core = img[:, :, max(py0, 0):min(py1, H), max(px0, 0):min(px1, W)]
    if top or left or bot or right:
        # reflect needs pad < size; tiny borders are padded in several steps
        while top or left or bot or right:
            h, w = core.shape[-2:]
            t, b_, l, r = min(top, h - 1), min(bot, h - 1), min(left, w - 1), min(right, w - 1)
            core = F.pad(core, (l, r, t, b_), mode="reflect")
            top, bot, left, right = top - t, bot - b_, left - l, right - r
    return core


class GraphedForward:
    """The no-grad forward of `model` replayed from HIP graphs, one graph per input shape.

    A DSWT forward is about a thousand launches; below some 80 k LR pixels per tile the host cannot issue them as
    fast as the GPU retires them.  The inference path has no host read (DS-MSA keeps its pass count on the
    device), so the whole forward can be captured once per tile shape and replayed.  Replays read the
    parameters where they live - in-place updates are seen - and the prepared bf16 operands are rebuilt inside
    the graph (functional._may_keep_operands).  The returned tensor is the graph's own output buffer: consume it
    before the next call with the same shape.  Modules that draw host-side randomness per call (attention
    dropout seeds) would replay the captured draw; DSWT inference has none."""

    def __init__(self, model, max_graphs=8):
        self.model = model
        self.max_graphs = max_graphs
        self._graphs = {}

    @torch.no_grad()
    def __call__(self, x):
        if not x.is_cuda:
            return self.model(x)
        key = (tuple(x.shape), x.dtype, x.device, torch.is_autocast_enabled("cuda"),
               torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None)
        entry = self._graphs.get(key)
        if entry is None:
            if len(self._graphs) >= self.max_graphs:
                return self.model(x)
            static_x = x.clone()
            for _ in range(2):     # eager warm-up: library algorithm choices, allocator growth, code loading
                self.model(static_x)
            torch.cuda.synchronize(x.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_y = self.model(static_x)
            entry = self._graphs[key] = (graph, static_x, static_y)
        graph, static_x, static_y = entry
        static_x.copy_(x)
        graph.replay()
        return static_y


class _BlockwiseSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, block):
        ctx.shape = t.shape
        t = t.detach().reshape(-1)
        while t.numel() > 4 * block:
            n = t.numel()
            m = n // block * block
            head = t.view(-1, block).sum(1) if m == n else t[:m].view(-1, block).sum(1)
            t = head if m == n else torch.cat([head, t[m:]])
        return t.sum()

    @staticmethod
    def backward(ctx, g):
        return g.expand(ctx.shape), None     # as torch.sum's backward: a broadcast view, no pass over memory


def graph_safe_sum(t, block=4096):
    """Sum of all elements of t by reductions that each fit one workgroup (rows of `block` elements, then the row
    sums the same way), differentiable (the gradient is the broadcast of the incoming one, as for torch.sum).  For
    losses inside GraphedTrainStep: torch's reduction of a large tensor to one value runs over several workgroups
    behind a memset of its arrival counters, and in a replayed graph on ROCm 7.2 that memset node was seen to lose its
    place among the kernel nodes when the host had synchronised before the replay - the value came out wrong (a
    partial sum) while every gradient stayed right."""
    return _BlockwiseSumFn.apply(t, block)


def graph_safe_mean(t, block=4096):
    """mean of all elements of t, see graph_safe_sum"""
    return graph_safe_sum(t, block) / t.numel()


class GraphedTrainStep:
    """Forward, loss and backward of `model` captured once per input shape as a HIP graph and replayed.

    The attention pair's training step is about sixty launches of 5-200 us each, and the Python and autograd work
    that issues them (about 1.4 ms) is as long as the GPU's share: replayed from a graph the step costs the host one
    call.  What makes the capture possible is that the training path has no host read (DS-MSA keeps its pass count
    on the device, every window slot is computed or zeroed by the kernels) and that every kernel is deterministic.

    step = GraphedTrainStep(model, loss_fn); loss = step(x, *loss_args) leaves the gradients in p.grad - the
    graph's own buffers, OVERWRITTEN by the next call (not accumulated: do not zero them, and take optimizer steps
    between calls as usual; they read p.grad and update the parameters in place, which the replays see).  x and
    loss_args are copied into the graph's input buffers unless they ARE those buffers (the tensors passed at
    capture are adopted), so a loader that fills the same tensors every step pays no copy; x.grad is produced when
    x requires it.  A `model` wrapped in FlatDataParallel has its one all-reduce inside the graph.  Host-side
    randomness drawn per call (attention dropout seeds) is replayed as captured; the Gumbel noise of DS-MSA is a
    tensor and can be refilled in place between calls.  Reduce the loss to its scalar with graph_safe_sum /
    graph_safe_mean (see there) rather than with a plain .sum() / .mean() of a large tensor.  One graph per (shapes, dtypes, autocast) key, at most
    max_graphs; beyond that, and on CPU tensors, the step runs eagerly."""

    def __init__(self, model, loss_fn, autocast_dtype=torch.bfloat16, warmup=3, max_graphs=4):
        self.model = model
        self.loss_fn = loss_fn
        self.autocast_dtype = autocast_dtype
        self.warmup = warmup
        self.max_graphs = max_graphs
        self._graphs = {}

    def _eager(self, x, *loss_args):
        if self.autocast_dtype is not None and x.is_cuda:
            with torch.autocast("cuda", dtype=self.autocast_dtype):
                y = self.model(x)
        else:
            y = self.model(x)
        loss = self.loss_fn(y, *loss_args)
        loss.backward()
        return loss.detach()

    def _params(self):
        return [p for p in self.model.parameters() if p.requires_grad]

    def __call__(self, x, *loss_args):
        if not x.is_cuda:
            return self._eager(x, *loss_args)
        key = (tuple(x.shape), x.dtype, x.device, x.requires_grad, self.autocast_dtype,
               tuple((tuple(a.shape), a.dtype) for a in loss_args))
        entry = self._graphs.get(key)
        if entry is None:
            if len(self._graphs) >= self.max_graphs:
                return self._eager(x, *loss_args)
            entry = self._graphs[key] = self._capture(x, loss_args)
        graph, static_x, static_args, static_loss, grads, x_grad = entry
        if x.data_ptr() != static_x.data_ptr():
            static_x.copy_(x)
        for a, sa in zip(loss_args, static_args):
            if a.data_ptr() != sa.data_ptr():
                sa.copy_(a)
        graph.replay()
        for p, g in grads:
            if p.grad is not g:
                p.grad = g
        if x_grad is not None:
            x.grad = x_grad
        return static_loss

    def _capture(self, x, loss_args):
        params = self._params()
        # eager warm-up on a side stream (code objects, allocator growth, library set-up), as graph capture asks
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                for p in params:
                    p.grad = None
                x.grad = None
                self._eager(x, *loss_args)
        torch.cuda.current_stream(x.device).wait_stream(side)
        torch.cuda.synchronize(x.device)
        for p in params:
            p.grad = None      # the captured backward then CREATES the gradients (in the graph's pool): no accumulate
        x.grad = None
        graph = torch.cuda.CUDAGraph()
        from . import functional as K
        # the operands of the projections are rebuilt inside the graph from the live parameters: one launch for all of them
        with torch.cuda.graph(graph), K.weight_scope(self.model):
            static_loss = self._eager(x, *loss_args)
        grads = [(p, p.grad) for p in params if p.grad is not None]
        return graph, x, tuple(loss_args), static_loss, grads, x.grad


@torch.no_grad()
def tiled_super_resolve(model, lr, scale, tile=256, halo=32, rank=0, world=1, reduce=True, clip=False):
    """x`scale` super-resolution of lr [B,3,H,W] by tiles with halo.  Rank r computes tiles
    r, r + world, ...; with reduce=True the disjoint output regions are summed over the group so
    every rank returns the whole image (a single collective, after all compute).
    tile="balanced" takes the tiling of balanced_grid for this image and world size (halo clipped
    at the image border); otherwise tile / clip are passed to plan_tiles."""
    B, C, H, W = lr.shape
    out = torch.zeros(B, C, H * scale, W * scale, dtype=lr.dtype, device=lr.device)
    if tile == "balanced":
        tile, clip = balanced_grid(H, W, world, halo=halo), True
    for i, t in enumerate(plan_tiles(H, W, tile, halo, clip)):
        if i % world != rank:
            continue
        patch = _reflect_crop(lr, t.py0, t.py1, t.px0, t.px1)
        sr = model(patch)
        oy, ox = (t.y0 - t.py0) * scale, (t.x0 - t.px0) * scale
        # one pass from the network's output (bf16 under autocast, channels-last when the NHWC tail produced it) into
        # the image: dtype, layout and crop in the same copy
        out[:, :, t.y0 * scale:t.y1 * scale, t.x0 * scale:t.x1 * scale].copy_(
            sr[:, :, oy:oy + (t.y1 - t.y0) * scale, ox:ox + (t.x1 - t.x0) * scale])
    if reduce and world > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out
