"""torch.autograd Functions over the C ABI of libdswin_hip.so.

Each Function allocates its outputs with torch (so memory stays in the caching allocator),
passes raw device pointers and the current HIP stream to the library, and saves for backward
exactly what the backward kernels read.  Nothing here computes on the CPU.
"""
import contextlib
import ctypes
import os
import weakref
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import call, dtype_code, ptr, stream

_c_int = ctypes.c_int
_c_float = ctypes.c_float


class _NoGradCtx:
    """Stand-in for the autograd context when gradients are disabled: nothing is saved, nothing needs a
    gradient.  Calling Function.forward with it skips the dispatch cost of Function.apply, which at a few
    hundred small launches per inference forward is a measurable part of the host time."""

    class _Never:
        def __getitem__(self, i):
            return (False,) * 16 if isinstance(i, slice) else False

        def __iter__(self):
            return iter(())

    needs_input_grad = _Never()

    def __setattr__(self, name, value):
        pass

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, flag):
        pass


_NO_GRAD = _NoGradCtx()
_NULL_CTX = contextlib.nullcontext()


_FP16_WARNED = [False]


def autocast_dtype(fallback):
    """The 16-bit type the kernels compute in under the caller's CUDA autocast, else `fallback`.

    The kernels are built for fp32 and bf16.  Under fp16 autocast (torch's default for autocast("cuda"), the
    mode of the reference's trainer, toolkit/model_helper2.py:35,120) the modules compute in bf16 instead -
    the same storage width with fp32's exponent range, so GradScaler's scaled losses and gradients pass
    through unharmed - and say so once."""
    if not torch.is_autocast_enabled("cuda"):
        return fallback
    act = torch.get_autocast_dtype("cuda")
    if act == torch.float16:
        if not _FP16_WARNED[0]:
            _FP16_WARNED[0] = True
            import warnings
            warnings.warn("visiontransformer_amd: fp16 autocast requested; the HIP kernels are built for bf16 and "
                          "fp32, so the attention modules compute in bf16 under this autocast (use "
                          "torch.autocast('cuda', dtype=torch.bfloat16) to make that explicit)", stacklevel=3)
        return torch.bfloat16
    return act


def supported_input(x):
    """fp16 tensors (what stock ops hand over under fp16 autocast) widened to fp32 - lossless, and a type the
    kernels read; everything else as it is"""
    return x.float() if x.dtype == torch.float16 else x


def _run(fn, *args):
    """fn.apply(*args); without autograd, fn.forward directly"""
    if torch.is_grad_enabled():
        return fn.apply(*args)
    return fn.forward(_NO_GRAD, *args)


@dataclass
class Selection:
    """Outcome of the window selection for one DS-MSA call (all tensors on the GPU).

    n            number of windows that passed the hard gate (host int)
    win_b/cell/h1/w1  int32 [n]: image, source grid cell (flat), top-left corner
    cellmap      int32 [B,Hw,Ww]: window id started from that cell, or -1
    scores_delta, gates_delta  fp32 [n] straight-through values (differentiable)
    cand_idx, cand_pass  int32 [B,k]: top-k cells by descending probability, gate decision
    probs        fp32 [B,Hw*Ww]
    """
    n: int
    win_b: torch.Tensor
    win_cell: torch.Tensor
    win_h1: torch.Tensor
    win_w1: torch.Tensor
    cellmap: torch.Tensor
    scores_delta: torch.Tensor
    gates_delta: torch.Tensor
    cand_idx: torch.Tensor
    cand_pass: torch.Tensor
    probs: torch.Tensor
    n_dev: torch.Tensor = None   # static selection only: the number of passing windows, int32 [1] on the device

    def detached(self):
        """the same record without autograd history: what a module may keep as an attribute (a tensor with a
        grad_fn there would pin the graph of the last forward and make copy.deepcopy of the module fail)"""
        return Selection(**{k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in self.__dict__.items()})


class _SelectFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sd, sig, noise, tau, k, w, H, W):
        _lib.require_cuda(sd, sig, noise)
        B, four, Hw, Ww = sd.shape
        assert four == 4 and (sig is None or (sig.shape == (B, 3, Hw, Ww) and sig.dtype == sd.dtype))
        sd = sd.contiguous()
        sig = None if sig is None else sig.contiguous()
        n = Hw * Ww
        dev = sd.device
        if noise is not None:
            noise = noise.to(torch.float32).contiguous()
            assert noise.shape == (B, n)
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        probs = torch.empty(B, n, **f32)
        cand_idx = torch.empty(B, k, **i32)
        cand_p = torch.empty(B, k, **f32)
        cand_pass = torch.empty(B, k, **i32)
        n_pass = torch.empty(1, **i32)
        win = torch.empty(4, B * k, **i32)
        deltas = torch.empty(2, B * k, **f32)
        cellmap = torch.empty(B, Hw, Ww, **i32)
        call("dsw_select_fwd", ptr(sd), ptr(sig), dtype_code(sd.dtype), ptr(noise), _c_float(tau),
             B, Hw, Ww, k, w, H, W, ptr(probs), ptr(cand_idx), ptr(cand_p), ptr(cand_pass), ptr(n_pass),
             ptr(win[0]), ptr(win[1]), ptr(win[2]), ptr(win[3]), ptr(deltas[0]), ptr(deltas[1]),
             ptr(cellmap), stream(dev))
        npass = int(n_pass.item())  # the one host sync of a DS-MSA call (the reference has two)
        win_b, win_cell, win_h1, win_w1 = (win[i, :npass].contiguous() for i in range(4))
        scores_delta = deltas[0, :npass].contiguous()
        gates_delta = deltas[1, :npass].contiguous()
        ctx.save_for_backward(probs, win_b, win_cell, sd if sig is None else None)
        ctx.meta = (tau, B, Hw, Ww, npass, sd.dtype)
        ctx.mark_non_differentiable(win_b, win_cell, win_h1, win_w1, cellmap, cand_idx, cand_pass, probs)
        ctx.set_materialize_grads(False)
        return scores_delta, gates_delta, win_b, win_cell, win_h1, win_w1, cellmap, cand_idx, cand_pass, probs

    @staticmethod
    def backward(ctx, d_scores_delta, d_gates_delta, *unused):
        probs, win_b, win_cell, sd = ctx.saved_tensors
        tau, B, Hw, Ww, npass, dt = ctx.meta
        dev = probs.device
        n = Hw * Ww
        zero = lambda: torch.zeros(npass, dtype=torch.float32, device=dev)
        dsd = zero() if d_scores_delta is None else d_scores_delta.to(torch.float32).contiguous()
        dgd = zero() if d_gates_delta is None else d_gates_delta.to(torch.float32).contiguous()
        if sd is not None:  # the sigmoids were taken inside the forward kernels: one kernel, gradient of sd itself
            d_sd = torch.empty(B, 4, Hw, Ww, dtype=dt, device=dev)
            call("dsw_select_bwd_sd", ptr(probs), ptr(dsd), ptr(dgd), ptr(win_b), ptr(win_cell), npass, _c_float(tau),
                 B, Hw, Ww, ptr(sd), dtype_code(dt), ptr(d_sd), None, stream(dev))
            return d_sd, None, None, None, None, None, None, None
        grads = torch.empty(2, B, n, dtype=torch.float32, device=dev)
        call("dsw_select_bwd", ptr(probs), ptr(dsd), ptr(dgd), ptr(win_b), ptr(win_cell), npass, _c_float(tau),
             B, Hw, Ww, ptr(grads[0]), ptr(grads[1]), None, stream(dev))
        d_sd = torch.zeros(B, 4, Hw, Ww, dtype=dt, device=dev)
        d_sd[:, 0] = grads[0].view(B, Hw, Ww).to(dt)
        d_sig = torch.zeros(B, 3, Hw, Ww, dtype=dt, device=dev)  # offsets (channels 1, 2) get no gradient
        d_sig[:, 0] = grads[1].view(B, Hw, Ww).to(dt)
        return d_sd, d_sig, None, None, None, None, None, None


def select_windows(sd, sig, noise, tau, k, w, H, W):
    """Top-k + hard gate + compaction + corners (DSwinT_arch.py:201-237, :280-295).  sig =
    sd[:, 1:].sigmoid(), or None to have the kernels take the sigmoids themselves."""
    out = _run(_SelectFn, sd, sig, noise, float(tau), int(k), int(w), int(H), int(W))
    sdl, gdl, win_b, win_cell, win_h1, win_w1, cellmap, cand_idx, cand_pass, probs = out
    return Selection(n=win_b.numel(), win_b=win_b, win_cell=win_cell, win_h1=win_h1, win_w1=win_w1,
                     cellmap=cellmap, scores_delta=sdl, gates_delta=gdl, cand_idx=cand_idx,
                     cand_pass=cand_pass, probs=probs)


class _SelectStaticFn(torch.autograd.Function):
    """dsw_select_fwd on all B * k window slots, the pass count staying on the device (see select_windows_static)"""

    @staticmethod
    def forward(ctx, sd, noise, tau, k, w, H, W):
        _lib.require_cuda(sd, noise)
        B, four, Hw, Ww = sd.shape
        assert four == 4
        sd = sd.contiguous()
        n = Hw * Ww
        dev = sd.device
        if noise is not None:
            noise = noise.to(torch.float32).contiguous()
            assert noise.shape == (B, n)
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        probs = torch.empty(B, n, **f32)
        cand_idx = torch.empty(B, k, **i32)
        cand_p = torch.empty(B, k, **f32)
        cand_pass = torch.empty(B, k, **i32)
        n_pass = torch.empty(1, **i32)
        win = torch.empty(4, B * k, **i32)       # the kernel zeroes the slots behind the passing windows
        deltas = torch.empty(2, B * k, **f32)
        cellmap = torch.empty(B, Hw, Ww, **i32)
        call("dsw_select_fwd", ptr(sd), None, dtype_code(sd.dtype), ptr(noise), _c_float(tau), B, Hw, Ww, k, w, H, W, ptr(probs),
             ptr(cand_idx), ptr(cand_p), ptr(cand_pass), ptr(n_pass), ptr(win[0]), ptr(win[1]), ptr(win[2]), ptr(win[3]),
             ptr(deltas[0]), ptr(deltas[1]), ptr(cellmap), stream(dev))
        ctx.save_for_backward(probs, win, sd, n_pass)
        ctx.meta = (tau, B, Hw, Ww, k)
        ctx.mark_non_differentiable(win, cellmap, cand_idx, cand_pass, probs, n_pass)
        ctx.set_materialize_grads(False)
        return deltas[0], deltas[1], win, cellmap, cand_idx, cand_pass, probs, n_pass

    @staticmethod
    def backward(ctx, d_scores_delta, d_gates_delta, *unused):
        probs, win, sd, n_pass = ctx.saved_tensors
        tau, B, Hw, Ww, k = ctx.meta
        dev = probs.device
        zero = lambda: torch.zeros(B * k, dtype=torch.float32, device=dev)
        dsd = zero() if d_scores_delta is None else d_scores_delta.to(torch.float32).contiguous()
        dgd = zero() if d_gates_delta is None else d_gates_delta.to(torch.float32).contiguous()
        d_sd = torch.empty(B, 4, Hw, Ww, dtype=sd.dtype, device=dev)
        call("dsw_select_bwd_sd", ptr(probs), ptr(dsd), ptr(dgd), ptr(win[0]), ptr(win[1]), B * k, _c_float(tau),
             B, Hw, Ww, ptr(sd), dtype_code(sd.dtype), ptr(d_sd), ptr(n_pass), stream(dev))
        return d_sd, None, None, None, None, None, None


def select_windows_static(sd, noise, tau, k, w, H, W):
    """select_windows without the host read of the number of passing windows (the reference has two such reads
    per call, DSwinT_arch.py:273, :291): every candidate gets a window slot (B * k of them); the slots behind
    the passing ones hold a harmless window (image 0, corner 0,0) with zero straight-through values that the
    cell map never points to, so the gather / attention / projections run on all slots, the scatter writes
    back only the passing ones, and in the backward the dead slots receive and contribute exactly zero.
    Selection.n_dev is the pass count on the device.  The sigmoids of sd[:, 1:] are taken inside the kernels."""
    out = _run(_SelectStaticFn, sd, noise, float(tau), int(k), int(w), int(H), int(W))
    sdl, gdl, win, cellmap, cand_idx, cand_pass, probs, n_pass = out
    B = sd.shape[0]
    return Selection(n=B * int(k), win_b=win[0], win_cell=win[1], win_h1=win[2], win_w1=win[3], cellmap=cellmap,
                     scores_delta=sdl, gates_delta=gdl, cand_idx=cand_idx, cand_pass=cand_pass, probs=probs,
                     n_dev=n_pass)


class PendingGrad:
    """A gradient on its way to a tensor that a later backward kernel of the same pass also produces a gradient for:
    the residual around a module (y = x + f(x), DSwinT_arch.py:355-356) hands dy here instead of returning it to
    autograd, and the kernel that writes f's input gradient reads it as one more operand - out = dy + its own
    result - instead of autograd adding the two maps in a pass of its own.  Never modified: it may be a tensor
    that other consumers of dy still read.  The state belongs to one backward pass (see GradSink._same_pass)."""

    def __init__(self):
        self.g = None
        self.task = None

    def put(self, g):
        self.task = torch._C._current_graph_task_id()
        self.g = g

    def take(self):
        g, self.g = self.g, None
        return g if self.task == torch._C._current_graph_task_id() else None


class _ParkGradFn(torch.autograd.Function):
    """identity whose backward parks the gradient in a PendingGrad instead of returning it (stock-op paths)"""

    @staticmethod
    def forward(ctx, x, pending):
        ctx.pending = pending
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.pending.put(g)
        return None, None


class _AddPendingFn(torch.autograd.Function):
    """identity whose backward adds what a PendingGrad holds to the gradient's leading columns (stock-op paths)"""

    @staticmethod
    def forward(ctx, x, pending):
        ctx.pending = pending
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        e = ctx.pending.take()
        if e is not None:
            g = g.clone()
            g[..., :e.shape[-1]] += e.reshape(g.shape[:-1] + (e.shape[-1],))
        return g, None


class GradSink:
    """Meeting point for the gradients that several custom ops send to the SAME input tensor.

    DS-MSA reads its input three times (sparse discriminator, window gather, scatter residual).
    Left to autograd, each backward writes a dense gradient and the engine adds them with two more
    full passes over memory.  With a sink shared by the `expected` ops, the first backward to run
    writes its gradient into a fresh buffer, the later ones add theirs into that buffer inside
    their own kernels (the gather touches only the pixels its windows cover), and only the last
    one returns the buffer to autograd; the others return None.  All `expected` ops must take part
    in every backward pass that needs the input's gradient - true when they consume one tensor
    inside one module's forward - and the order in which they run does not matter."""

    def __init__(self, expected):
        self.expected = expected
        self.count = 0
        self.buf = None
        self.task = None
        self.pending = PendingGrad()   # the residual's gradient, when the caller's residual is taken inside the module

    def _same_pass(self):
        """state belongs to one backward pass: a pass that ended early (an exception, a partial backward over a
        retained graph) must not leave its buffer and count to the next one"""
        task = torch._C._current_graph_task_id()
        if task != self.task:
            self.task, self.count, self.buf = task, 0, None

    def target(self, shape, dtype, device):
        """(buffer to write / add into, accumulate flag)"""
        self._same_pass()
        if self.buf is None:
            return torch.empty(shape, dtype=dtype, device=device), 0
        return self.buf, 1

    def deliver(self, buf):
        self._same_pass()
        self.count += 1
        if self.count == self.expected:
            self.count, self.buf = 0, None
            extra = self.pending.take()
            if extra is not None:   # no kernel of this pass read it as an operand: one pass over our own buffer
                buf.add_(extra.reshape(buf.shape))
            return buf
        self.buf = buf
        return None


def _dx_target(sink, shape, dtype, device):
    return (torch.empty(shape, dtype=dtype, device=device), 0) if sink is None else sink.target(shape, dtype, device)


def _dx_deliver(sink, buf):
    return buf if sink is None else sink.deliver(buf)


class _GatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, win_b, win_h1, win_w1, cellmap, w, out_dtype, sink, pad_out):
        _lib.require_cuda(x)
        x = x.contiguous()
        B, H, W, C = x.shape
        n = win_b.numel()
        ldo = _ceil_pad(C) if pad_out else C
        out = _mark_padded(torch.empty(n, w * w, ldo, dtype=out_dtype, device=x.device))
        call("dsw_gather_fwd", ptr(x), dtype_code(x.dtype), ptr(win_b), ptr(win_h1), ptr(win_w1), n,
             B, H, W, C, w, ptr(out), dtype_code(out_dtype), ldo, stream(x.device))
        ctx.save_for_backward(cellmap, win_h1, win_w1)
        ctx.meta = (B, H, W, C, w, n, x.dtype, sink)
        return out[..., :C]

    @staticmethod
    def backward(ctx, dout):
        cellmap, win_h1, win_w1 = ctx.saved_tensors
        B, H, W, C, w, n, xdt, sink = ctx.meta
        if not ctx.needs_input_grad[0]:
            return (None,) * 9
        dout2, ldg = _row_view4(dout, C)
        dx, acc = _dx_target(sink, (B, H, W, C), xdt, dout.device)
        call("dsw_gather_bwd", _dptr(dout2), dtype_code(dout2.dtype), ptr(cellmap), ptr(win_h1), ptr(win_w1), n,
             B, H, W, C, w, ldg, ptr(dx), dtype_code(xdt), acc, stream(dout.device))
        return (_dx_deliver(sink, dx),) + (None,) * 8


def gather_windows(x, sel, w, out_dtype=None, sink=None, pad_out=False):
    """roi_align of the selected windows straight from NHWC to [n, w*w, C]
    (DSwinT_arch.py:259, :297, :303).  `sink`: see GradSink.  pad_out: the result is a view of the
    first C columns of rows padded (with zeros) to a multiple of 64 elements."""
    return _run(_GatherFn, x, sel.win_b, sel.win_h1, sel.win_w1, sel.cellmap, int(w), out_dtype or x.dtype, sink,
                           bool(pad_out))


class _ScatterMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, after, gates_delta, scores_delta, cellmap, win_b, win_h1, win_w1, w, sink, n_live=None, skip_if_dead=False):
        # skip_if_dead (with n_live): when no window passed, the result is not written at all - the caller's next kernel
        # reads x itself in that case (dwconv3x3_*(..., identity_src=x)) and the pass over the map is saved
        _lib.require_cuda(x, after)
        x = x.contiguous()
        B, H, W, C = x.shape
        after_shape = after.shape
        after, lda = _row_view4(after, C)
        gates_delta = gates_delta.to(torch.float32).contiguous()
        scores_delta = scores_delta.to(torch.float32).contiguous()
        n = win_b.numel()
        y = torch.empty_like(x)
        if skip_if_dead and n_live is not None:
            call("dsw_scatter_mean_fwd_live", ptr(x), dtype_code(x.dtype), _dptr(after), dtype_code(after.dtype),
                 ptr(gates_delta), ptr(scores_delta), ptr(cellmap), ptr(win_h1), ptr(win_w1), n,
                 B, H, W, C, w, lda, ptr(y), dtype_code(y.dtype), ptr(n_live), stream(x.device))
        else:
            call("dsw_scatter_mean_fwd", ptr(x), dtype_code(x.dtype), _dptr(after), dtype_code(after.dtype),
                 ptr(gates_delta), ptr(scores_delta), ptr(cellmap), ptr(win_h1), ptr(win_w1), n,
                 B, H, W, C, w, lda, ptr(y), dtype_code(y.dtype), stream(x.device))
        ctx.save_for_backward(x, after, gates_delta, scores_delta, cellmap, win_b, win_h1, win_w1, n_live)
        ctx.w = w
        ctx.sink = sink
        ctx.after_meta = (after_shape, lda)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, after, gates_delta, scores_delta, cellmap, win_b, win_h1, win_w1, n_live = ctx.saved_tensors
        w = ctx.w
        B, H, W, C = x.shape
        n = win_b.numel()
        dy = dy.contiguous()
        dev = dy.device
        sink = ctx.sink if ctx.needs_input_grad[0] else None
        dx, acc = _dx_target(sink, x.shape, x.dtype, x.device)
        after_shape, lda = ctx.after_meta
        # d_after keeps the row stride of `after` (padded rows stay padded for the projection's backward)
        d_after = _mark_padded(torch.empty(after.shape[0], lda, dtype=after.dtype, device=dev))
        dg = torch.empty(2, n, dtype=torch.float32, device=dev)
        partial = torch.empty(2 * n * w, dtype=torch.float32, device=dev)
        extra = sink.pending.take() if sink is not None and acc == 0 else None
        if extra is not None and not (extra.dtype == dx.dtype and extra.shape == dx.shape and extra.is_contiguous()):
            sink.pending.put(extra)   # not a map this kernel can read: the sink's last delivery adds it
            extra = None
        if extra is not None:
            call("dsw_scatter_mean_bwd_add", ptr(dy), dtype_code(dy.dtype), ptr(x), dtype_code(x.dtype),
                 _dptr(after), dtype_code(after.dtype), ptr(gates_delta), ptr(scores_delta), ptr(cellmap),
                 ptr(win_b), ptr(win_h1), ptr(win_w1), n, B, H, W, C, w, lda, ptr(dx), dtype_code(dx.dtype), ptr(extra),
                 ptr(d_after), dtype_code(d_after.dtype), lda, ptr(dg[0]), ptr(dg[1]), ptr(partial),
                 None if n_live is None else ptr(n_live), stream(dev))
        else:
            call("dsw_scatter_mean_bwd", ptr(dy), dtype_code(dy.dtype), ptr(x), dtype_code(x.dtype),
                 _dptr(after), dtype_code(after.dtype), ptr(gates_delta), ptr(scores_delta), ptr(cellmap),
                 ptr(win_b), ptr(win_h1), ptr(win_w1), n, B, H, W, C, w, lda, ptr(dx), dtype_code(dx.dtype), acc,
                 ptr(d_after), dtype_code(d_after.dtype), lda, ptr(dg[0]), ptr(dg[1]), ptr(partial),
                 None if n_live is None else ptr(n_live), stream(dev))
        return _dx_deliver(sink, dx), d_after[:, :C].view(after_shape), dg[0], dg[1], None, None, None, None, None, None, None, None


SKIP_DEAD_SCATTER = True   # False: the static DS-MSA path writes the scattered map even when no window passed


def scatter_mean_residual(x, after, sel, w, sink=None, skip_if_dead=False):
    """x + mean over covering windows of (after - roi_align(x)) * gates * scores
    (DSwinT_arch.py:308-315), NHWC in and out.  `sink`: see GradSink.  skip_if_dead (selections with a device-side
    count): the result is left unwritten when no window passed; only for a caller whose next kernel then reads x
    instead (identity_src of the dwconv3x3 functions)."""
    return _run(_ScatterMeanFn, x, after, sel.gates_delta, sel.scores_delta, sel.cellmap, sel.win_b,
                                sel.win_h1, sel.win_w1, int(w), sink, sel.n_dev, bool(skip_if_dead and SKIP_DEAD_SCATTER))


def _identity_src(x, src, ident):
    """pointer of the tensor the depthwise kernels copy when the counter holds 0, or None"""
    if src is None or ident is None:
        return None
    assert src.shape == x.shape and src.dtype == x.dtype and src.is_contiguous() and src.data_ptr() % 16 == 0
    return ptr(src)


def _row_view(t, n):
    """(2-D tensor to hand to the kernels, row stride in elements) for a tensor whose last dim is n.
    A tensor whose rows are contiguous and evenly spaced (a dense tensor, or a view of the first n
    columns of a buffer with padded rows) is taken as it is; anything else is made dense."""
    t2 = t.reshape(-1, n)
    rows = t2.shape[0]
    ok = t2.stride(1) == 1 and (rows == 1 or t2.stride(0) == n or (t2.stride(0) > n and t2.stride(0) % 2 == 0)) and \
        (t2.stride(0) == n or t2.data_ptr() % 4 == 0)
    if not ok:
        t2 = t2.contiguous()
    return t2, (t2.stride(0) if rows > 1 else n)


def _row_view4(t, n):
    """as _row_view, for the kernels that move 4 elements per lane (row stride a multiple of 4,
    rows 16-byte aligned for fp32 / 8-byte for bf16)"""
    t2 = t.reshape(-1, n)
    rows = t2.shape[0]
    ok = t2.stride(1) == 1 and (rows == 1 or (t2.stride(0) >= n and t2.stride(0) % 4 == 0)) and \
        t2.data_ptr() % (4 * t2.element_size()) == 0
    if not ok:
        t2 = t2.contiguous()
    return t2, (t2.stride(0) if rows > 1 else n)


def _dptr(t):
    return ctypes.c_void_p(t.data_ptr())


class _WinAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, table, lepe_w, lepe_b, layout, n_img, H, W, w, heads, scale, pad_out, drop_p=0.0, drop_seed=0, n_live=None):
        _lib.require_cuda(qkv)
        C3 = qkv.shape[-1]
        C = C3 // 3
        lead = qkv.shape[:-1]
        qkv2, ldq = _row_view(qkv, C3)
        rows = qkv2.shape[0]
        L = w * w
        n_win = n_img * (H // w) * (W // w) if H else n_img
        assert rows == n_win * L, "qkv rows %d != windows %d * %d" % (rows, n_win, L)
        dev = qkv.device
        tab = None if table is None else table.detach().to(torch.float32).contiguous()
        lw = None if lepe_w is None else lepe_w.detach().to(torch.float32).reshape(C, 9).contiguous()
        lb = None if lepe_b is None else lepe_b.detach().to(torch.float32).contiguous()
        ldo = _ceil_pad(C) if pad_out else C
        out = _mark_padded(torch.empty(rows, ldo, dtype=qkv.dtype, device=dev))
        lse = torch.empty(n_win, heads, L, dtype=torch.float32, device=dev)
        call("dsw_win_attn_fwd", _dptr(qkv2), dtype_code(qkv.dtype), layout, n_img, H, W, w, C, heads, ldq, ldo,
             ptr(tab), _c_float(scale), ptr(lw), ptr(lb), ptr(out), ptr(lse), _c_float(drop_p), ctypes.c_ulonglong(drop_seed),
             None if n_live is None else ptr(n_live), stream(dev))
        ctx.save_for_backward(qkv2, out, lse, tab, lw, lb)
        ctx.drop = (float(drop_p), int(drop_seed))
        ctx.n_live = n_live
        ctx.meta = (layout, n_img, H, W, w, C, heads, scale, ldq, ldo, lead,
                    None if table is None else (table.shape, table.dtype),
                    None if lepe_w is None else (lepe_w.shape, lepe_w.dtype, lepe_b.dtype))
        return out[:, :C].view(lead + (C,))

    @staticmethod
    def backward(ctx, dout):
        qkv2, out, lse, tab, lw, lb = ctx.saved_tensors
        layout, n_img, H, W, w, C, heads, scale, ldq, ldo, lead, tmeta, lmeta = ctx.meta
        dev = qkv2.device
        if dout.dtype != qkv2.dtype:
            dout = dout.to(qkv2.dtype)
        dout2, ldg = _row_view(dout, C)
        rows = qkv2.shape[0]
        # dqkv keeps the row stride of qkv: a padded qkv (from a projection that left its result in
        # padded rows) gets a padded gradient, padding columns zeroed by the kernel
        dqkv = _mark_padded(torch.empty(rows, ldq, dtype=qkv2.dtype, device=dev))
        d_tab = None if tab is None else torch.empty_like(tab)
        d_lw = None if lw is None else torch.empty_like(lw)
        d_lb = None if lb is None else torch.empty_like(lb)
        lib = _lib.load()
        nbytes = lib.dsw_win_attn_bwd_workspace(n_img, H, W, w, C, heads, int(tab is not None), int(lw is not None))
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)
        call("dsw_win_attn_bwd", _dptr(qkv2), ptr(out), _dptr(dout2), dtype_code(qkv2.dtype), layout, n_img, H, W, w, C,
             heads, ldq, ldo, ldg, ptr(tab), _c_float(scale), ptr(lw), ptr(lb), ptr(lse), ptr(dqkv), ptr(d_tab),
             ptr(d_lw), ptr(d_lb), ptr(ws), ctypes.c_size_t(int(nbytes)), _c_float(ctx.drop[0]), ctypes.c_ulonglong(ctx.drop[1]),
             None if ctx.n_live is None else ptr(ctx.n_live), stream(dev))
        g_tab = None if tab is None else d_tab.view(tmeta[0]).to(tmeta[1])
        g_lw = None if lw is None else d_lw.view(lmeta[0]).to(lmeta[1])
        g_lb = None if lb is None else d_lb.to(lmeta[2])
        return (dqkv[:, :3 * C].view(lead + (3 * C,)), g_tab, g_lw, g_lb) + (None,) * 11


def draw_dropout_seed():
    """64 bits from torch's CPU generator (no device sync; reproducible under torch.manual_seed, and replayed by
    activation checkpointing, which restores the CPU generator's state before it recomputes a forward)"""
    return int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF


def window_attention(qkv, table, w, heads, layout, grid=None, scale=None, lepe_w=None, lepe_b=None, pad_out=False,
                     dropout_p=0.0, dropout_seed=None, n_live=None):
    """softmax(q k^T * scale + bias) v per (window, head), de-headified.

    dropout_p > 0: attention dropout on the probabilities, decided inside the kernels by a hash of
    dropout_seed (drawn from torch's CPU generator when None) and the element's position.

    qkv [..., 3C].  grid=(B, H, W): tokens are the pixels of B images, windows are cut out by
    address arithmetic and the result keeps the image layout.  grid=None: qkv is [n, w*w, 3C],
    windows already contiguous.  table: [(2w-1)^2, heads] relative-position bias or None.
    lepe_w/lepe_b: depthwise 3x3 conv [C,1,3,3]/[C] applied to V inside each window and added.
    qkv may be a view of the first 3C columns of a buffer with padded rows (see `linear`); with
    pad_out the result is such a view too (rows padded to a multiple of 64 elements, zeros).
    n_live (packed windows only): int32 counter on the device; windows with index >= its value are switched off -
    no arithmetic, zero rows in the result, zero gradients.
    """
    C = qkv.shape[-1] // 3
    if scale is None:
        scale = float((C // heads) ** -0.5)
    if grid is None:
        n_img, H, W = qkv.shape[0], 0, 0
    else:
        n_img, H, W = grid
    dropout_p = float(dropout_p)
    if dropout_p > 0.0 and dropout_seed is None:
        dropout_seed = draw_dropout_seed()
    return _run(_WinAttnFn, qkv, table, lepe_w, lepe_b, int(layout), int(n_img), int(H), int(W), int(w),
                            int(heads), float(scale), bool(pad_out), dropout_p, int(dropout_seed or 0), n_live)


FUSE_PROJECTION = True   # False: projected_window_attention always declines (the two separate kernels run)


# Image-grid calls at 16x16 windows and head dim 30 (DSWT's WindowedMSA) have a forward kernel of their own
# (win_attn_fwd_w16_kernel) that, with the projection kernel in front of it, is faster than the fusion built on the
# general attention core (106 against 120 us at 131 072 tokens), so WindowedMSA does not ask for the fusion there;
# the packed-window calls of DS-MSA, where a quarter of the tokens are attended, stay fused (36 against 42 us).
# DSW_FUSE_PROJ_GRID=1 in the environment (or setting this flag) asks for it everywhere.
FUSE_PROJECTION_GRID_D30 = os.environ.get("DSW_FUSE_PROJ_GRID") == "1"


def projected_window_attention(x, lin, table, w, heads, layout, grid=None, scale=None, pad_out=False, n_live=None):
    """window_attention(linear(x, lin), ...) as one kernel (dsw_win_attn_proj_fwd): q|k|v are projected per
    (window, head) inside the attention kernel and never written to memory.  Inference under bf16 autocast only,
    head dim <= 32, x in zero-padded bf16 rows (what rms_norm(..., for_projection=True) and
    gather_windows(..., pad_out=True) return); returns None when the call is not one the kernel takes, and the
    caller then composes linear() and window_attention()."""
    if not FUSE_PROJECTION or torch.is_grad_enabled() or autocast_dtype(None) != torch.bfloat16:
        return None
    weight, bias = lin.weight, lin.bias
    N, C = weight.shape
    if not (x.is_cuda and x.dtype == torch.bfloat16 and N == 3 * C and C % heads == 0 and C // heads <= 32 and w in (8, 16)
            and weight.is_contiguous() and weight.dtype in (torch.float32, torch.bfloat16) and weight.data_ptr() % 16 == 0):
        return None
    x2 = x.reshape(-1, C)
    xk = _padded_rows(x2, torch.bfloat16)
    if xk is None:
        if C % 64 or not x2.is_contiguous():
            return None
        xk = x2
    if xk.data_ptr() % 16:
        return None
    if grid is None:
        n_img, H, W = x.shape[0], 0, 0
    else:
        n_img, H, W = grid
    L = w * w
    n_win = n_img * (H // w) * (W // w) if H else n_img
    if xk.shape[0] != n_win * L:
        return None
    if scale is None:
        scale = float((C // heads) ** -0.5)
    w_p, _ = _weight_operands(weight)
    b32 = None if bias is None else bias.detach().float().contiguous()
    tab = None if table is None else table.detach().to(torch.float32).contiguous()
    ldo = _ceil_pad(C) if pad_out else C
    out = _mark_padded(torch.empty(xk.shape[0], ldo, dtype=torch.bfloat16, device=x.device))
    call("dsw_win_attn_proj_fwd", _dptr(xk), xk.stride(0), ptr(w_p), w_p.shape[0], w_p.shape[1], ptr(b32), int(layout), int(n_img),
         int(H), int(W), int(w), C, int(heads), ldo, ptr(tab), _c_float(scale), ptr(out), None, None if n_live is None else ptr(n_live),
         stream(x.device))
    return out[:, :C].view(x.shape[:-1] + (C,))


# ---------------------------------------------------------------------------------------------
# MSFT multi-scale reshapes (msft_arch.py:37-78, :156-167)
# ---------------------------------------------------------------------------------------------
class _UnshuffleLNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, s, eps, out_dtype):
        _lib.require_cuda(x)
        x = x.contiguous()
        B, c, H, W = x.shape
        Co = c * s * s
        dev = x.device
        g32 = gamma.detach().to(torch.float32).contiguous()
        b32 = beta.detach().to(torch.float32).contiguous()
        y = torch.empty(B, H // s, W // s, Co, dtype=out_dtype, device=dev)
        stats = torch.empty(2, B * (H // s) * (W // s), dtype=torch.float32, device=dev)
        call("dsw_unshuffle_nhwc_ln_fwd", ptr(x), dtype_code(x.dtype), B, c, H, W, s, ptr(g32), ptr(b32), _c_float(eps),
             ptr(y), dtype_code(out_dtype), ptr(stats[0]), ptr(stats[1]), stream(dev))
        ctx.save_for_backward(x, g32, stats)
        ctx.meta = (s, gamma.dtype, beta.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g32, stats = ctx.saved_tensors
        s, gdt, bdt = ctx.meta
        B, c, H, W = x.shape
        Co = c * s * s
        dev = x.device
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        dgb = torch.empty(2, Co, dtype=torch.float32, device=dev)
        nblk = _lib.load().dsw_unshuffle_blocks(B, H, W, s)
        partial = torch.empty(nblk * 2 * Co, dtype=torch.float32, device=dev)
        call("dsw_unshuffle_nhwc_ln_bwd", ptr(dy), ptr(x), dtype_code(x.dtype), B, c, H, W, s, ptr(g32), ptr(stats[0]),
             ptr(stats[1]), ptr(dx), ptr(dgb[0]), ptr(dgb[1]), ptr(partial), stream(dev))
        return dx, dgb[0].to(gdt), dgb[1].to(bdt), None, None, None


def unshuffle_nhwc_layernorm(x, gamma, beta, scale=1, eps=1e-5, out_dtype=None):
    """PixelUnshuffle(scale) + NCHW->NHWC + LayerNorm over channels in one pass
    (msft_arch.py:45-50; scale=1: :68-70 and :160-162).  x [B,c,H,W] -> [B,H/s,W/s,c*s*s]."""
    return _run(_UnshuffleLNFn, x, gamma, beta, int(scale), float(eps), out_dtype or x.dtype)


class _ShuffleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, s):
        _lib.require_cuda(t)
        t = t.contiguous()
        B, H, W, C = t.shape
        y = torch.empty(B, C // (s * s), H * s, W * s, dtype=t.dtype, device=t.device)
        call("dsw_shuffle_nchw_fwd", ptr(t), dtype_code(t.dtype), B, H, W, C, s, ptr(y), stream(t.device))
        ctx.s = s
        return y

    @staticmethod
    def backward(ctx, dy):
        return _UnshuffleFn.apply(dy, ctx.s), None


class _UnshuffleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        _lib.require_cuda(x)
        x = x.contiguous()
        B, c, H, W = x.shape
        y = torch.empty(B, H // s, W // s, c * s * s, dtype=x.dtype, device=x.device)
        call("dsw_unshuffle_nhwc_fwd", ptr(x), dtype_code(x.dtype), B, c, H, W, s, ptr(y), stream(x.device))
        ctx.s = s
        return y

    @staticmethod
    def backward(ctx, dy):
        return _ShuffleFn.apply(dy, ctx.s), None


def shuffle_nchw(t, scale=1):
    """NHWC->NCHW + PixelShuffle(scale) in one pass (msft_arch.py:54, :74-76, :166).
    t [B,H,W,C] -> [B,C/s^2,H*s,W*s]."""
    return _run(_ShuffleFn, t, int(scale))


def unshuffle_nhwc(x, scale=1):
    """PixelUnshuffle(scale) + NCHW->NHWC in one pass.  x [B,c,H,W] -> [B,H/s,W/s,c*s*s]."""
    return _run(_UnshuffleFn, x, int(scale))


# ---------------------------------------------------------------------------------------------
# NHWC convolutions of the DS-MSA path (DSwinT_arch.py:180-186, :198, :266, :317)
# ---------------------------------------------------------------------------------------------
class _DWConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, out_dtype, identity_if_zero=None, identity_src=None):
        # identity_src: the tensor to copy when the counter holds 0, in place of x (whose producer left it unwritten then)
        _lib.require_cuda(x)
        x = x.contiguous()
        B, H, W, C = x.shape
        w32 = weight.detach().to(torch.float32).reshape(C, 9).contiguous()
        b32 = None if bias is None else bias.detach().to(torch.float32).contiguous()
        y = torch.empty(B, H, W, C, dtype=out_dtype, device=x.device)
        src = _identity_src(x, identity_src, identity_if_zero)
        if src is not None:
            call("dsw_dwconv3x3_fwd_src", ptr(x), dtype_code(x.dtype), ptr(w32), ptr(b32), B, H, W, C, ptr(y),
                 dtype_code(out_dtype), ptr(identity_if_zero), src, stream(x.device))
        else:
            call("dsw_dwconv3x3_fwd", ptr(x), dtype_code(x.dtype), ptr(w32), ptr(b32), B, H, W, C, ptr(y),
                 dtype_code(out_dtype), None if identity_if_zero is None else ptr(identity_if_zero), stream(x.device))
        ctx.save_for_backward(x, w32, identity_if_zero)
        ctx.meta = (weight.shape, weight.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w32, ident = ctx.saved_tensors
        wshape, wdt, bdt = ctx.meta
        B, H, W, C = x.shape
        dev = x.device
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dw = torch.empty(C, 9, dtype=torch.float32, device=dev)
        db = torch.empty(C, dtype=torch.float32, device=dev)
        nblk = _lib.load().dsw_dwconv3x3_bwd_blocks(B, H, W)
        partial = torch.empty((nblk + 1) * C * 10, dtype=torch.float32, device=dev)
        call("dsw_dwconv3x3_bwd", ptr(dy), dtype_code(dy.dtype), ptr(x), dtype_code(x.dtype), ptr(w32), B, H, W, C,
             ptr(dx), dtype_code(dx.dtype), ptr(dw), ptr(db), ptr(partial), None if ident is None else ptr(ident), stream(dev))
        return dx, dw.view(wshape).to(wdt), (None if bdt is None else db.to(bdt)), None, None, None


def dwconv3x3_nhwc(x, weight, bias=None, out_dtype=None, identity_if_zero=None, identity_src=None):
    """Depthwise 3x3 conv, stride 1, zero padding 1, on an NHWC tensor (out_conv of DS-MSA,
    DSwinT_arch.py:198, :317).  weight [C,1,3,3].  identity_if_zero: an int32 device counter; the result
    is x itself (cast) when it holds 0, and the backward then passes the gradient through to x alone.
    identity_src: the tensor that is copied in that case instead of x (see scatter_mean_residual's skip_if_dead)."""
    return _run(_DWConvFn, x, weight, bias, out_dtype or x.dtype, identity_if_zero, identity_src)


class _DWConvResidualFn(torch.autograd.Function):
    """resid + dwconv3x3(x) as fp32 (dsw_dwconv3x3_residual_fwd with a scale of one): DS-MSA's closing convolution
    with the plain residual of the block around it.  The backward reads the fp32 gradient as it arrives; the
    residual's share of it - dy itself - goes to `pending` when the caller gives one (PendingGrad) and back to
    autograd otherwise."""

    @staticmethod
    def forward(ctx, x, weight, bias, resid, pending, identity_if_zero=None, identity_src=None):
        _lib.require_cuda(x, resid)
        x = x.contiguous()
        B, H, W, C = x.shape
        w32 = weight.detach().to(torch.float32).reshape(C, 9).contiguous()
        b32 = None if bias is None else bias.detach().to(torch.float32).contiguous()
        one = torch.ones(1, dtype=torch.float32, device=x.device)
        y = torch.empty(B, H, W, C, dtype=torch.float32, device=x.device)
        src = _identity_src(x, identity_src, identity_if_zero)
        if src is not None:
            call("dsw_dwconv3x3_residual_fwd_src", ptr(x), dtype_code(x.dtype), ptr(w32), ptr(b32), B, H, W, C, ptr(resid), ptr(one), ptr(y),
                 ptr(identity_if_zero), src, stream(x.device))
        else:
            call("dsw_dwconv3x3_residual_fwd", ptr(x), dtype_code(x.dtype), ptr(w32), ptr(b32), B, H, W, C, ptr(resid), ptr(one), ptr(y),
                 None if identity_if_zero is None else ptr(identity_if_zero), stream(x.device))
        ctx.save_for_backward(x, w32, identity_if_zero)
        ctx.meta = (weight.shape, weight.dtype, None if bias is None else bias.dtype)
        ctx.pending = pending
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w32, ident = ctx.saved_tensors
        wshape, wdt, bdt = ctx.meta
        B, H, W, C = x.shape
        dev = x.device
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dw = torch.empty(C, 9, dtype=torch.float32, device=dev)
        db = torch.empty(C, dtype=torch.float32, device=dev)
        nblk = _lib.load().dsw_dwconv3x3_bwd_blocks(B, H, W)
        partial = torch.empty((nblk + 1) * C * 10, dtype=torch.float32, device=dev)
        call("dsw_dwconv3x3_bwd", ptr(dy), dtype_code(dy.dtype), ptr(x), dtype_code(x.dtype), ptr(w32), B, H, W, C,
             ptr(dx), dtype_code(dx.dtype), ptr(dw), ptr(db), ptr(partial), None if ident is None else ptr(ident), stream(dev))
        d_resid = None
        if ctx.needs_input_grad[3]:
            if ctx.pending is not None:
                ctx.pending.put(dy)
            else:
                d_resid = dy
        return dx, dw.view(wshape).to(wdt), (None if bdt is None else db.to(bdt)), d_resid, None, None, None


def dwconv3x3_plus_residual_nhwc(x, weight, bias, resid, pending=None, identity_if_zero=None, identity_src=None):
    """resid + dwconv3x3(x), fp32, with autograd (see _DWConvResidualFn); None when the call is not one the kernel
    takes - the caller then composes dwconv3x3_nhwc() and an addition."""
    if not (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and resid.is_cuda and resid.dtype == torch.float32
            and resid.is_contiguous() and resid.shape == x.shape and x.shape[-1] % 4 == 0 and resid.data_ptr() % 16 == 0):
        return None
    return _run(_DWConvResidualFn, x, weight, bias, resid, pending, identity_if_zero, identity_src)


def dwconv3x3_residual_nhwc(x, weight, bias, resid, r, identity_if_zero=None, identity_src=None):
    """resid + r * dwconv3x3(x) as fp32 in one kernel (dsw_dwconv3x3_residual_fwd): DS-MSA's closing convolution with
    the ReZero residual of the block around it (DSwinT_arch.py:317 with :356).  Without autograd only; returns None
    when the call is not one the kernel takes, and the caller then composes dwconv3x3_nhwc() and rezero_residual()."""
    if torch.is_grad_enabled() or not (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and resid.is_cuda
                                       and resid.dtype == torch.float32 and resid.is_contiguous() and resid.shape == x.shape
                                       and r.numel() == 1 and x.shape[-1] % 4 == 0 and resid.data_ptr() % 16 == 0):
        return None
    x = x.contiguous()
    B, H, W, C = x.shape
    w32 = weight.detach().to(torch.float32).reshape(C, 9).contiguous()
    b32 = None if bias is None else bias.detach().to(torch.float32).contiguous()
    r32 = r.detach().float().reshape(1).contiguous()
    y = torch.empty(B, H, W, C, dtype=torch.float32, device=x.device)
    src = _identity_src(x, identity_src, identity_if_zero)
    if src is not None:
        call("dsw_dwconv3x3_residual_fwd_src", ptr(x), dtype_code(x.dtype), ptr(w32), ptr(b32), B, H, W, C, ptr(resid), ptr(r32), ptr(y),
             ptr(identity_if_zero), src, stream(x.device))
    else:
        call("dsw_dwconv3x3_residual_fwd", ptr(x), dtype_code(x.dtype), ptr(w32), ptr(b32), B, H, W, C, ptr(resid), ptr(r32), ptr(y),
             None if identity_if_zero is None else ptr(identity_if_zero), stream(x.device))
    return y


class _SparseDiscFn(torch.autograd.Function):
    """The sparse discriminator.  fp32 runs: NHWC fp32 kernels for all three convolutions.  bf16 runs:
    the first convolution (C -> 4, 3x3: nearly all of the arithmetic and traffic) goes to the matrix
    cores as a [pixels, C] x [C, 36] product followed by a 9-point gather, and its input / weight
    gradients are the transposed products (see dsw_sd_stencil_fwd in the header)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w2, b2, w4, b4, w, out_dtype, sink):
        _lib.require_cuda(x)
        x = x.contiguous()
        B, H, W, C = x.shape
        dev = x.device
        ps = [p.detach().to(torch.float32).contiguous() for p in (w0, b0, w2, b2, w4, b4)]
        t0 = torch.empty(B, H, W, 4, dtype=torch.float32, device=dev)
        u = torch.empty(B, 4, H // w, W // w, dtype=torch.float32, device=dev)
        sd = torch.empty(B, 4, H // w, W // w, dtype=out_dtype, device=dev)
        Kp = _ceil_pad(C)
        mfma = (out_dtype == torch.bfloat16 and C % 4 == 0 and Kp <= 192 and x.dtype in (torch.float32, torch.bfloat16)
                and _mfma_operand(x.view(-1, C)) is not None)
        if mfma:
            M = B * H * W
            wf = ps[0].permute(2, 3, 0, 1).reshape(36, C).contiguous()   # Wf[4 t + o][c] = w0[o][c][t]
            w_p, w_pt = _weight_operands(wf)
            xk = _mark_padded(torch.empty(M, Kp, dtype=torch.bfloat16, device=dev)) if ctx.needs_input_grad[1] else None
            P = torch.empty(M, 36, dtype=torch.float32, device=dev)
            _mfma_linear(x.view(M, C), C, w_p, None, 0, P, 36, xk)
            call("dsw_sd_stencil_fwd", ptr(P), 36, ptr(ps[1]), B, H, W, ptr(t0), stream(dev))
            call("dsw_sparse_disc_tail_fwd", ptr(t0), *[ptr(p) for p in ps[2:]], B, H, W, w, ptr(u), ptr(sd), dtype_code(out_dtype),
                 stream(dev))
            ctx.save_for_backward(xk if xk is not None else x, w_pt, ps[2], ps[4], t0, u)
        else:
            call("dsw_sparse_disc_fwd", ptr(x), dtype_code(x.dtype), *[ptr(p) for p in ps], B, H, W, C, w, ptr(t0), ptr(u),
                 ptr(sd), dtype_code(out_dtype), stream(dev))
            ctx.save_for_backward(x, ps[0], ps[2], ps[4], t0, u)
        ctx.meta = (w, [(p.shape, p.dtype) for p in (w0, b0, w2, b2, w4, b4)], sink, mfma, (x.shape, x.dtype))
        return sd

    @staticmethod
    def backward(ctx, dsd):
        x, w0, w2, w4, t0, u = ctx.saved_tensors
        w, pmeta, sink, mfma, (xshape, xdtype) = ctx.meta
        B, H, W, C = xshape
        dev = x.device
        dsd = dsd.contiguous()
        if not ctx.needs_input_grad[0]:
            sink = None
        dx, acc = _dx_target(sink, xshape, xdtype, dev)
        f32 = dict(dtype=torch.float32, device=dev)
        g = [torch.empty(4 * C * 9, **f32), torch.empty(4, **f32), torch.empty(16 * w * w, **f32), torch.empty(4, **f32),
             torch.empty(144, **f32), torch.empty(4, **f32)]
        if mfma:
            M = B * H * W
            xk, w_pt = x, w0
            du = torch.empty(B * 4 * (H // w) * (W // w), **f32)
            dt0 = torch.empty(M, 4, **f32)
            call("dsw_sparse_disc_tail_bwd", ptr(dsd), dtype_code(dsd.dtype), ptr(w2), ptr(w4), ptr(t0), ptr(u), B, H, W, w, ptr(du),
                 ptr(dt0), ptr(g[2]), ptr(g[3]), ptr(g[4]), ptr(g[5]), stream(dev))
            G = torch.empty(M, 64, dtype=torch.bfloat16, device=dev)
            call("dsw_sd_gather_bwd", ptr(dt0), B, H, W, ptr(G), stream(dev))
            if ctx.needs_input_grad[0]:
                if dx.dtype == torch.float32:
                    _mfma_linear(G, 64, w_pt, None, 0, dx.view(M, C), C, None, accumulate=bool(acc))
                else:
                    d32 = torch.empty(M, C, **f32)
                    _mfma_linear(G, 64, w_pt, None, 0, d32, C, None)
                    dx.view(M, C).add_(d32) if acc else dx.view(M, C).copy_(d32)
            if ctx.needs_input_grad[1]:
                Kp = xk.shape[1]
                dwf = torch.empty(36, C, **f32)
                dbf = torch.empty(36, **f32) if C < Kp else None
                ws = torch.empty(int(_lib.load().dsw_linear_wgrad_workspace(M, 36, C)), **f32)
                call("dsw_linear_wgrad", ptr(G), 64, ptr(xk), Kp, M, 36, C, ptr(dwf), None if dbf is None else ptr(dbf), ptr(ws), None, 0, stream(dev))
                g[0] = dwf.view(3, 3, 4, C).permute(2, 3, 0, 1).contiguous()
                g[1] = dbf[16:20] if dbf is not None else dt0.sum(0)
            else:
                g[0] = g[1] = None
        else:
            nfl = _lib.load().dsw_sparse_disc_bwd_scratch(B, H, W, C, w)
            scratch = torch.empty(int(nfl), **f32)
            call("dsw_sparse_disc_bwd", ptr(dsd), dtype_code(dsd.dtype), ptr(x), dtype_code(x.dtype), ptr(w0), ptr(w2), ptr(w4),
                 ptr(t0), ptr(u), B, H, W, C, w, ptr(dx), dtype_code(dx.dtype), acc, *[ptr(t) for t in g], ptr(scratch),
                 stream(dev))
        grads = [None if t is None else t.reshape(shape).to(dt) for t, (shape, dt) in zip(g, pmeta)]
        return (_dx_deliver(sink, dx), *grads, None, None, None)


def sparse_discriminator_nhwc(x, convs, w, out_dtype=None, sink=None):
    """The DS-MSA sparse discriminator on an NHWC tensor: conv C->4 3x3 + SiLU + conv 4->4 k=w
    stride w + SiLU + conv 4->4 3x3 (DSwinT_arch.py:180-186, :266).  `convs` = the three Conv2d
    modules; returns sd [B,4,H/w,W/w]."""
    c0, c2, c4 = convs
    return _run(_SparseDiscFn, x, c0.weight, c0.bias, c2.weight, c2.bias, c4.weight, c4.bias, int(w),
                               out_dtype or x.dtype, sink)


# ---------------------------------------------------------------------------------------------
# Linear projections of the attention modules (DSwinT_arch.py:303, :308, :334, :337)
# ---------------------------------------------------------------------------------------------
_WGRAD_CHUNK = 2048


def colsum(x2):
    """fp32 column sums of a contiguous 2-D tensor (bias gradients), in a fixed summation order."""
    rows, cols = x2.shape
    if cols % 4 or cols > 2048 or x2.dtype not in (torch.float32, torch.bfloat16):
        return x2.sum(0, dtype=torch.float32)
    _lib.require_cuda(x2)
    x2 = x2.contiguous()
    out = torch.empty(cols, device=x2.device, dtype=torch.float32)
    part = torch.empty(_lib.load().dsw_colsum_blocks(rows) * cols, device=x2.device, dtype=torch.float32)
    _lib.call("dsw_colsum", _lib.ptr(x2), _lib.dtype_code(x2.dtype), rows, cols, _lib.ptr(out), _lib.ptr(part), _lib.stream(x2.device))
    return out


_PAD = 64


def _ceil_pad(n):
    return (n + _PAD - 1) // _PAD * _PAD


# how often a projection took an operand with its padding as it stood / had to re-pad it by a copy
PAD_STATS = {"taken": 0, "copied": 0}


def _mark_padded(buf):
    """Tag a buffer whose padding columns this module keeps at zero (see _padded_rows)."""
    buf._dsw_zero_padded = True
    return buf


def _padded_rows(t2, act):
    """If the 2-D view t2 [T, n] is the first n columns of a [T, ceil64(n)] buffer that this module
    created with zero padding columns, return that buffer as a view, else None.  The GEMMs below
    multiply the padding columns by zero weights, which only works if they hold finite numbers:
    rows that merely look padded (a gradient laid out by autograd, a slice of a foreign tensor)
    may hold anything there and are copied into a fresh padded buffer instead."""
    T, n = t2.shape
    ld = _ceil_pad(n)
    base = t2._base if t2._base is not None else t2
    if not getattr(base, "_dsw_zero_padded", False):
        return None
    if ld == n or t2.dtype != act or t2.stride(1) != 1 or (T > 1 and t2.stride(0) != ld):
        return None
    if (t2.storage_offset() + T * ld) * t2.element_size() > t2.untyped_storage().nbytes():
        return None
    return t2.as_strided((T, ld), (ld, 1))


def cast_rows(t2, dtype, ld=None):
    """[T, n] (rows contiguous, any row stride) -> [T, ld or n] of `dtype`; with ld > n the result has
    zero padding columns.  One vectorised pass (dsw_cast_rows); torch ops for shapes it does not take."""
    T, n = t2.shape
    ld = n if ld is None else ld
    ok = (t2.is_cuda and T > 0 and n % 4 == 0 and ld % 4 == 0 and t2.stride(1) == 1 and (T == 1 or t2.stride(0) % 4 == 0)
          and t2.dtype in (torch.float32, torch.bfloat16) and dtype in (torch.float32, torch.bfloat16)
          and t2.storage_offset() % 4 == 0)
    if not ok:
        out = torch.zeros(T, ld, dtype=dtype, device=t2.device)
        out[:, :n].copy_(t2)
        return out
    out = torch.empty(T, ld, dtype=dtype, device=t2.device)
    src_ld = t2.stride(0) if T > 1 else n
    _lib.call("dsw_cast_rows", ctypes.c_void_p(t2.data_ptr()), _lib.dtype_code(t2.dtype), src_ld, _lib.ptr(out),
              _lib.dtype_code(dtype), ld, T, T, n, 1 if ld > n else 0, _lib.stream(t2.device))
    return out


def _padded_matrix(m, dtype, rows_p, cols_p):
    """[r, c] parameter -> fresh [rows_p, cols_p] tensor of `dtype`, zero outside the first r x c"""
    r, c = m.shape
    m = m.detach()
    ok = (m.is_cuda and m.is_contiguous() and c % 4 == 0 and cols_p % 4 == 0 and m.dtype in (torch.float32, torch.bfloat16)
          and dtype in (torch.float32, torch.bfloat16) and m.data_ptr() % 16 == 0)
    if not ok:
        out = torch.zeros(rows_p, cols_p, dtype=dtype, device=m.device)
        out[:r, :c].copy_(m)
        return out
    out = torch.empty(rows_p, cols_p, dtype=dtype, device=m.device)
    _lib.call("dsw_cast_rows", _lib.ptr(m), _lib.dtype_code(m.dtype), c, _lib.ptr(out), _lib.dtype_code(dtype), cols_p,
              rows_p, r, c, 1 if cols_p > c else 0, _lib.stream(m.device))
    return out


def _cast_padded(t2, act):
    """[T, n] -> a fresh [T, ceil64(n)] tensor of dtype act with zero padding columns"""
    return cast_rows(t2, act, _ceil_pad(t2.shape[1]))


_PREP_CACHE = {}


def _may_keep_operands():
    """Prepared operands are kept between calls only without autograd and outside stream capture.  Under
    capture a kept tensor would live in the capturing graph's private memory pool and dangle for every later
    capture or eager call once that graph is gone; and a graph that skipped the preparation would replay on a
    copy of the weights taken before capture - so a captured forward always prepares its operands itself,
    from the parameters as they are at replay time."""
    return not torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing()


BATCH_WEIGHT_PREP = True   # False: every weight is prepared by a launch of its own when a projection first needs it
_WEIGHT_SCOPE = None


@contextlib.contextmanager
def weight_scope(module):
    """Inside the context the prepared operands of every nn.Linear weight of `module` come from ONE preparation launch
    made at entry (dsw_weight_prep_batch) instead of one launch per projection.  For code that cannot keep operands
    between calls - a forward under stream capture, whose operands must be rebuilt inside the graph from the live
    parameters: the pairs live as long as the context (and as what the forward saved for its backward)."""
    global _WEIGHT_SCOPE
    ws = []
    for m in module.modules():
        w = getattr(m, "weight", None) if isinstance(m, torch.nn.Linear) else None
        if w is not None and w.is_cuda and w.dim() == 2 and w.is_contiguous() and w.dtype in (torch.float32, torch.bfloat16) \
                and w._base is None and all(w is not o for o in ws):
            ws.append(w)
    scope = {}
    by_dev = {}
    for w in ws:
        by_dev.setdefault(w.device, []).append(w)
    for dev, group in by_dev.items():
        n = len(group)
        pairs = [(torch.empty(_ceil_pad(w.shape[0]), _ceil_pad(w.shape[1]), dtype=torch.bfloat16, device=dev),
                  torch.empty(_ceil_pad(w.shape[1]), _ceil_pad(w.shape[0]), dtype=torch.bfloat16, device=dev)) for w in group]
        vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
        tabs = (vp(*[w.data_ptr() for w in group]), ip(*[dtype_code(w.dtype) for w in group]), ip(*[w.shape[0] for w in group]),
                ip(*[w.shape[1] for w in group]), vp(*[a.data_ptr() for a, _ in pairs]), vp(*[b.data_ptr() for _, b in pairs]))
        call("dsw_weight_prep_batch", n, *[ctypes.cast(t, ctypes.c_void_p) for t in tabs], stream(dev))
        for w, pr in zip(group, pairs):
            scope[(id(w), w.shape[0], w.shape[1])] = (w, w._version, w.data_ptr()) + pr
    prev, _WEIGHT_SCOPE = _WEIGHT_SCOPE, scope
    try:
        yield
    finally:
        _WEIGHT_SCOPE = prev


def _weight_operands(weight):
    """nn.Linear weight [N, K] -> zero-padded bf16 [ceil64(N), ceil64(K)] and its transpose.  For a parameter (or a
    whole-tensor view of one: a 1x1 convolution's weight as a matrix) the pair is kept and reused until the
    parameter is modified in place (its version counter moves: an optimizer step, a load_state_dict) or replaced.
    When a kept pair has gone stale, every other kept pair whose parameter has moved on is refreshed in the same
    launch (dsw_weight_prep_batch): after an optimizer step the first projection of the next forward prepares the
    whole network's weights, 48 per launch, instead of one launch per projection.  Under stream capture nothing is
    kept or reused (see _may_keep_operands); a weight computed inside autograd (not a leaf) is prepared per call."""
    N, Kd = weight.shape
    owner = weight._base if weight._base is not None else weight
    keep = (owner.is_leaf and owner.numel() == N * Kd and owner.data_ptr() == weight.data_ptr() and weight.is_contiguous()
            and not (weight.is_cuda and torch.cuda.is_current_stream_capturing()))
    key = (id(owner), N, Kd)
    if _WEIGHT_SCOPE is not None:
        e = _WEIGHT_SCOPE.get(key)
        if e is not None and e[0] is owner and e[1] == owner._version and e[2] == owner.data_ptr():
            return e[3], e[4]
    e = _PREP_CACHE.get(key) if keep else None
    if e is not None and e[0]() is owner and e[2] == owner.data_ptr():
        if e[1] == owner._version:
            return e[3], e[4]
        if BATCH_WEIGHT_PREP:
            _refresh_stale_operands(weight.device)
            e = _PREP_CACHE[key]
            return e[3], e[4]
    Np, Kp = _ceil_pad(N), _ceil_pad(Kd)
    w = weight.detach()
    w_p = torch.empty(Np, Kp, dtype=torch.bfloat16, device=w.device)
    w_pt = torch.empty(Kp, Np, dtype=torch.bfloat16, device=w.device)
    call("dsw_weight_prep", ptr(w), dtype_code(w.dtype), N, Kd, ptr(w_p), ptr(w_pt), stream(w.device))
    if keep:
        if len(_PREP_CACHE) > 4096:
            _PREP_CACHE.clear()
        _PREP_CACHE[key] = (weakref.ref(owner), owner._version, owner.data_ptr(), w_p, w_pt)
    return w_p, w_pt


def _refresh_stale_operands(device):
    """New operand pairs for every kept Linear weight on `device` whose parameter has been modified since its pair was
    made.  The pairs are new tensors, not the old ones overwritten: a backward pass that is still to run may hold
    the old transpose (saved by the forward that used it)."""
    stale = []
    for key, e in list(_PREP_CACHE.items()):
        if len(key) != 3:      # other kinds of kept operands (3x3 convolutions)
            continue
        w = e[0]()
        if w is None or e[2] != w.data_ptr():
            del _PREP_CACHE[key]
        elif w.device == device and e[1] != w._version and w.dtype in (torch.float32, torch.bfloat16):
            stale.append((key, w))
    if not stale:
        return
    n = len(stale)
    pairs = []
    for (_, N, Kd), _w in stale:
        Np, Kp = _ceil_pad(N), _ceil_pad(Kd)
        pairs.append((torch.empty(Np, Kp, dtype=torch.bfloat16, device=device), torch.empty(Kp, Np, dtype=torch.bfloat16, device=device)))
    vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
    tabs = (vp(*[w.data_ptr() for _, w in stale]), ip(*[dtype_code(w.dtype) for _, w in stale]), ip(*[k[1] for k, _ in stale]),
            ip(*[k[2] for k, _ in stale]), vp(*[a.data_ptr() for a, _ in pairs]), vp(*[b.data_ptr() for _, b in pairs]))
    call("dsw_weight_prep_batch", n, *[ctypes.cast(t, ctypes.c_void_p) for t in tabs], stream(device))
    for (key, w), (a, b) in zip(stale, pairs):
        _PREP_CACHE[key] = (weakref.ref(w), w._version, w.data_ptr(), a, b)


def _mfma_operand(t2):
    """Can dsw_linear read the rows of the 2-D tensor t2 where they lie?  -> row stride or None"""
    T, n = t2.shape
    ld = t2.stride(0) if T > 1 else n
    if (t2.dtype in (torch.float32, torch.bfloat16) and t2.stride(1) == 1 and n % 4 == 0 and ld % 4 == 0 and ld >= n
            and t2.data_ptr() % (4 * t2.element_size()) == 0):
        return ld
    return None


def _mfma_linear(a2, K, w_op, bias, N, out, ncols, a_copy, accumulate=False, live=None):
    """out[:, :ncols] (+)= a2[:, :K] @ w_op[:ncols, :]^T (+ bias); a_copy (optional) receives a2 in padded bf16 rows"""
    call("dsw_linear", ctypes.c_void_p(a2.data_ptr()), dtype_code(a2.dtype), a2.stride(0) if a2.shape[0] > 1 else a2.shape[1],
         a2.shape[0], K, ptr(w_op), w_op.shape[0], w_op.shape[1], None if bias is None else ptr(bias), N, ptr(out),
         dtype_code(out.dtype), out.stride(0) if out.shape[0] > 1 else out.shape[1], ncols, 1 if accumulate else 0,
         None if a_copy is None else ptr(a_copy),
         0 if a_copy is None else a_copy.shape[1], None if live is None else ptr(live[0]), 0 if live is None else int(live[1]),
         stream(out.device))


_ONE = {}


def _one(device):
    """a float32 [1] tensor holding 1.0 on `device` (the scale operand of the residual kernels); kept per device, but
    never one that was allocated inside a stream capture (it would live in that graph's memory pool)"""
    t = _ONE.get(device)
    if t is None:
        t = torch.ones(1, dtype=torch.float32, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _ONE[device] = t
    return t


class GeluLink:
    """What ties the two projections of y = fc2(gelu(fc1(x))) (msft's MLP, msft_arch.py:19-33) together when the
    activation lives in their epilogues: fc1's kernel writes gelu(h) as its result and h on the side (kept here for
    the backward), and fc2's input-gradient kernel multiplies dy W2 by gelu'(h) on its way out, so that what
    reaches fc1's backward is dh already.  `done` records the backward pass in which fc2's kernel did that; in a
    pass where it did not (another path computed fc2's input gradient), fc1's backward applies the slope itself."""

    def __init__(self):
        self.h = None
        self.done = None


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b for the projections of the attention modules.

    Under bf16 autocast on the GPU the product and the input gradient run in the library's own
    MFMA kernel (dsw_linear), which reads the activations where they lie - the fp32 residual stream,
    a gradient laid out by autograd, padded bf16 rows from the attention kernels - rounds them to bf16
    on the way in, writes the bf16 copy the weight gradient needs on the side, and adds the bias / casts
    the result in its accumulators: no separate cast or padding passes.  fp32 runs and shapes it does
    not take go through the stock GEMM library.  Either way:

    * operands and results live in rows padded to multiples of 64 elements wherever that costs no
      extra pass (the library's kernels for 180 / 540 wide rows run 2-4x slower than for 192 / 576):
      padded inputs are taken with their padding, outputs keep their padded width (the caller sees a
      view of the first n columns);
    * the weight gradient dY^T X has a tiny output and a huge reduction dimension (all tokens), which
      the library runs on a handful of CUs; it is issued as a batched GEMM over token chunks and the
      chunk results are summed in fp32 (a split-K the library does not pick by itself)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act_dtype, pad_out, live=None, resid=None, pending_out=None, pending_in=None, gelu=None, drop=None):
        # drop = (p, seed), with resid: the result is resid + dropout(x W^T + b), the residual dropout of the caller's block
        # (seeded decisions as in rezero_residual); the backward masks the incoming gradient as it loads it.
        # pending_out may also be a GradSink: the residual's share of the gradient is then delivered to it
        # gelu = ("out", GeluLink): the result is gelu(x W^T + b) (the caller has checked that the kernel takes the call);
        # gelu = ("in", GeluLink): x is such a result, and the input gradient leaves as the gradient of the link's h
        # resid (fp32, the result's shape): the result is resid + (x W^T + b) as fp32, one kernel (the caller has
        # checked that the kernel takes the call); its gradient share, dy itself, goes to pending_out when given.
        # pending_in: a PendingGrad whose gradient the input-gradient kernel adds to dx as it writes it.
        N, Kd = weight.shape
        x2 = x.reshape(-1, Kd)
        T = x2.shape[0]
        mfma = (act_dtype == torch.bfloat16 and x2.is_cuda and T > 0 and N % 4 == 0 and weight.is_contiguous()
                and weight.dtype in (torch.float32, torch.bfloat16) and weight.data_ptr() % 16 == 0)
        # only the stock-GEMM branch calls ops that autocast would intercept
        with (_NULL_CTX if mfma else torch.autocast("cuda", enabled=False)):
            xk = _padded_rows(x2, act_dtype)
            PAD_STATS["taken" if xk is not None else "copied"] += 1
            Np, Kp = _ceil_pad(N), _ceil_pad(Kd)
            Nout = Np if pad_out else N
            if mfma:
                b32 = None if bias is None else bias.detach().float().contiguous()
                w_p, w_pt = _weight_operands(weight)
                a_copy = None
                if xk is not None:
                    a2, K = xk, Kp
                elif x2.dtype == act_dtype and Kd == Kp and x2.is_contiguous() and x2.data_ptr() % 16 == 0:
                    a2, K, xk = x2, Kd, x2
                elif _mfma_operand(x2) is not None:
                    a2, K = x2, Kd
                    if ctx.needs_input_grad[1]:
                        a_copy = xk = _mark_padded(torch.empty(T, Kp, dtype=act_dtype, device=x2.device))
                else:
                    a2 = xk = _cast_padded(x2, act_dtype)
                    K = Kp
                if resid is not None:
                    assert a_copy is None and live is None
                    y_p = torch.empty(T, N, dtype=torch.float32, device=x2.device)
                    if drop is None:
                        call("dsw_linear_residual", a2.data_ptr(), dtype_code(a2.dtype), a2.stride(0) if T > 1 else a2.shape[1], T, K,
                             ptr(w_p), w_p.shape[0], w_p.shape[1], None if b32 is None else ptr(b32), N, ptr(resid), N, ptr(_one(x2.device)),
                             ptr(y_p), N, N, stream(x2.device))
                    else:
                        call("dsw_linear_residual_dropout", a2.data_ptr(), dtype_code(a2.dtype), a2.stride(0) if T > 1 else a2.shape[1], T, K,
                             ptr(w_p), w_p.shape[0], w_p.shape[1], None if b32 is None else ptr(b32), N, ptr(resid), N, ptr(_one(x2.device)),
                             ptr(y_p), N, N, _c_float(drop[0]), ctypes.c_ulonglong(drop[1]), stream(x2.device))
                elif gelu is not None and gelu[0] == "out":
                    assert live is None and Nout % 8 == 0
                    y_p = torch.empty(T, Nout, dtype=act_dtype, device=x2.device)
                    keep_h = any(ctx.needs_input_grad[:3])
                    h = _mark_padded(torch.empty(T, Nout, dtype=act_dtype, device=x2.device)) if keep_h else None
                    call("dsw_linear_gelu", ctypes.c_void_p(a2.data_ptr()), dtype_code(a2.dtype), a2.stride(0) if T > 1 else a2.shape[1], T, K,
                         ptr(w_p), w_p.shape[0], w_p.shape[1], None if b32 is None else ptr(b32), N, ptr(y_p), Nout, Nout,
                         None if h is None else ptr(h), Nout, None if a_copy is None else ptr(a_copy),
                         0 if a_copy is None else a_copy.shape[1], stream(x2.device))
                    gelu[1].h = h
                else:
                    y_p = torch.empty(T, Nout, dtype=act_dtype, device=x2.device)
                    _mfma_linear(a2, K, w_p, b32, N, y_p, Nout, a_copy, live=live)
                if xk is None:
                    xk = x2  # no weight gradient wanted: only the shape is read
                w_saved = w_pt
            else:
                if xk is None:
                    xk = _cast_padded(x2, act_dtype) if x2.dtype != act_dtype else x2.contiguous()
                Kx = xk.shape[1]
                w_p = _padded_matrix(weight, act_dtype, Np, Kp)
                if bias is not None:
                    b_p = _padded_matrix(bias.view(1, N), act_dtype, 1, Nout).view(Nout)
                    y_p = torch.addmm(b_p, xk, w_p[:Nout, :Kx].t())
                else:
                    y_p = xk @ w_p[:Nout, :Kx].t()
                w_saved = w_p
            if resid is None:
                _mark_padded(y_p)  # padding columns = zero weight rows x finite inputs + zero bias
        ctx.save_for_backward(xk, w_saved)
        ctx.meta = (x.shape, x.dtype, weight.dtype, None if bias is None else bias.dtype, N, Kd, mfma)
        ctx.live = live if mfma else None
        ctx.pending = (pending_out, pending_in)
        ctx.gelu = gelu if mfma else None
        ctx.drop = drop
        assert (gelu is None and drop is None) or mfma
        return y_p[:, :N].view(x.shape[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        xk, w_s = ctx.saved_tensors
        xshape, xdt, wdt, bdt, N, Kd, mfma = ctx.meta
        act = w_s.dtype
        Np, Kp = _ceil_pad(N), _ceil_pad(Kd)
        need_dx = ctx.needs_input_grad[0]
        need_dw = ctx.needs_input_grad[1] or (bdt is not None and ctx.needs_input_grad[2])
        gelu = ctx.gelu
        if gelu is not None and gelu[0] == "out":
            link = gelu[1]
            if link.done != torch._C._current_graph_task_id():
                # the gradient of gelu(h) arrived, not that of h: the slope is applied here
                h = link.h[:, :N]
                d2, ldg = _row_view4(dy.to(h.dtype), N)
                dh = _mark_padded(torch.empty_like(link.h))
                call("dsw_gelu_bwd", _dptr(d2), ldg, ptr(link.h), dtype_code(h.dtype), link.h.shape[1], N, h.shape[0], ptr(dh), dh.shape[1],
                     stream(dh.device))
                dy = dh[:, :N].view(dy.shape)
            link.done = None
        drop = ctx.drop
        with torch.autocast("cuda", enabled=False):
            dy2 = dy.reshape(-1, N)
            T = dy2.shape[0]
            if drop is not None:
                # the gradient of resid + dropout(branch) arrives unmasked: the input-gradient kernel below masks it on load
                # (and writes the masked bf16 copy that the weight and bias gradients are taken from)
                if dy2.dtype != torch.float32 or not dy2.is_contiguous():
                    dy2 = dy2.float().contiguous()
                if not (mfma and need_dx and xdt == act and act == torch.bfloat16 and _mfma_operand(dy2) is not None and ctx.live is None):
                    raise RuntimeError("projection with the residual dropout inside: its backward needs the kernel path")
            dyk = _padded_rows(dy2, act) if drop is None else None
            PAD_STATS["taken" if dyk is not None else "copied"] += 1
            dx = None
            if mfma and need_dx and xdt in (torch.float32, torch.bfloat16):
                # dx = dy W: the same kernel on the transposed weight [Kp, Np]
                a_copy = None
                if dyk is not None:
                    a2, K = dyk, Np
                elif dy2.dtype == act and N == Np and dy2.is_contiguous() and dy2.data_ptr() % 16 == 0:
                    a2, K, dyk = dy2, N, dy2
                else:
                    if dy2.stride(1) != 1:
                        dy2 = dy2.contiguous()
                    if _mfma_operand(dy2) is not None:
                        a2, K = dy2, N
                        if need_dw or drop is not None:
                            a_copy = dyk = _mark_padded(torch.empty(T, Np, dtype=act, device=dy2.device))
                    else:
                        a2 = dyk = _cast_padded(dy2, act)
                        K = Np
                if xdt == act:
                    dx_p = _mark_padded(torch.empty(T, Kp, dtype=act, device=dy2.device))
                    link = gelu[1] if gelu is not None and gelu[0] == "in" else None
                    if link is not None and link.h is not None and link.h.shape == (T, Kp) and ctx.live is None and act == torch.bfloat16:
                        # dh = (dy W) * gelu'(h) in the kernel's epilogue
                        call("dsw_linear_gelu_bwd", ctypes.c_void_p(a2.data_ptr()), dtype_code(a2.dtype), a2.stride(0) if T > 1 else a2.shape[1],
                             T, K, ptr(w_s), w_s.shape[0], w_s.shape[1], ptr(link.h), Kp, ptr(dx_p), Kp, Kp,
                             None if a_copy is None else ptr(a_copy), 0 if a_copy is None else a_copy.shape[1],
                             _c_float(0.0 if drop is None else drop[0]), ctypes.c_ulonglong(0 if drop is None else drop[1]), stream(dy2.device))
                        link.done = torch._C._current_graph_task_id()
                    elif drop is not None:
                        call("dsw_linear_dropout_in", ptr(a2), a2.stride(0) if T > 1 else a2.shape[1], T, K, ptr(w_s), w_s.shape[0], w_s.shape[1],
                             ptr(dx_p), dtype_code(dx_p.dtype), Kp, Kp, None if a_copy is None else ptr(a_copy),
                             0 if a_copy is None else a_copy.shape[1], _c_float(drop[0]), ctypes.c_ulonglong(drop[1]), stream(dy2.device))
                    else:
                        _mfma_linear(a2, K, w_s, None, 0, dx_p, Kp, a_copy, live=ctx.live)
                    dx = dx_p[:, :Kd].view(xshape) if Kp != Kd else dx_p.view(xshape)
                else:
                    dx = torch.empty(T, Kd, dtype=xdt, device=dy2.device)
                    extra = ctx.pending[1].take() if ctx.pending[1] is not None else None
                    e2 = _row_view_f32(extra) if extra is not None and xdt == torch.float32 else None
                    if e2 is not None and a_copy is None and ctx.live is None and e2.shape == (T, Kd):
                        # dx = extra + dy W: the gradient that reaches x by another way (the residual around the module,
                        # a concatenation that x is part of), read in the epilogue where it lies - a dense map or the
                        # leading columns of a wider one - instead of being added by a pass of its own
                        call("dsw_linear_residual", a2.data_ptr(), dtype_code(a2.dtype), a2.stride(0) if T > 1 else a2.shape[1], T, K,
                             ptr(w_s), w_s.shape[0], w_s.shape[1], None, 0, _dptr(e2), e2.stride(0) if T > 1 else Kd, ptr(_one(dy2.device)),
                             ptr(dx), Kd, Kd, stream(dy2.device))
                    else:
                        _mfma_linear(a2, K, w_s, None, 0, dx, Kd, a_copy, live=ctx.live)
                        if extra is not None:
                            dx.add_(extra.reshape(dx.shape))
                    dx = dx.view(xshape)
            if dyk is None and (need_dw or (need_dx and dx is None)):
                dyk = _cast_padded(dy2, act) if dy2.dtype != act else dy2.contiguous()
            if need_dx and dx is None:
                w_p = w_s.t() if mfma else w_s       # [Np, Kp] either way
                dx = _mark_padded(dyk @ w_p[:dyk.shape[1]]).as_strided((T, Kd), (w_p.shape[1], 1))
                if xdt != act:
                    dx = cast_rows(dx, xdt)
                dx = dx.view(xshape)
            dw = db = None
            want_db = bdt is not None and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                Ndy, Kx = dyk.shape[1], xk.shape[1]
                if (mfma and Kd % 4 == 0 and Ndy == Np and Kx == Kp and xk.dtype == act and dyk.is_contiguous()
                        and xk.is_contiguous() and dyk.data_ptr() % 16 == 0 and xk.data_ptr() % 16 == 0):
                    # own split-K kernel; the bias gradient rides in x's first padding column when there is one
                    dw = torch.empty(N, Kd, dtype=torch.float32, device=dyk.device)
                    fused_db = want_db and Kd < Kp
                    dbf = torch.empty(N, dtype=torch.float32, device=dyk.device) if fused_db else None
                    ws = torch.empty(int(_lib.load().dsw_linear_wgrad_workspace(T, N, Kd)), dtype=torch.float32, device=dyk.device)
                    live = ctx.live
                    call("dsw_linear_wgrad", ptr(dyk), Np, ptr(xk), Kp, T, N, Kd, ptr(dw), None if dbf is None else ptr(dbf), ptr(ws),
                         None if live is None else ptr(live[0]), 0 if live is None else int(live[1]), stream(dyk.device))
                    dw = dw.to(wdt)
                    if fused_db:
                        db = dbf.to(bdt)
                elif T >= 8 * _WGRAD_CHUNK and T % _WGRAD_CHUNK == 0:
                    S = T // _WGRAD_CHUNK
                    part = torch.bmm(dyk.view(S, _WGRAD_CHUNK, Ndy).transpose(1, 2), xk.view(S, _WGRAD_CHUNK, Kx))
                    dw = part.sum(0, dtype=torch.float32)[:N, :Kd].to(wdt)
                else:
                    dw = (dyk.t() @ xk).float()[:N, :Kd].to(wdt)
            if want_db and db is None:
                db = colsum(dyk)[:N].to(bdt)
            # a residual's gradient that no kernel above read (a path without the epilogue, or no dx wanted here)
            extra = ctx.pending[1].take() if ctx.pending[1] is not None else None
            if extra is not None and dx is not None:
                dx = dx + extra.reshape(dx.shape).to(dx.dtype)
        d_resid = None
        if ctx.needs_input_grad[6]:
            out = ctx.pending[0]
            if isinstance(out, GradSink):
                # as the residual op does it: dy itself is the residual's share; the norm's backward kernel adds it to its own
                out._same_pass()
                if out.buf is not None:
                    out.buf.view(dy.shape).add_(dy)
                    d_resid = out.deliver(out.buf)
                else:
                    d_resid = out.deliver(dy if dy.dtype == torch.float32 and dy.is_contiguous() else dy.float().contiguous())
            elif out is not None:
                out.put(dy)
            else:
                d_resid = dy
        return dx, dw, db, None, None, None, d_resid, None, None, None, None


def linear_plus_residual(a, lin, resid, pending=None, dropout=None, dropout_seed=None, gelu=None):
    """resid + lin(a) as fp32 in one kernel, with autograd (dsw_linear_residual with a scale of one: the projection that
    closes a branch with the plain residual around it, DSwinT_arch.py:337 with :355).  bf16 autocast, `a` in padded
    bf16 rows (what the attention kernels write), resid a contiguous fp32 tensor of the result's shape.  The backward
    reads the fp32 gradient as it arrives; the residual's share of it goes to `pending` (a PendingGrad, or the GradSink
    shared with the norm in front of the branch) when given, back to autograd otherwise.  dropout: the nn.Dropout the
    caller's block puts on the branch (msft_arch.py:163-165); in training it is applied inside the kernel in front of
    the residual (dsw_linear_residual_dropout, the seeded decisions of rezero_residual) and the backward masks the
    gradient as the input-gradient kernel loads it (dsw_linear_dropout_in): no pass over the stream for the residual
    or its dropout, forward or backward.  gelu: see mlp_gelu.  None when the call is not one the kernel takes."""
    weight = lin.weight
    N, Kd = weight.shape
    if not (autocast_dtype(None) == torch.bfloat16 and a.is_cuda and resid.is_cuda and resid.dtype == torch.float32
            and resid.is_contiguous() and resid.shape[-1] == N and N % 4 == 0 and a.shape[:-1] == resid.shape[:-1]
            and a.numel() > 0 and weight.is_contiguous() and weight.dtype in (torch.float32, torch.bfloat16)
            and weight.data_ptr() % 16 == 0 and resid.data_ptr() % 16 == 0):
        return None
    a2 = a.reshape(-1, Kd)
    if _padded_rows(a2, torch.bfloat16) is None and not (a2.dtype == torch.bfloat16 and Kd == _ceil_pad(Kd) and a2.is_contiguous()
                                                            and a2.data_ptr() % 16 == 0):
        return None      # (padded bf16 rows, or dense ones of a width that needs no padding)
    drop = None
    if dropout is not None and dropout.training and dropout.p > 0:
        if dropout.p >= 1.0 or not torch.is_grad_enabled():
            return None
        drop = (float(dropout.p), draw_dropout_seed() if dropout_seed is None else int(dropout_seed))
    return _run(_LinearFn, a, weight, lin.bias, torch.bfloat16, False, None, resid, pending, None, gelu, drop)


def linear(x, lin, pad_out=False, live=None, grad_extra=None):
    """`lin(x)` for an nn.Linear `lin`, honouring CUDA autocast like F.linear does.  pad_out: leave
    the result in rows padded to a multiple of 64 elements (returned as a view of the first
    out_features columns); for callers whose consumers accept such views.
    live = (counter, unit_rows): an int32 counter on the device; the rows at and beyond counter * unit_rows are
    switched off (DS-MSA: the window slots behind the ones that passed the gate) - under bf16 autocast the kernels do
    no work for them, forward or backward, and leave zeros in their place; other paths compute them like any row.
    grad_extra: a PendingGrad; what it holds when the backward pass gets here is added to x's gradient by the kernel
    that writes it."""
    act = autocast_dtype(x.dtype)
    return _run(_LinearFn, x, lin.weight, lin.bias, act, bool(pad_out), live, None, None, grad_extra, None, None)


FUSE_GELU = True   # False: mlp_gelu always declines
FUSE_RESIDUAL_DROPOUT = True   # False: the msft blocks keep their residual and its dropout as a pass of its own (rezero_residual)


def mlp_gelu(x, fc1, fc2, residual=None):
    """fc2(gelu(fc1(x))) of msft's MLP (msft_arch.py:19-33) with the activation inside the projection kernels: fc1's
    epilogue writes gelu(h) (and h on the side when a backward will want it), fc2's input-gradient kernel applies
    gelu'(h) as it writes dh - no pass over the hidden rows for the activation, forward or backward.  bf16 autocast
    only; None when the call is not one the kernels take (the caller composes linear / gelu / linear).
    residual = (stream, dropout module or None, GradSink / PendingGrad or None): returns stream + dropout(mlp(x)), the
    residual of the caller's block (msft_arch.py:165), taken inside fc2's kernel where that is possible."""
    w1, w2 = fc1.weight, fc2.weight
    Hd, Kd = w1.shape
    ok = (FUSE_GELU and autocast_dtype(None) == torch.bfloat16 and x.is_cuda and x.numel() > 0 and x.dtype in (torch.float32, torch.bfloat16)
          and Hd % 4 == 0 and w2.shape[1] == Hd and w2.shape[0] % 4 == 0
          and all(w.is_contiguous() and w.dtype in (torch.float32, torch.bfloat16) and w.data_ptr() % 16 == 0 for w in (w1, w2)))
    if not ok:
        return None
    link = GeluLink()
    g = _run(_LinearFn, x, w1, fc1.bias, torch.bfloat16, True, None, None, None, None, ("out", link), None)
    if residual is not None:
        # the block's residual and its dropout inside fc2's kernel (and the mask inside its input-gradient kernel)
        y = linear_plus_residual(g, fc2, residual[0], residual[2], dropout=residual[1], gelu=("in", link))
        if y is not None:
            return y
        branch = _run(_LinearFn, g, w2, fc2.bias, torch.bfloat16, False, None, None, None, None, ("in", link), None)
        return rezero_residual(residual[0], _one(x.device), branch, sink=residual[2] if isinstance(residual[2], GradSink) else None,
                               dropout=residual[1])
    return _run(_LinearFn, g, w2, fc2.bias, torch.bfloat16, False, None, None, None, None, ("in", link), None)


def linear_residual(a, lin, resid, r):
    """resid + r * lin(a) as fp32, in one kernel (dsw_linear_residual): the projection that closes a branch of
    the reference's ViTBlock fused with the ReZero residual around it (DSwinT_arch.py:337 / :69 with
    :355-357).  Inference under bf16 autocast only; returns None when the call is not one the kernel takes
    (autograd on, other dtypes or layouts), and the caller then composes linear() and rezero_residual()."""
    if torch.is_grad_enabled() or autocast_dtype(None) != torch.bfloat16:
        return None
    weight, bias = lin.weight, lin.bias
    N, Kd = weight.shape
    if not (a.is_cuda and resid.is_cuda and resid.dtype == torch.float32 and resid.is_contiguous() and resid.shape[-1] == N
            and N % 4 == 0 and r.numel() == 1 and a.shape[:-1] == resid.shape[:-1] and weight.is_contiguous()
            and weight.dtype in (torch.float32, torch.bfloat16) and weight.data_ptr() % 16 == 0 and resid.data_ptr() % 16 == 0):
        return None
    a2 = a.reshape(-1, Kd)
    T = a2.shape[0]
    if T == 0:
        return None
    xk = _padded_rows(a2, torch.bfloat16)
    if xk is not None:
        a2, K_ = xk, _ceil_pad(Kd)
    elif _mfma_operand(a2) is not None:
        K_ = Kd
    else:
        return None
    w_p, _ = _weight_operands(weight)
    b32 = None if bias is None else bias.detach().float().contiguous()
    r32 = r.detach().float().reshape(1).contiguous()
    out = torch.empty(T, N, dtype=torch.float32, device=a.device)
    call("dsw_linear_residual", a2.data_ptr(), dtype_code(a2.dtype), a2.stride(0) if T > 1 else a2.shape[1], T, K_, ptr(w_p),
         w_p.shape[0], w_p.shape[1], None if b32 is None else ptr(b32), N, ptr(resid), N, ptr(r32), ptr(out), N, N, stream(a.device))
    return out.view(resid.shape)


FUSE_GATED_SILU = True   # False: linear_gated_silu always declines


def interleave_gate_content(gate_w, content_w, gate_b, content_b):
    """The operand layout of dsw_linear_gated_silu: the rows of the two fc1 weights [H, K] in blocks of 32 -
    gate rows 32c..32c+31, then the content rows of the same hidden columns - zero rows past H; the biases
    alike.  -> (weight [2*ceil32(H), K], bias [2*ceil32(H)])"""
    H, Kd = gate_w.shape
    Hp = (H + 31) // 32 * 32
    w = gate_w.new_zeros(Hp // 32, 2, 32, Kd)
    b = gate_w.new_zeros(Hp // 32, 2, 32, dtype=torch.float32)
    idx = torch.arange(H, device=gate_w.device)
    w[idx // 32, 0, idx % 32] = gate_w.detach()
    w[idx // 32, 1, idx % 32] = content_w.detach()
    if gate_b is not None:
        b[idx // 32, 0, idx % 32] = gate_b.detach().float()
        b[idx // 32, 1, idx % 32] = content_b.detach().float()
    return w.reshape(2 * Hp, Kd), b.reshape(2 * Hp)


FOLD_RMSNORM = True   # False: linear_prenorm declines and linear_gated_silu declines a prenorm (the norm runs as a pass of its own)
_FOLD_CACHE = {}


def _folded_weight(weight, norm_weight):
    """weight [N, K] with the RMSNorm weight [K] multiplied into its columns (fp32), kept until either changes"""
    key = (id(weight), id(norm_weight))
    e = _FOLD_CACHE.get(key)
    if (e is not None and e[0]() is weight and e[1]() is norm_weight and e[2] == (weight._version, norm_weight._version,
                                                                                weight.data_ptr(), norm_weight.data_ptr())):
        return e[3]
    f = (weight.detach().float() * norm_weight.detach().float()[None, :]).contiguous()
    if not torch.cuda.is_current_stream_capturing():
        if len(_FOLD_CACHE) > 4096:
            _FOLD_CACHE.clear()
        _FOLD_CACHE[key] = (weakref.ref(weight), weakref.ref(norm_weight),
                            (weight._version, norm_weight._version, weight.data_ptr(), norm_weight.data_ptr()), f)
    return f


FUSE_MLP = True   # False: mlp_gated_residual always declines (the MLP runs as two projection kernels)
MLP_PREP_KERNEL = True   # False: mlp_gated_operands builds the images with tensor expressions (tests)


def mlp_gated_operands(gate_w, content_w, gate_b, content_b, fc2_w, fc2_b, norm_weight):
    """The operand images of dsw_mlp_gated_residual: (w1c [chunks, 64, 192] bf16, b1c [chunks, 64] fp32,
    w2c [chunks, 192, 32] bf16, b2 [192] fp32, chunks) from the two fc1 weights [H, C] (with the RMSNorm weight [C]
    multiplied into their columns), fc2's weight [C_out, H] and the biases (None = zero)."""
    H, C = gate_w.shape
    N = fc2_w.shape[0]
    dev = gate_w.device
    nch = (H + 31) // 32
    Hp = nch * 32
    ps = (gate_w, content_w, gate_b, content_b, fc2_w, fc2_b, norm_weight)
    if MLP_PREP_KERNEL and gate_w.is_cuda and all(p is None or (p.dtype == torch.float32 and p.is_contiguous()) for p in ps):
        # one launch (dsw_mlp_gated_prep); the tensor expressions below are the same function for other dtypes
        w1c = torch.empty(nch, 64, 192, dtype=torch.bfloat16, device=dev)
        b1c = torch.empty(nch, 64, dtype=torch.float32, device=dev)
        w2c = torch.empty(nch, 192, 32, dtype=torch.bfloat16, device=dev)
        b2 = torch.empty(192, dtype=torch.float32, device=dev)
        call("dsw_mlp_gated_prep", *(None if p is None else ptr(p.detach()) for p in ps), H, C, N, ptr(w1c), ptr(b1c), ptr(w2c), ptr(b2),
             stream(dev))
        return w1c, b1c, w2c, b2, nch
    g = norm_weight.detach().float()[None, :] if norm_weight is not None else 1.0
    w1 = torch.zeros(nch, 2, 32, 192, device=dev)
    b1 = torch.zeros(nch, 2, 32, device=dev)
    for u, (w, b) in enumerate(((gate_w, gate_b), (content_w, content_b))):
        wp = torch.zeros(Hp, 192, device=dev)
        wp[:H, :C] = w.detach().float() * g
        w1[:, u] = wp.view(nch, 32, 192)
        if b is not None:
            bp = torch.zeros(Hp, device=dev)
            bp[:H] = b.detach().float()
            b1[:, u] = bp.view(nch, 32)
    w2 = torch.zeros(192, Hp, device=dev)
    w2[:N, :H] = fc2_w.detach().float()
    # every 16 hidden units in the order 0-3, 8-11, 4-7, 12-15: as views (an index tensor would be a host copy, which a
    # stream capture does not take)
    w2 = w2.view(192, Hp // 16, 2, 2, 4).transpose(2, 3).reshape(192, nch, 32).permute(1, 0, 2)
    b2 = torch.zeros(192, device=dev)
    if fc2_b is not None:
        b2[:N] = fc2_b.detach().float()
    return (w1.view(nch, 64, 192).to(torch.bfloat16).contiguous(), b1.view(nch, 64).contiguous(),
            w2.to(torch.bfloat16).contiguous(), b2, nch)


def mlp_gated_ok(x, norm, hidden, out_features):
    """whether mlp_gated_residual takes the call: inference under bf16 autocast on the contiguous fp32 stream"""
    C = x.shape[-1]
    return (FUSE_MLP and _prenorm_ok(x, norm, C) and x.dtype == torch.float32 and x.is_contiguous() and C % 4 == 0
            and out_features == C and 0 < hidden <= 512)


def mlp_gated_residual(x, norm, operands, r):
    """x + r * mlp(norm(x)) for the SiLU-gated MLP (DSwinT_arch.py:47-69 with :345 and :357) in one kernel
    (dsw_mlp_gated_residual): x is read once and the result written once, the hidden activations stay in registers.
    operands from mlp_gated_operands (with norm.weight folded in); r a one-element tensor; the caller has checked
    mlp_gated_ok."""
    C = x.shape[-1]
    w1c, b1c, w2c, b2, nch = operands
    x2 = x.view(-1, C)
    y = torch.empty_like(x2)
    eps = norm.eps if norm.eps is not None else torch.finfo(x.dtype).eps
    r32 = r.detach().float().reshape(1)
    call("dsw_mlp_gated_residual", ptr(x2), C, x2.shape[0], C, _c_float(eps), ptr(w1c), ptr(b1c), ptr(w2c), ptr(b2), nch, ptr(r32),
         ptr(y), C, stream(x.device))
    return y.view(x.shape)


def _prenorm_ok(x, norm, Kd):
    return (FOLD_RMSNORM and not torch.is_grad_enabled() and autocast_dtype(None) == torch.bfloat16 and _norm_kernel_ok(x, norm)
            and x.shape[-1] == Kd and Kd <= 192 and x.numel() > 0 and x.data_ptr() % 16 == 0)


def linear_prenorm(x, norm, lin, pad_out=False):
    """lin(norm(x)) for an nn.RMSNorm `norm` and an nn.Linear `lin` in one kernel (dsw_linear_rmsnorm): the rows of the
    fp32 stream are normalised as the projection loads them, the norm's weight sits in the prepared weight - no pass
    over the stream for the norm, no normalised copy.  Inference under bf16 autocast, K <= 192; None when the call
    is not one the kernel takes (the caller runs rms_norm() and linear())."""
    weight = lin.weight
    N, Kd = weight.shape
    if not (_prenorm_ok(x, norm, Kd) and N % 4 == 0 and weight.is_contiguous() and weight.dtype in (torch.float32, torch.bfloat16)):
        return None
    a2 = x.reshape(-1, Kd)
    T = a2.shape[0]
    eps = norm.eps if norm.eps is not None else torch.finfo(x.dtype).eps
    w_p, _ = _weight_operands(_folded_weight(weight, norm.weight))
    b32 = None if lin.bias is None else lin.bias.detach().float().contiguous()
    Nout = _ceil_pad(N) if pad_out else N
    out = _mark_padded(torch.empty(T, Nout, dtype=torch.bfloat16, device=x.device))
    call("dsw_linear_rmsnorm", ptr(a2), dtype_code(a2.dtype), Kd, T, Kd, _c_float(eps), ptr(w_p), w_p.shape[0], w_p.shape[1],
         None if b32 is None else ptr(b32), N, ptr(out), dtype_code(out.dtype), Nout, Nout, stream(x.device))
    return out[:, :N].view(x.shape[:-1] + (N,))


def linear_gated_silu(x, inter_w, inter_b, H, prenorm=None):
    """silu(fc1_gate(x)) * fc1_content(x) (DSwinT_arch.py:69) in the projection's own epilogue
    (dsw_linear_gated_silu): the [tokens, 2H] pre-activations are never written.  inter_w / inter_b from
    interleave_gate_content.  Inference under bf16 autocast only, K <= 192; returns the result as a view of the
    first H columns of zero-padded bf16 rows, or None when the call is not one the kernel takes (the caller then
    composes linear() and gated_silu()).  Same bits as that composition.  prenorm: an nn.RMSNorm to apply to x
    first, inside the same kernel (see linear_prenorm); None is returned when that is not possible."""
    if not FUSE_GATED_SILU or torch.is_grad_enabled() or autocast_dtype(None) != torch.bfloat16:
        return None
    N2, Kd = inter_w.shape
    if prenorm is not None:
        Hp = N2 // 2
        if not (_prenorm_ok(x, prenorm, Kd) and N2 % 64 == 0 and Hp == _ceil_pad(H) and inter_w.is_contiguous()
                and inter_w.dtype in (torch.float32, torch.bfloat16)):
            return None
        a2 = x.reshape(-1, Kd)
        T = a2.shape[0]
        eps = prenorm.eps if prenorm.eps is not None else torch.finfo(x.dtype).eps
        w_p, _ = _weight_operands(_folded_weight(inter_w, prenorm.weight))
        out = _mark_padded(torch.empty(T, Hp, dtype=torch.bfloat16, device=x.device))
        call("dsw_linear_rmsnorm_gated_silu", ptr(a2), dtype_code(a2.dtype), Kd, T, Kd, _c_float(eps), ptr(w_p), N2, w_p.shape[1],
             ptr(inter_b), ptr(out), Hp, stream(x.device))
        return out[:, :H].view(x.shape[:-1] + (H,))
    Hp = N2 // 2
    if not (x.is_cuda and Kd <= 192 and Kd % 4 == 0 and N2 % 64 == 0 and Hp == _ceil_pad(H) and inter_w.is_contiguous()
            and inter_w.dtype in (torch.float32, torch.bfloat16) and inter_w.data_ptr() % 16 == 0):
        return None
    a2 = x.reshape(-1, Kd)
    T = a2.shape[0]
    if T == 0:
        return None
    xk = _padded_rows(a2, torch.bfloat16)
    if xk is not None:
        a2, K_ = xk, _ceil_pad(Kd)
    elif _mfma_operand(a2) is not None:
        K_ = Kd
    else:
        return None
    w_p, _ = _weight_operands(inter_w)
    out = _mark_padded(torch.empty(T, Hp, dtype=torch.bfloat16, device=x.device))
    call("dsw_linear_gated_silu", a2.data_ptr(), dtype_code(a2.dtype), a2.stride(0) if T > 1 else a2.shape[1], T, K_, ptr(w_p),
         N2, w_p.shape[1], ptr(inter_b), ptr(out), Hp, stream(x.device))
    return out[:, :H].view(x.shape[:-1] + (H,))


# ---------------------------------------------------------------------------------------------
# RMSNorm in front of the attention modules / the MLP (DSwinT_arch.py:343-345, :355-357)
# ---------------------------------------------------------------------------------------------
class _RMSNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, eps, out_dtype, pad_out, sink=None):
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        M = x2.shape[0]
        ldo = _ceil_pad(C) if pad_out else C
        w32 = weight.detach().float().contiguous()
        out = _mark_padded(torch.empty(M, ldo, dtype=out_dtype, device=x.device))
        need_bwd = any(ctx.needs_input_grad[:2])
        rstd = torch.empty(M, dtype=torch.float32, device=x.device) if need_bwd else None
        call("dsw_rmsnorm_fwd", ptr(x2), ptr(w32), float(eps), M, C, ptr(out), dtype_code(out_dtype), ldo,
             None if rstd is None else ptr(rstd), stream(x.device))
        if need_bwd:
            ctx.save_for_backward(x2, w32, rstd)
        ctx.meta = (x.shape, weight.dtype, sink)
        return out[:, :C].view(x.shape) if ldo != C else out.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w32, rstd = ctx.saved_tensors
        xshape, wdt, sink = ctx.meta
        M, C = x2.shape
        dy2 = dy.reshape(-1, C)
        if dy2.dtype not in (torch.float32, torch.bfloat16) or dy2.stride(1) != 1 or (M > 1 and dy2.stride(0) % 4) or \
                dy2.data_ptr() % (4 * dy2.element_size()):
            dy2 = dy2.float().contiguous()
        if not ctx.needs_input_grad[0]:
            sink = None
        # the gradient that reached x around the branch (the residual's dy), if it is already there: the
        # kernel writes its sum with the norm's gradient into a fresh buffer (the other tensor is only read)
        if sink is not None:
            sink._same_pass()
        prev = sink.buf if sink is not None else None
        if prev is not None:
            prev = prev.reshape(M, C)
            if prev.dtype != torch.float32 or not prev.is_contiguous():
                prev = prev.float().contiguous()
        dx = torch.empty(xshape, dtype=torch.float32, device=x2.device)
        nblk = _lib.load().dsw_rmsnorm_blocks(M)
        partial = torch.empty(nblk, C, dtype=torch.float32, device=x2.device)
        call("dsw_rmsnorm_bwd", ctypes.c_void_p(dy2.data_ptr()), dtype_code(dy2.dtype), dy2.stride(0) if M > 1 else C, ptr(x2), ptr(w32),
             ptr(rstd), M, C, ptr(dx), None if prev is None else ptr(prev), ptr(partial), stream(x2.device))
        dw = colsum(partial).to(wdt) if ctx.needs_input_grad[1] else None
        return _dx_deliver(sink, dx), dw, None, None, None, None


def _norm_kernel_ok(x, norm):
    C = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and C % 4 == 0 and _ceil_pad(C) <= 256
            and norm.weight is not None and tuple(norm.normalized_shape) == (C,))


def rms_norm(x, norm, for_projection=False, sink=None, dense_act=False):
    """`norm(x)` for an nn.RMSNorm over the last dimension of the fp32 stream.  for_projection: the
    result feeds a projection - under bf16 autocast it is produced in bf16 rows padded to 64
    elements (what autocast's cast in front of the Linear would make of it, without the fp32 round
    trip) and returned as a view of the first C columns.  dense_act: under bf16 autocast the result is a dense
    bf16 tensor (for a consumer whose kernels round their input to bf16 anyway and want NHWC rows of C)."""
    if not _norm_kernel_ok(x, norm):
        assert sink is None, "a gradient sink needs the kernel path (see prenorm_sink)"
        return norm(x)
    eps = norm.eps if norm.eps is not None else torch.finfo(x.dtype).eps
    act = autocast_dtype(x.dtype)
    to_act = for_projection and act == torch.bfloat16
    if dense_act and not to_act and act == torch.bfloat16:
        return _run(_RMSNormFn, x, norm.weight, eps, act, False, sink)
    return _run(_RMSNormFn, x, norm.weight, eps, act if to_act else torch.float32, to_act, sink)


# ---------------------------------------------------------------------------------------------
# gated activation of the MLP (DSwinT_arch.py:47-69)
# ---------------------------------------------------------------------------------------------
class _GatedSiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, H, pad_out):
        h2, ldh = _row_view4(h, 2 * H)
        M = h2.shape[0]
        ldo = _ceil_pad(H) if pad_out else H
        out = _mark_padded(torch.empty(M, ldo, dtype=h.dtype, device=h.device))
        call("dsw_gated_silu_fwd", _dptr(h2), dtype_code(h2.dtype), ldh, H, M, ptr(out), dtype_code(out.dtype), ldo, stream(h.device))
        ctx.save_for_backward(h2)
        ctx.meta = (h.shape, H, ldh)
        return (out[:, :H] if ldo != H else out).view(h.shape[:-1] + (H,))

    @staticmethod
    def backward(ctx, dout):
        h2, = ctx.saved_tensors
        hshape, H, ldh = ctx.meta
        M = h2.shape[0]
        if dout.dtype != h2.dtype:
            dout = dout.to(h2.dtype)
        d2, ldg = _row_view4(dout, H)
        # the gradient keeps the row stride of h: padded rows (padding zeroed) go on to the projection's backward
        dh = _mark_padded(torch.empty(M, ldh, dtype=h2.dtype, device=h2.device))
        call("dsw_gated_silu_bwd", _dptr(d2), dtype_code(d2.dtype), ldg, _dptr(h2), dtype_code(h2.dtype), ldh, H, M, ptr(dh),
             stream(h2.device))
        return (dh[:, :2 * H] if ldh != 2 * H else dh).view(hshape), None, None


def gated_silu(h, pad_out=False):
    """silu(h[..., :H]) * h[..., H:] for h = [gate | content] (fp32 or bf16 on the GPU)"""
    H = h.shape[-1] // 2
    if not (h.is_cuda and h.dtype in (torch.float32, torch.bfloat16) and h.shape[-1] == 2 * H and H % 4 == 0):
        return torch.nn.functional.silu(h[..., :H]) * h[..., H:]
    return _run(_GatedSiluFn, h, H, bool(pad_out))


class _GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pad_out):
        C = x.shape[-1]
        x2, ldx = _row_view4(x, C)
        M = x2.shape[0]
        ldo = _ceil_pad(C) if pad_out else C
        out = _mark_padded(torch.empty(M, ldo, dtype=x.dtype, device=x.device))
        call("dsw_gelu_fwd", _dptr(x2), dtype_code(x2.dtype), ldx, C, M, ptr(out), ldo, stream(x.device))
        ctx.save_for_backward(x2)
        ctx.meta = (x.shape, ldx)
        return (out[:, :C] if ldo != C else out).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, = ctx.saved_tensors
        xshape, ldx = ctx.meta
        M, C = x2.shape
        if dy.dtype != x2.dtype:
            dy = dy.to(x2.dtype)
        d2, ldg = _row_view4(dy, C)
        # the gradient keeps the row stride of x: padded rows (padding zeroed) go on to the projection's backward
        dx = _mark_padded(torch.empty(M, ldx, dtype=x2.dtype, device=x2.device))
        call("dsw_gelu_bwd", _dptr(d2), ldg, _dptr(x2), dtype_code(x2.dtype), ldx, C, M, ptr(dx), ldx, stream(x2.device))
        return (dx[:, :C] if ldx != C else dx).view(xshape), None


def gelu(x, pad_out=False):
    """nn.GELU() (the erf form, msft_arch.py:25) on fp32 / bf16 rows, dense or views of zero-padded rows (what the
    projections return and take); with pad_out the result is such a view too"""
    if not (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.shape[-1] % 4 == 0 and x.numel() > 0):
        return torch.nn.functional.gelu(x)
    return _run(_GeluFn, x, bool(pad_out))


# ---------------------------------------------------------------------------------------------
# ReZero residual of the ViTBlock (DSwinT_arch.py:355-357)
# ---------------------------------------------------------------------------------------------
class _RezeroFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, branch, sink=None, mask=None, keep=1.0, seeded=None, pending=None):
        # pending (PendingGrad): x's share of the gradient, dy itself, is parked there instead of being returned
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        b2, ldb = _row_view4(branch, C)
        M = x2.shape[0]
        r32 = r.detach().float().reshape(1).contiguous()
        y = torch.empty(M, C, dtype=torch.float32, device=x.device)
        if seeded is not None:     # (p, seed): the keep decisions are a hash of the seed and the position, no mask tensor
            call("dsw_rezero_dropout_seeded_fwd", ptr(x2), _dptr(b2), dtype_code(b2.dtype), ldb, ptr(r32), _c_float(seeded[0]),
                 ctypes.c_ulonglong(seeded[1]), M, C, ptr(y), stream(x.device))
            ctx.save_for_backward(b2, r32)
        elif mask is None:
            call("dsw_rezero_fwd", ptr(x2), _dptr(b2), dtype_code(b2.dtype), ldb, ptr(r32), M, C, ptr(y), stream(x.device))
            ctx.save_for_backward(b2, r32)
        else:
            call("dsw_rezero_dropout_fwd", ptr(x2), _dptr(b2), dtype_code(b2.dtype), ldb, ptr(r32), ptr(mask), float(keep), M, C,
                 ptr(y), stream(x.device))
            ctx.save_for_backward(b2, r32, mask)
        ctx.meta = (x.shape, branch.shape, ldb, r.shape, r.dtype, sink, float(keep))
        ctx.seeded = seeded
        ctx.pending = pending
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        b2, r32, *rest = ctx.saved_tensors
        mask = rest[0] if rest else None
        xshape, bshape, ldb, rshape, rdt, sink, keep = ctx.meta
        M, C = b2.shape
        dy2 = dy.reshape(-1, C)
        if dy2.dtype != torch.float32 or not dy2.is_contiguous():
            dy2 = dy2.float().contiguous()
        db = dr = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            ld_db = ldb if ldb > C else C     # padded rows stay padded on the way back
            dbuf = _mark_padded(torch.empty(M, ld_db, dtype=b2.dtype, device=b2.device))
            partial = torch.empty(_lib.load().dsw_rezero_blocks(M, ld_db), dtype=torch.float32, device=b2.device)
            if ctx.seeded is not None:
                call("dsw_rezero_dropout_seeded_bwd", ptr(dy2), _dptr(b2), dtype_code(b2.dtype), ldb, ptr(r32), _c_float(ctx.seeded[0]),
                     ctypes.c_ulonglong(ctx.seeded[1]), M, C, ptr(dbuf), ld_db, ptr(partial), stream(b2.device))
            elif mask is None:
                call("dsw_rezero_bwd", ptr(dy2), _dptr(b2), dtype_code(b2.dtype), ldb, ptr(r32), M, C, ptr(dbuf), ld_db, ptr(partial),
                     stream(b2.device))
            else:
                call("dsw_rezero_dropout_bwd", ptr(dy2), _dptr(b2), dtype_code(b2.dtype), ldb, ptr(r32), ptr(mask), keep, M, C,
                     ptr(dbuf), ld_db, ptr(partial), stream(b2.device))
            db = (dbuf[:, :C] if ld_db != C else dbuf).view(bshape)
            dr = partial.sum().reshape(rshape).to(rdt) if ctx.needs_input_grad[1] else None
        if ctx.pending is not None and ctx.needs_input_grad[0]:
            assert sink is None
            ctx.pending.put(dy2.view(xshape))
            return None, dr, db, None, None, None, None, None
        if sink is None or not ctx.needs_input_grad[0]:
            return dy, dr, db, None, None, None, None, None
        # the residual's share of x's gradient is dy itself: the norm's backward reads it and writes the sum
        sink._same_pass()
        if sink.buf is not None:  # the norm's gradient came first: its buffer is ours to add into
            sink.buf.view(M, C).add_(dy2)
            return sink.deliver(sink.buf), dr, db, None, None, None, None, None
        return sink.deliver(dy2.view(xshape)), dr, db, None, None, None, None, None


def rezero_residual(x, r, branch, sink=None, dropout=None, dropout_seed=None, dropout_mask=None, pending=None):
    """x + r * dropout(branch) for the fp32 stream x [..., C], a one-element parameter r and a branch of the
    same shape in fp32 / bf16 (dense or a view of padded rows).  dropout: the nn.Dropout the reference puts on
    the branch (DSwinT_arch.py:352, :355-357) or None; in training it is applied inside the residual kernel
    (forward and backward), in place of a dropout pass, its mask pass and their backward over the branch.  The
    keep decisions are a hash of a 64-bit seed - dropout_seed, or drawn from torch's CPU generator (reproducible
    under torch.manual_seed, replayed by activation checkpointing) - and the element's position: no mask tensor is
    drawn, stored or read.  dropout_mask: an explicit uint8 mask [..., C] (0 = dropped) to use instead.
    pending: a PendingGrad that receives x's share of the gradient (dy itself) instead of autograd - the caller has
    made sure that a later backward kernel of the same pass takes it (kernel path only; not together with a sink)."""
    C = x.shape[-1]
    ok = (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and C % 4 == 0 and r.numel() == 1
          and branch.shape == x.shape and branch.dtype in (torch.float32, torch.bfloat16))
    if not ok:
        assert sink is None, "a gradient sink needs the kernel path (see prenorm_sink)"
        if pending is not None and torch.is_grad_enabled() and x.requires_grad:
            x = _ParkGradFn.apply(x, pending)
        return torch.addcmul(x, r, branch if dropout is None else dropout(branch))
    if dropout is None or not dropout.training or dropout.p == 0:
        return _run(_RezeroFn, x, r, branch, sink, None, 1.0, None, pending)
    p = float(dropout.p)
    if dropout_mask is None and p < 1.0:
        seed = draw_dropout_seed() if dropout_seed is None else int(dropout_seed)
        return _run(_RezeroFn, x, r, branch, sink, None, 1.0, (p, seed), pending)
    if dropout_mask is None:        # p = 1: everything dropped
        dropout_mask = torch.zeros(x.numel() // C, C, dtype=torch.uint8, device=x.device)
    mask = dropout_mask.reshape(-1, C).contiguous()
    return _run(_RezeroFn, x, r, branch, sink, mask, 1.0 / (1.0 - p) if p < 1.0 else 0.0, None, pending)


def prenorm_sink(x, norm, r):
    """For y = x + r * f(norm(x)): a GradSink through which the two gradients of x (the residual's dy and
    the norm's dx) meet in one buffer instead of being added by autograd in a pass of its own; None when
    no gradient is needed or either op would not run in its kernel."""
    C = x.shape[-1]
    if not (torch.is_grad_enabled() and x.requires_grad and _norm_kernel_ok(x, norm) and r.numel() == 1):
        return None
    return GradSink(2)


GRAD_MEETING = True   # False: the msft blocks leave the sums of their inputs' gradients to autograd (tests, comparisons)


def layer_norm_sink(x, norm):
    """prenorm_sink for y = x + dropout(f(layer_norm(x))) (msft_arch.py:163-165, the residual taken by rezero_residual
    with a scale of one): None when no gradient is needed or the norm would not run in its kernel"""
    if not (GRAD_MEETING and torch.is_grad_enabled() and x.requires_grad and _layer_norm_kernel_ok(x, norm)):
        return None
    return GradSink(2)


class _ConcatPendingFn(torch.autograd.Function):
    """cat([x, other], -1) whose backward parks x's share of the gradient - a view of the leading columns of the
    incoming map - in a PendingGrad instead of returning it: the kernel that writes another gradient of x later in
    the same pass (the input gradient of a projection of x that feeds `other`) reads it as an operand."""

    @staticmethod
    def forward(ctx, x, other, pending):
        ctx.cx = x.shape[-1]
        ctx.pending = pending
        return torch.cat([x, other], dim=-1)

    @staticmethod
    def backward(ctx, g):
        gx = g[..., :ctx.cx]
        if ctx.needs_input_grad[0]:
            ctx.pending.put(gx)
        return None, (g[..., ctx.cx:] if ctx.needs_input_grad[1] else None), None


def concat_pending(x, other, pending):
    """torch.cat([x, other], dim=-1); with a PendingGrad, x's gradient through the concatenation is handed to it (the
    caller has made sure that a projection of x with grad_extra=pending runs later in every backward pass: `other`
    must depend on that projection's result)."""
    if pending is None or not (torch.is_grad_enabled() and x.requires_grad and other.requires_grad):
        return torch.cat([x, other], dim=-1)
    return _ConcatPendingFn.apply(x, other, pending)


# ---------------------------------------------------------------------------------------------
# dense 3x3 convolution on NHWC (LastChannelConv2D, DSwinT_arch.py:36-44), inference
# ---------------------------------------------------------------------------------------------
def _taps_operand(w4):
    """[Co, Ci, 3, 3] fp32 tensor -> bf16 [ceil64(Co), 9 * ceil64(Ci)], tap-major, channels of every tap padded to 64"""
    Co, Ci = w4.shape[:2]
    Kp = _ceil_pad(Ci)
    taps = torch.zeros(Co, 9, Kp, dtype=torch.float32, device=w4.device)
    taps[:, :, :Ci] = w4.permute(0, 2, 3, 1).reshape(Co, 9, Ci)
    w_p = torch.empty(_ceil_pad(Co), 9 * Kp, dtype=torch.bfloat16, device=w4.device)
    call("dsw_weight_prep", ptr(taps), dtype_code(torch.float32), Co, 9 * Kp, ptr(w_p), None, stream(w4.device))
    return w_p


def _conv3x3_operand(weight):
    """Conv2d weight [Co, Ci, 3, 3] -> the kernel's operand; without autograd it is kept per parameter like
    the projection weights."""
    caching = _may_keep_operands()
    if caching:
        e = _PREP_CACHE.get(("conv3x3", id(weight)))
        if e is not None and e[0]() is weight and e[1] == weight._version and e[2] == weight.data_ptr():
            return e[3]
    w_p = _taps_operand(weight.detach().float())
    if caching:
        _PREP_CACHE[("conv3x3", id(weight))] = (weakref.ref(weight), weight._version, weight.data_ptr(), w_p)
    return w_p


def _conv3x3_call(x2, B, H, W, Ci, w_p, bias32, Co, out, x_copy):
    call("dsw_conv3x3_nhwc", _dptr(x2), dtype_code(x2.dtype), x2.stride(0), B, H, W, Ci, ptr(w_p), w_p.shape[0], w_p.shape[1],
         None if bias32 is None else ptr(bias32), Co, ptr(out), dtype_code(out.dtype), out.shape[-1], Co,
         None if x_copy is None else ptr(x_copy), 0 if x_copy is None else x_copy.shape[1], stream(out.device))


class _Conv3x3Fn(torch.autograd.Function):
    """3x3 convolution on NHWC under bf16 autocast: product, input gradient (the same kernel on the
    flipped, transposed weight) and weight / bias gradient (dsw_conv3x3_wgrad) on the matrix cores."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, resid=None):
        # resid (fp32, the result's shape): the result is resid + conv(x) as fp32, the skip connection of the residual
        # block this convolution closes (DSwinT_arch.py:373-374) added in the kernel's epilogue; its gradient is dy itself
        B, H, W, Ci = x.shape
        Co = weight.shape[0]
        M = B * H * W
        x2 = x.reshape(M, Ci)
        b32 = None if bias is None else bias.detach().float().contiguous()
        out = torch.empty(B, H, W, Co, dtype=act if resid is None else torch.float32, device=x.device)
        xk = _mark_padded(torch.empty(M, _ceil_pad(Ci), dtype=torch.bfloat16, device=x.device)) if ctx.needs_input_grad[1] else None
        if resid is None:
            _conv3x3_call(x2, B, H, W, Ci, _conv3x3_operand(weight), b32, Co, out, xk)
        else:
            w_p = _conv3x3_operand(weight)
            call("dsw_conv3x3_nhwc_residual_copy", _dptr(x2), dtype_code(x2.dtype), x2.stride(0), B, H, W, Ci, ptr(w_p), w_p.shape[0], w_p.shape[1],
                 None if b32 is None else ptr(b32), Co, ptr(resid), Co, ptr(out), Co, Co, None if xk is None else ptr(xk),
                 0 if xk is None else xk.shape[1], stream(out.device))
        ctx.save_for_backward(xk if xk is not None else x2, weight)
        ctx.meta = (x.shape, x.dtype, None if bias is None else bias.dtype)
        ctx.with_resid = resid is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        xk, weight = ctx.saved_tensors
        (B, H, W, Ci), xdt, bdt = ctx.meta
        Co = weight.shape[0]
        M = B * H * W
        dev = dy.device
        dy2 = dy.reshape(M, Co)
        if _mfma_operand(dy2) is None:
            dy2 = dy2.contiguous()
        need_w = ctx.needs_input_grad[1] or (bdt is not None and ctx.needs_input_grad[2])
        dyk = _padded_rows(dy2, torch.bfloat16)
        fresh = dyk is None and need_w
        if fresh:
            dyk = _mark_padded(torch.empty(M, _ceil_pad(Co), dtype=torch.bfloat16, device=dev))
        dx = None
        if ctx.needs_input_grad[0]:
            w_t = _taps_operand(weight.detach().float().flip(2, 3).transpose(0, 1))   # [Ci, Co, 3, 3], taps mirrored
            dx = torch.empty(B, H, W, Ci, dtype=xdt if xdt in (torch.float32, torch.bfloat16) else torch.float32, device=dev)
            _conv3x3_call(dy2, B, H, W, Co, w_t, None, Ci, dx, dyk if fresh else None)
            if dx.dtype != xdt:
                dx = dx.to(xdt)
        elif fresh:
            dyk = _cast_padded(dy2, torch.bfloat16)
        dw = db = None
        if need_w:
            Kp = xk.shape[1]
            dwt = torch.empty(Co, 9, Ci, dtype=torch.float32, device=dev)
            fused_db = bdt is not None and Ci < Kp
            dbf = torch.empty(Co, dtype=torch.float32, device=dev) if fused_db else None
            ws = torch.empty(int(_lib.load().dsw_conv3x3_wgrad_workspace(B, H, W, Co, Ci)), dtype=torch.float32, device=dev)
            call("dsw_conv3x3_wgrad", ptr(dyk), dyk.shape[1], ptr(xk), Kp, B, H, W, Co, Ci, ptr(dwt), None if dbf is None else ptr(dbf),
                 ptr(ws), stream(dev))
            dw = dwt.view(Co, 3, 3, Ci).permute(0, 3, 1, 2).contiguous().to(weight.dtype)
            if bdt is not None:
                db = (dbf if fused_db else colsum(dyk)[:Co]).to(bdt)
        return dx, dw, db, None, (dy if ctx.with_resid and ctx.needs_input_grad[4] else None)


def conv3x3_nhwc(x, conv):
    """`conv` (an nn.Conv2d with a 3x3 kernel, stride 1, padding 1, one group) applied to the NHWC tensor
    x [B, H, W, Ci] -> [B, H, W, Co], or None when this call is not one the kernels take (not bf16
    autocast, other geometry): the caller then uses the stock convolution."""
    act = autocast_dtype(x.dtype)
    B, H, W, Ci = x.shape
    ok = (x.is_cuda and act == torch.bfloat16 and x.dtype in (torch.float32, torch.bfloat16)
          and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
          and conv.groups == 1 and conv.padding_mode == "zeros" and Ci % 4 == 0 and conv.out_channels % 4 == 0
          and B * H * W * max(_ceil_pad(Ci), _ceil_pad(conv.out_channels)) * 4 < 2 ** 31)
    if not ok:
        return None
    return _run(_Conv3x3Fn, x.contiguous(), conv.weight, conv.bias, act, None)


FUSE_CONV_RESIDUAL_TRAINING = True   # False: with autograd the skip connection behind a closing convolution is an addition of its own
_CONV_VARIANTS = ("plain", "shuffle2", "pad_in4", "pad_out4")


def _conv3x3_variant(conv, variant):
    """(operand, fp32 bias, Co, Ci) of a transformed copy of a 3x3 Conv2d's parameters, kept per parameter like the
    plain operands: "shuffle2" orders the output channels (r1, r2, channel) instead of nn.PixelShuffle(2)'s
    channel*4 + r1*2 + r2, "pad_in4" / "pad_out4" add zero input / output channels up to a multiple of 4."""
    weight, bias = conv.weight, conv.bias
    caching = _may_keep_operands()
    key = ("conv3x3", variant, id(weight))
    sig = (weight._version, weight.data_ptr(), None if bias is None else bias._version, None if bias is None else bias.data_ptr())
    if caching:
        e = _PREP_CACHE.get(key)
        if e is not None and e[0]() is weight and e[1] == sig:
            return e[2]
    w = weight.detach().float()
    b = None if bias is None else bias.detach().float()
    if variant == "shuffle2":
        Co = w.shape[0]
        w = w.view(Co // 4, 4, *w.shape[1:]).transpose(0, 1).reshape(Co, *w.shape[1:])
        b = None if b is None else b.view(Co // 4, 4).t().reshape(Co)
    elif variant == "pad_in4":
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, -w.shape[1] % 4))
    elif variant == "pad_out4":
        extra = -w.shape[0] % 4
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, extra))
        b = None if b is None else torch.nn.functional.pad(b, (0, extra))
    w = w.contiguous()
    out = (_taps_operand(w), None if b is None else b.contiguous(), w.shape[0], w.shape[1])
    if caching:
        _PREP_CACHE[key] = (weakref.ref(weight), sig, out)
    return out


def conv3x3_nhwc_infer_ok(shape, dtype, conv, variant="plain"):
    """whether conv3x3_nhwc_infer takes a CUDA tensor of this NHWC shape and dtype (geometry of `conv`, channel
    counts after `variant`, sizes)"""
    B, H, W, Ci = shape
    Co = conv.out_channels + (-conv.out_channels % 4 if variant == "pad_out4" else 0)
    Cin = conv.in_channels + (-conv.in_channels % 4 if variant == "pad_in4" else 0)
    esize = 4 if dtype == torch.float32 else 2
    return (dtype in (torch.float32, torch.bfloat16) and Cin == Ci and Ci % 4 == 0 and Co % 4 == 0
            and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros" and (variant != "shuffle2" or conv.out_channels % 16 == 0)
            and 0 < B * H * W < 2 ** 31 - 256 and B * H * W * Ci * esize < 2 ** 31)


def conv3x3_nhwc_infer(x, conv, variant="plain", out_dtype=torch.bfloat16):
    """Without autograd: `conv` (3x3, stride 1, padding 1) applied to the NHWC tensor x [B, H, W, Ci] on the matrix
    cores (bf16 operands, fp32 accumulation) -> [B, H, W, Co] in out_dtype, with the parameters transformed as
    `variant` says (_conv3x3_variant)."""
    assert variant in _CONV_VARIANTS and not torch.is_grad_enabled()
    B, H, W, Ci = x.shape
    x = x.contiguous()
    w_p, b32, Co, Cin = _conv3x3_variant(conv, variant)
    out = torch.empty(B, H, W, Co, dtype=out_dtype, device=x.device)
    _conv3x3_call(x.view(B * H * W, Ci), B, H, W, Ci, w_p, b32, Co, out, None)
    return out


def conv3x3_nhwc_residual(x, conv, resid):
    """Under bf16 autocast: resid + conv(x) as fp32 in one kernel (dsw_conv3x3_nhwc_residual; with autograd
    dsw_conv3x3_nhwc_residual_copy through _Conv3x3Fn), the closing convolution of a residual block with the block's
    skip connection (DSwinT_arch.py:373-374, :501); None when the call is not one the kernel takes."""
    if autocast_dtype(None) != torch.bfloat16 or not x.is_cuda:
        return None
    B, H, W, Ci = x.shape
    if not (conv3x3_nhwc_infer_ok(x.shape, x.dtype, conv) and resid.is_cuda and resid.dtype == torch.float32 and resid.is_contiguous()
            and resid.shape == (B, H, W, conv.out_channels) and resid.data_ptr() % 16 == 0):
        return None
    if torch.is_grad_enabled():
        if not FUSE_CONV_RESIDUAL_TRAINING:
            return None
        return _run(_Conv3x3Fn, x.contiguous(), conv.weight, conv.bias, torch.bfloat16, resid)
    x = x.contiguous()
    w_p, b32, Co, _ = _conv3x3_variant(conv, "plain")
    out = torch.empty(B, H, W, Co, dtype=torch.float32, device=x.device)
    call("dsw_conv3x3_nhwc_residual", _dptr(x), dtype_code(x.dtype), Ci, B, H, W, Ci, ptr(w_p), w_p.shape[0], w_p.shape[1],
         None if b32 is None else ptr(b32), Co, ptr(resid), Co, ptr(out), Co, Co, stream(x.device))
    return out


class _Shuffle2GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, H, W, C4 = x.shape
        x = x.contiguous()
        y = torch.empty(B, 2 * H, 2 * W, C4 // 4, dtype=x.dtype, device=x.device)
        call("dsw_shuffle2_gelu_nhwc", ptr(x), dtype_code(x.dtype), B, H, W, C4 // 4, ptr(y), stream(x.device))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        B, H, W, C4 = x.shape
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        call("dsw_shuffle2_gelu_nhwc_bwd", ptr(dy), ptr(x), dtype_code(x.dtype), B, H, W, C4 // 4, ptr(dx), stream(x.device))
        return dx


def shuffle2_gelu_nhwc(x):
    """nn.PixelShuffle(2) followed by nn.GELU() on an NHWC tensor whose channels are ordered (r1, r2, channel):
    [B, H, W, 4c] -> [B, 2H, 2W, c] (dsw_shuffle2_gelu_nhwc and its backward)"""
    return _run(_Shuffle2GeluFn, x)


def conv3x3_nhwc_variant(x, conv, variant="plain"):
    """`conv` (3x3, stride 1, padding 1) applied to the NHWC tensor x [B, H, W, Ci] under bf16 autocast with its
    parameters transformed as `variant` says (see _conv3x3_variant), with or without autograd: without, the kept
    operands of conv3x3_nhwc_infer; with, the transformation is done by differentiable torch ops on the parameters
    and the convolution by the NHWC kernels' autograd function, so the gradients land on `conv`'s own parameters."""
    if not torch.is_grad_enabled():
        return conv3x3_nhwc_infer(x, conv, variant)
    weight, bias = conv.weight, conv.bias
    if variant == "shuffle2":
        Co = weight.shape[0]
        weight = weight.view(Co // 4, 4, *weight.shape[1:]).transpose(0, 1).reshape(Co, *weight.shape[1:])
        bias = None if bias is None else bias.view(Co // 4, 4).t().reshape(Co)
    elif variant == "pad_in4":
        weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, -weight.shape[1] % 4))
    elif variant == "pad_out4":
        extra = -weight.shape[0] % 4
        weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, 0, 0, extra))
        bias = None if bias is None else torch.nn.functional.pad(bias, (0, extra))
    return _Conv3x3Fn.apply(x.contiguous(), weight, bias, torch.bfloat16, None)


# ---------------------------------------------------------------------------------------------
# LayerNorm of msft's ViTBlock (msft_arch.py:152-153, :163-165)
# ---------------------------------------------------------------------------------------------
class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype, pad_out, sink=None, pending=None):
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        M = x2.shape[0]
        ldo = _ceil_pad(C) if pad_out else C
        w32 = weight.detach().float().contiguous()
        b32 = None if bias is None else bias.detach().float().contiguous()
        out = _mark_padded(torch.empty(M, ldo, dtype=out_dtype, device=x.device))
        need_bwd = any(ctx.needs_input_grad[:3])
        stats = torch.empty(2, M, dtype=torch.float32, device=x.device) if need_bwd else None
        call("dsw_layernorm_fwd", ptr(x2), ptr(w32), None if b32 is None else ptr(b32), float(eps), M, C, ptr(out), dtype_code(out_dtype),
             ldo, None if stats is None else ptr(stats[0]), None if stats is None else ptr(stats[1]), stream(x.device))
        if need_bwd:
            ctx.save_for_backward(x2, w32, stats)
        ctx.meta = (x.shape, weight.dtype, None if bias is None else bias.dtype, sink, pending)
        return out[:, :C].view(x.shape) if ldo != C else out.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w32, stats = ctx.saved_tensors
        xshape, wdt, bdt, sink, pending = ctx.meta
        M, C = x2.shape
        dy2 = dy.reshape(-1, C)
        if dy2.dtype not in (torch.float32, torch.bfloat16) or dy2.stride(1) != 1 or (M > 1 and dy2.stride(0) % 4) or \
                dy2.data_ptr() % (4 * dy2.element_size()):
            dy2 = dy2.float().contiguous()
        if not ctx.needs_input_grad[0]:
            sink = pending = None
        # the gradients that reached x around the norm, if they are already there: the residual's dy (a map of x's
        # shape, through the sink) and one that covers x's first columns (pending); the kernel adds them as it writes dx
        if sink is not None:
            sink._same_pass()
        prev = sink.buf if sink is not None else None
        if prev is not None:
            prev = prev.reshape(M, C)
            if prev.dtype != torch.float32 or not prev.is_contiguous():
                prev = prev.float().contiguous()
        part = pending.take() if pending is not None else None
        part2 = None
        if part is not None:
            part2 = _row_view_f32(part)
            if part2 is None or part2.shape[0] != M or part2.shape[1] > C:
                part2 = None
        dx = torch.empty(M, C, dtype=torch.float32, device=x2.device)
        nblk = _lib.load().dsw_rmsnorm_blocks(M)
        partial = torch.empty(nblk, 2 * C, dtype=torch.float32, device=x2.device)
        call("dsw_layernorm_bwd", ctypes.c_void_p(dy2.data_ptr()), dtype_code(dy2.dtype), dy2.stride(0) if M > 1 else C, ptr(x2), ptr(w32),
             ptr(stats[0]), ptr(stats[1]), M, C, ptr(dx), None if prev is None else ptr(prev), None if part2 is None else _dptr(part2),
             0 if part2 is None else (part2.stride(0) if M > 1 else part2.shape[1]), 0 if part2 is None else part2.shape[1], ptr(partial),
             stream(x2.device))
        if part is not None and part2 is None:    # a form the kernel does not read: one pass of torch's
            dx[:, :part.shape[-1]] += part.reshape(M, -1)
        dwb = colsum(partial)
        dw = dwb[:C].to(wdt) if ctx.needs_input_grad[1] else None
        db = dwb[C:].to(bdt) if bdt is not None and ctx.needs_input_grad[2] else None
        return _dx_deliver(sink, dx.view(xshape)), dw, db, None, None, None, None, None


def _row_view_f32(t):
    """t [..., n] (fp32) as a 2-D view [rows, n] with unit column stride, a row stride that is a multiple of 4 and a
    16-byte aligned start - a dense tensor or a slice of the leading columns of one; None when t is not of that form"""
    if t.dtype != torch.float32 or t.dim() < 1 or t.shape[-1] % 4:
        return None
    n = t.shape[-1]
    try:
        t2 = t.view(-1, n)
    except RuntimeError:
        return None
    if t2.stride(1) != 1 or (t2.shape[0] > 1 and t2.stride(0) % 4) or t2.data_ptr() % 16:
        return None
    return t2


def _layer_norm_kernel_ok(x, norm):
    C = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and C % 4 == 0 and _ceil_pad(C) <= 256
            and norm.weight is not None and tuple(norm.normalized_shape) == (C,))


def layer_norm(x, norm, for_projection=False, sink=None, pending=None):
    """`norm(x)` for an nn.LayerNorm over the last dimension of the fp32 stream; for_projection as in rms_norm.
    sink (GradSink, see layer_norm_sink; kernel path only) and pending (PendingGrad holding a gradient for x's leading
    columns): gradients of x that the norm's backward kernel adds as it writes its own."""
    if not _layer_norm_kernel_ok(x, norm):
        assert sink is None, "a gradient sink needs the kernel path (see layer_norm_sink)"
        if pending is not None and torch.is_grad_enabled() and x.requires_grad:
            x = _AddPendingFn.apply(x, pending)
        return norm(x)
    act = autocast_dtype(x.dtype)
    to_act = for_projection and act == torch.bfloat16
    return _run(_LayerNormFn, x, norm.weight, norm.bias, norm.eps, act if to_act else torch.float32, to_act, sink, pending)
