"""DSWT attention modules backed by the HIP kernels.

Same class names, constructor arguments, forward signatures and state_dict keys as
DSwinT_arch.py:95-338 of the reference, so a reference checkpoint loads unchanged and the
reference's ViTBlock / MSFTransformer can hold these modules instead of their own.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import functional as K
from ._lib import QKV_CHUNKED, QKV_INTERLEAVED


class RelativePositionalEncoding(nn.Module):
    """Bias table [(2k-1)^2, heads] + int64 lookup buffer (DSwinT_arch.py:95-125).

    forward() still returns the expanded [1, heads, Lq, Lk] bias for callers that want it; the
    attention kernels read the table itself and never materialise that tensor."""

    def __init__(self, q_size, k_size, num_heads=1, scale=1):
        super().__init__()
        assert q_size <= k_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * k_size - 1) * (2 * k_size - 1), num_heads))
        self.scale = scale
        ii, jj = torch.meshgrid(torch.arange(k_size), torch.arange(k_size), indexing="ij")
        grid = torch.stack([ii, jj])
        lo = (k_size - q_size) // 2
        cq = grid[:, lo:lo + q_size, lo:lo + q_size].reshape(2, -1)
        ck = grid.reshape(2, -1)
        rel = (cq[:, :, None] - ck[:, None, :]).permute(1, 2, 0).contiguous() + (k_size - 1)
        self.register_buffer("relative_position_index", rel[..., 0] * (2 * k_size - 1) + rel[..., 1])
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02, a=-2., b=2.)

    def table(self):
        t = self.relative_position_bias_table
        return t if self.scale == 1 else self.scale * t

    def forward(self):
        bias = self.scale * self.relative_position_bias_table[self.relative_position_index]
        return bias.permute(2, 0, 1).contiguous().unsqueeze(0)


class WindowedAttention(nn.Module):
    """Base class with the reference's reshaping helpers (DSwinT_arch.py:128-172).  The helpers
    are kept for API compatibility; the subclasses' forward never calls them - partition,
    headify, de-headify and reverse are address arithmetic inside the attention kernel."""

    def __init__(self, window_size, dim, num_heads=4):
        super().__init__()
        self.win = window_size
        self.num_heads = num_heads
        assert dim % num_heads == 0, f"dim ({dim}) must be divisible by num_heads ({num_heads})"

    def window_partition(self, x, win, keep_batch=False):
        B, H, W, C = x.shape
        x = x.view(B, H // win, win, W // win, win, C).permute(0, 1, 3, 2, 4, 5).contiguous()
        nw = (H // win) * (W // win)
        return x.view(B, nw, win ** 2, C) if keep_batch else x.view(B * nw, win ** 2, C)

    def reverse_window_partition(self, x, win, H, W, with_batch=False):
        if with_batch:
            x = x.reshape(-1, win ** 2, x.shape[-1])
        N, _, C = x.shape
        x = x.view(N, win, win, C).view(-1, H // win, W // win, win, win, C)
        return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, H, W, C)

    def headify(self, *xs):
        outs = []
        for x in xs:
            N, win, C = x.shape
            outs.append(x.view(N, win, self.num_heads, C // self.num_heads).permute(0, 2, 1, 3).contiguous())
        return outs if len(outs) > 1 else outs[0]

    def deheadify(self, *xs):
        outs = []
        for x in xs:
            N, _, win, _ = x.shape
            outs.append(x.permute(0, 2, 1, 3).contiguous().view(N, win, -1))
        return outs if len(outs) > 1 else outs[0]

    def attention(self, q, k, v, C, attn_drop=0., pe=0):
        """Stock-torch attention on explicit [N,h,L,d] tensors (API compatibility only)."""
        p = ((q @ k.transpose(-2, -1)) * (C // self.num_heads) ** -0.5 + pe).softmax(dim=-1)
        return F.dropout(p, p=attn_drop, training=self.training) @ v


class WindowedMSA(WindowedAttention):
    """DSwinT_arch.py:320-338.  qkv Linear on the un-partitioned [B,H,W,C] tensor (a Linear is
    per token, so it commutes with the partition), one fused attention kernel that cuts the
    windows by address arithmetic (per-head interleaved q|k|v layout, :335), out_proj."""

    def __init__(self, window_size, dim, num_heads=4, attn_drop=0., proj_drop=0.):
        super().__init__(window_size, dim, num_heads)
        self.win = window_size
        self.num_heads = num_heads
        self.attn_drop = attn_drop
        self.qkv = nn.Linear(dim, 3 * dim)
        self.pe = RelativePositionalEncoding(q_size=self.win, k_size=self.win, num_heads=num_heads)
        self.out_proj = nn.Linear(dim, dim)
        # explicit 64-bit seed of the attention dropout of the next training forward (tests / reproducible runs);
        # None = draw it from torch's CPU generator
        self.dropout_seed = None
        # True: when dim is not a multiple of 64 the result is a view of a buffer whose rows are
        # padded to one (what the GEMM library runs fastest on); values and shape are the same, but
        # the tensor is not contiguous.  False: always return a contiguous tensor.
        self.padded_output = True

    def forward(self, x, residual=None, prenorm=None):
        """residual = (stream, r): returns stream + r * attention(x) - the caller's residual (DSwinT_arch.py:355) -
        instead of attention(x), with the residual taken inside out_proj's kernel where that is possible: without
        autograd for any r (ReZero), with autograd for a plain residual (r None or 1.0).  When the stream is x itself
        (y = x + attention(x)), its gradient dy is not added to the projection's input gradient by a pass of its own:
        the kernel that writes that gradient reads it as one more operand (functional.PendingGrad).
        prenorm (inference): the nn.RMSNorm the caller applies in front of the module (DSwinT_arch.py:355); x is then
        the stream itself, and the norm runs inside the q|k|v projection's kernel where that is possible."""
        B, H, W, C = x.shape
        qkv = None
        if prenorm is not None:
            if not (K.FUSE_PROJECTION_GRID_D30 or not (self.win == 16 and C // self.num_heads == 30)) or (self.training and self.attn_drop):
                qkv = K.linear_prenorm(x, prenorm, self.qkv, pad_out=True)
            if qkv is None:
                x = K.rms_norm(x, prenorm, for_projection=True)
        # attention dropout as F.dropout(..., training=self.training) applies it (DSwinT_arch.py:172), inside the kernels
        drop_p = self.attn_drop if self.training else 0.0
        plain = residual is not None and (residual[1] is None or (isinstance(residual[1], (int, float)) and residual[1] == 1))
        pend = K.PendingGrad() if (plain and torch.is_grad_enabled() and residual[0] is x and x.requires_grad
                                   and x.dtype == torch.float32 and x.is_contiguous()) else None
        z = None
        # without autograd: projection and attention in one kernel, q|k|v never written - where that is the faster form
        if qkv is None and drop_p == 0.0 and (K.FUSE_PROJECTION_GRID_D30 or not (self.win == 16 and C // self.num_heads == 30)):
            z = K.projected_window_attention(x, self.qkv, self.pe.table(), self.win, self.num_heads, QKV_INTERLEAVED,
                                             grid=(B, H, W), pad_out=True)
        if z is None:
            if qkv is None:
                qkv = K.linear(x, self.qkv, pad_out=True, grad_extra=pend)
            z = K.window_attention(qkv, self.pe.table(), self.win, self.num_heads, QKV_INTERLEAVED, grid=(B, H, W),
                                   pad_out=True, dropout_p=drop_p, dropout_seed=self.dropout_seed)
        if plain:
            y = K.linear_plus_residual(z, self.out_proj, residual[0], pend) if torch.is_grad_enabled() else None
            if y is None and not torch.is_grad_enabled():
                y = K.linear_residual(z, self.out_proj, residual[0], torch.ones(1, device=x.device))
            if y is not None:
                return y
            # (pend, if any, stays empty: the addition below hands the stream its gradient through autograd)
            return residual[0] + K.linear(z, self.out_proj, pad_out=True)
        if residual is not None:
            y = K.linear_residual(z, self.out_proj, *residual)
            return y if y is not None else K.rezero_residual(residual[0], residual[1], K.linear(z, self.out_proj, pad_out=True))
        return K.linear(z, self.out_proj, pad_out=self.padded_output)


class DeformableSparseWindowAttention(WindowedAttention):
    """DSwinT_arch.py:174-317, NHWC throughout (the reference's NCHW round trip at :259/:317
    is gone).  Kernels: window selection, bilinear window gather, fused attention, gated
    scatter-mean write-back, and NHWC kernels for the sparse discriminator and the depthwise
    out_conv.  Only the two Linears are torch ops."""

    def __init__(self, window_size, dim, num_heads):
        super().__init__(window_size, dim, num_heads)
        self.temperature = 1.0
        self.w = window_size
        self.num_win_selected = 0.25
        self.sparse_discriminator = nn.Sequential(
            nn.Conv2d(dim, 4, 3, 1, 1),
            nn.SiLU(),
            nn.Conv2d(4, 4, self.w, stride=self.w),
            nn.SiLU(),
            nn.Conv2d(4, 4, 3, 1, 1),
        )
        gh, gw = torch.meshgrid(torch.arange(window_size), torch.arange(window_size), indexing="ij")
        self.register_buffer("win_grid", torch.stack([gh, gw], dim=0), persistent=False)
        self.in_proj = nn.Linear(dim, 3 * dim)
        self.pe = RelativePositionalEncoding(q_size=self.w, k_size=self.w, num_heads=num_heads)
        self.out_proj = nn.Linear(dim, dim)
        self.out_conv = nn.Conv2d(dim, dim, 3, 1, 1, groups=dim)
        # explicit Gumbel noise [B, Hw*Ww] for the next training forward (tests / reproducible
        # runs); None = draw it the way F.gumbel_softmax does (DSwinT_arch.py:211)
        self.gumbel_noise = None
        self.last_selection = None
        # True: run the sparse discriminator and out_conv through torch (MIOpen, NCHW copies)
        # instead of the NHWC HIP kernels
        self.use_torch_convs = False
        # True: the gradients of the three readers of x (sparse discriminator, gather, scatter
        # residual) are accumulated in one buffer inside the kernels instead of being added by autograd
        self.fuse_input_grads = True
        # True: the number of windows that passed the gate is never read on the host (the reference reads it
        # twice per call, DSwinT_arch.py:273, :291): all B*k candidates get a window slot, only the passing
        # ones are written back, and in the backward the other slots receive and contribute exactly zero.
        # Same values and gradients; the call does not stall the host, every parameter gets a gradient on
        # every step (zeros when no window passed), last_selection.n is the slot count and .n_dev the pass
        # count on the device.  False restores the host read and processes only the passing windows.
        self.static_inference = True   # without autograd
        self.static_training = True    # with autograd

    # ---- the reference's helper methods (DSwinT_arch.py:201-253), kept for callers and harnesses that use or wrap
    # them.  forward() does not call them: selection, corners and the write-back are the kernels above.  Plain
    # torch ops on whatever device the arguments live on; same arguments, same return values.
    def candidate_indexes(self, x, scores):
        """top-k windows of scores [B,1,Hw,Ww] by (Gumbel-)softmax probability -> flat (b, h, w) indices [B*k]
        and the straight-through value (1 - p).detach() + p (DSwinT_arch.py:201-225)"""
        B, _, Hw, Ww = scores.shape
        k = int((Hw * Ww) * self.num_win_selected)
        flat = scores.reshape(B, -1)
        if self.training:
            noise = self.gumbel_noise.to(flat.device) if self.gumbel_noise is not None else self._noise(flat)
            probs = ((flat + noise) / self.temperature).softmax(dim=-1)
        else:
            probs = (flat / self.temperature).softmax(dim=-1)
        top_p, top_i = probs.topk(k, dim=1)
        b_idx = torch.arange(B, device=x.device).repeat_interleave(k)
        return b_idx, (top_i // Ww).flatten(), (top_i % Ww).flatten(), ((1.0 - top_p).detach() + top_p).flatten()

    def cords(self, b_idx, h_idx, w_idx, offsets, H, W):
        """window corners after the learned shift, clamped into the map (DSwinT_arch.py:227-237)"""
        shift_h = (offsets[b_idx, 0, h_idx, w_idx] * self.w - self.w / 2.).long()
        shift_w = (offsets[b_idx, 1, h_idx, w_idx] * self.w - self.w / 2.).long()
        h1 = (h_idx * self.w + shift_h).clamp(0, H - self.w)
        w1 = (w_idx * self.w + shift_w).clamp(0, W - self.w)
        return h1, w1, h1 + self.w, w1 + self.w

    def integrate(self, x, delta, b_idx, h1, w1):
        """mean of the window deltas [n,C,w,w] at the pixels they cover, zero elsewhere, as a [B,C,H,W] map
        (DSwinT_arch.py:239-253)"""
        B, C, H, W = x.shape
        gi, gj = self.win_grid.to(h1.device)
        pix = ((h1.view(-1, 1, 1) + gi) * W + (w1.view(-1, 1, 1) + gj)).flatten(1)                 # [n, w*w]
        plane = (b_idx.view(-1, 1) * C + torch.arange(C, device=h1.device).view(1, C)) * (H * W)    # [n, C]
        index = (plane.unsqueeze(2) + pix.unsqueeze(1)).reshape(-1)
        out = torch.zeros(B * C * H * W, dtype=delta.dtype, device=delta.device)
        out.scatter_reduce_(0, index, delta.reshape(-1), reduce="mean", include_self=False)
        return out.view(B, C, H, W)

    def _noise(self, scores_flat):
        if not self.training:
            return None
        if self.gumbel_noise is not None:
            return self.gumbel_noise.to(scores_flat.device)
        return -torch.empty_like(scores_flat, memory_format=torch.legacy_contiguous_format).exponential_().log()

    def forward(self, x, residual=None):
        """residual = (stream, r): returns stream + r * attention(x) - the caller's residual (DSwinT_arch.py:356) -
        taken inside the closing convolution's kernel where that is the last step: without autograd for any r (ReZero),
        with autograd for a plain residual (r None or 1.0; when the stream is x itself its gradient joins the
        module's input gradient inside the scatter's backward kernel, functional.PendingGrad)."""
        if residual is None:
            return self._forward(x, None)
        plain = residual[1] is None or (isinstance(residual[1], (int, float)) and residual[1] == 1)
        if plain and not torch.is_grad_enabled():
            residual = (residual[0], torch.ones(1, device=x.device))
        y = self._forward(x, residual)
        if isinstance(y, tuple):     # (result, True): the residual is in already
            return y[0]
        if plain:
            return residual[0] + y
        return K.rezero_residual(residual[0], residual[1], y)

    def _forward(self, x, residual):
        B, H, W, C = x.shape
        w = self.w
        x_in = x
        x = K.supported_input(x).contiguous()
        if x is not x_in:
            x_in = None      # the kernels read a copy: the caller's tensor is not the one the sink collects for
        act = K.autocast_dtype(x.dtype)
        # the three readers of x meet in one gradient buffer (only when all three are certain to run
        # in the backward pass: the early returns below leave the sparse discriminator alone)
        Hw0, Ww0 = H // w, W // w
        fuse = (self.fuse_input_grads and not self.use_torch_convs and x.requires_grad and
                int(Hw0 * Ww0 * self.num_win_selected) > 0)
        sink = K.GradSink(3) if fuse else None
        if self.use_torch_convs:
            sd = self.sparse_discriminator(x.permute(0, 3, 1, 2).contiguous()).contiguous()
        else:
            sd = K.sparse_discriminator_nhwc(x, (self.sparse_discriminator[0], self.sparse_discriminator[2],
                                                 self.sparse_discriminator[4]), w, out_dtype=act, sink=sink)
        Hw, Ww = sd.shape[2], sd.shape[3]
        k = int((Hw * Ww) * self.num_win_selected)
        self.last_selection = None
        if k == 0:  # DSwinT_arch.py:273-274
            return x
        noise = self._noise(sd[:, 0].reshape(B, -1))
        static = self.static_training if torch.is_grad_enabled() else self.static_inference
        if static and not self.use_torch_convs:
            sel = K.select_windows_static(sd, noise, self.temperature, k, w, H, W)
            self.last_selection = sel.detached()
            before = K.gather_windows(x, sel, w, out_dtype=act, sink=sink, pad_out=True)
            # the slots behind the windows that passed are switched off inside the kernels (device-side count):
            # no arithmetic for them in the projections and the attention, forward or backward
            live = (sel.n_dev, w * w)
            z = K.projected_window_attention(before, self.in_proj, self.pe.table(), w, self.num_heads, QKV_CHUNKED, pad_out=True,
                                             n_live=sel.n_dev)
            if z is None:
                qkv = K.linear(before, self.in_proj, pad_out=True, live=live)
                z = K.window_attention(qkv, self.pe.table(), w, self.num_heads, QKV_CHUNKED, pad_out=True, n_live=sel.n_dev)
            after = K.linear(z, self.out_proj, pad_out=True, live=live)
            # no window passed (DSwinT_arch.py:291-292: return x): the scatter leaves its map unwritten and the last
            # kernel copies x instead of convolving
            y = K.scatter_mean_residual(x, after, sel, w, sink=sink, skip_if_dead=True)
            if residual is not None and torch.is_grad_enabled():
                if residual[1] is None or isinstance(residual[1], (int, float)):   # a plain residual (forward() checked: 1)
                    # the stream's gradient meets the three readers of x in the sink when the stream is x
                    pend = sink.pending if (sink is not None and residual[0] is x_in) else None
                    out = K.dwconv3x3_plus_residual_nhwc(y, self.out_conv.weight, self.out_conv.bias, residual[0], pend,
                                                         identity_if_zero=sel.n_dev, identity_src=x)
                    if out is not None:
                        return out, True
            elif residual is not None:
                out = K.dwconv3x3_residual_nhwc(y, self.out_conv.weight, self.out_conv.bias, residual[0], residual[1],
                                                identity_if_zero=sel.n_dev, identity_src=x)
                if out is not None:
                    return out, True
            return K.dwconv3x3_nhwc(y, self.out_conv.weight, self.out_conv.bias, out_dtype=act, identity_if_zero=sel.n_dev, identity_src=x)
        sel = K.select_windows(sd, None, noise, self.temperature, k, w, H, W)  # sigmoids taken in the kernels
        self.last_selection = sel.detached()
        if sel.n == 0:  # DSwinT_arch.py:291-292
            if sink is not None:
                sink.expected = 1  # only the sparse discriminator saw x
            return x
        # rows of the window matrices stay padded to multiples of 64 elements between the kernels and
        # the projections (views of the first C / 3C columns)
        before = K.gather_windows(x, sel, w, out_dtype=act, sink=sink, pad_out=True)   # [n, L, C]
        z = K.projected_window_attention(before, self.in_proj, self.pe.table(), w, self.num_heads, QKV_CHUNKED, pad_out=True)
        if z is None:
            qkv = K.linear(before, self.in_proj, pad_out=True)
            z = K.window_attention(qkv, self.pe.table(), w, self.num_heads, QKV_CHUNKED, pad_out=True)
        after = K.linear(z, self.out_proj, pad_out=True)
        y = K.scatter_mean_residual(x, after, sel, w, sink=sink)          # [B,H,W,C]
        if self.use_torch_convs:
            return self.out_conv(y.permute(0, 3, 1, 2).contiguous()).permute(0, 2, 3, 1).contiguous()
        return K.dwconv3x3_nhwc(y, self.out_conv.weight, self.out_conv.bias, out_dtype=act)
