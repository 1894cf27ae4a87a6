"""MSFT attention and multi-scale modules backed by the HIP kernels.

Same class names, constructor arguments, forward signatures and state_dict keys as
msft_arch.py:37-143 of the reference.
"""
import torch
from torch import nn

from . import functional as K
from ._lib import QKV_CHUNKED


def _ln_out_dtype(x):
    # LayerNorm is on CUDA autocast's fp32 list: under autocast its output is fp32
    return torch.float32 if torch.is_autocast_enabled("cuda") else x.dtype


class PatchMerging(nn.Module):
    """msft_arch.py:37-56: PixelUnshuffle -> NHWC -> LayerNorm -> Linear(no bias) -> NCHW.
    The unshuffle, the layout copy and the LayerNorm are one kernel; the copy back is another."""

    def __init__(self, dim, scale):
        super().__init__()
        self.pixel_unshuffle = nn.PixelUnshuffle(scale)
        self.scale = scale
        self.norm = nn.LayerNorm(dim)
        self.linear = nn.Linear(dim, dim, bias=False)

    def forward(self, x):
        x = K.supported_input(x)
        t = K.unshuffle_nhwc_layernorm(x, self.norm.weight, self.norm.bias, self.scale, self.norm.eps,
                                       out_dtype=_ln_out_dtype(x))
        return K.shuffle_nchw(K.linear(t, self.linear) if t.is_cuda else self.linear(t), 1)

    def forward_nhwc(self, z):
        """The same function between channels-last tensors: z [B, H, W, c] -> tokens [B, (H/s)*(W/s), c*s*s]
        (channel order of PixelUnshuffle: c*s*s + dy*s + dx).  No NCHW tensor on the way."""
        B, H, W, c = z.shape
        s = self.scale
        u = torch.empty(B, (H // s) * (W // s), c * s * s, dtype=torch.float32, device=z.device)
        u.view(B, H // s, W // s, c, s, s).copy_(z.view(B, H // s, s, W // s, s, c).permute(0, 1, 3, 5, 2, 4))
        return K.linear(K.layer_norm(u, self.norm, for_projection=True), self.linear)


class PatchUnMerging(nn.Module):
    """msft_arch.py:59-78: NHWC -> LayerNorm -> Linear(no bias) -> NCHW -> PixelShuffle."""

    def __init__(self, dim, scale):
        super().__init__()
        self.pixel_shuffle = nn.PixelShuffle(scale)
        self.scale = scale
        self.norm = nn.LayerNorm(dim)
        self.linear = nn.Linear(dim, dim, bias=False)

    def forward(self, x):
        x = K.supported_input(x)
        t = K.unshuffle_nhwc_layernorm(x, self.norm.weight, self.norm.bias, 1, self.norm.eps,
                                       out_dtype=_ln_out_dtype(x))
        return K.shuffle_nchw(K.linear(t, self.linear) if t.is_cuda else self.linear(t), self.scale)

    def forward_tokens(self, t, h, w):
        """The same function from the fp32 tokens t [B, h*w, dim] to the channels-last map [B, h*s, w*s, dim/s/s]."""
        B, _, dim = t.shape
        s = self.scale
        c = dim // (s * s)
        y = K.linear(K.layer_norm(t, self.norm, for_projection=True), self.linear)
        return y.view(B, h, w, c, s, s).permute(0, 1, 4, 2, 5, 3).reshape(B, h * s, w * s, c)


class WindowedMSA(nn.Module):
    """msft_arch.py:80-143.  qkv Linear, then ONE kernel for window partition + head split +
    SDPA (scale d^-0.5, no bias) + LePE (depthwise 3x3 of V inside each window) + reverse
    partition, then out_proj.  The reference's eight layout copies are gone."""

    def __init__(self, window_size, dim, num_heads=4, attn_drop=0.1, proj_drop=0.):
        super().__init__()
        self.window_size = window_size
        self.num_heads = num_heads
        assert dim % num_heads == 0, f"dim ({dim}) must be divisible by num_heads ({num_heads})"
        self.qkv = nn.Linear(dim, 3 * dim)
        self.pe_encoder = nn.Conv2d(dim, dim, 3, 1, 1, groups=dim)
        self.out_proj = nn.Linear(dim, dim)
        self.attn_drop = attn_drop
        self.proj_drop = nn.Dropout(proj_drop)
        # explicit 64-bit seed of the attention dropout of the next forward (tests / reproducible runs);
        # None = draw it from torch's CPU generator
        self.dropout_seed = None

    def forward(self, x, H, W, residual=None):
        """residual = (stream, dropout module or None, GradSink / PendingGrad or None): returns
        stream + dropout(attention(x)) - the residual of the caller's block (msft_arch.py:163-164) - taken inside
        out_proj's kernel where that is possible."""
        B, N, C = x.shape
        assert N == H * W
        qkv = K.linear(x, self.qkv)
        # the reference hands attn_drop to SDPA whatever the mode (msft_arch.py:136): dropout in eval() too
        z = K.window_attention(qkv, None, self.window_size, self.num_heads, QKV_CHUNKED, grid=(B, H, W),
                               lepe_w=self.pe_encoder.weight, lepe_b=self.pe_encoder.bias,
                               dropout_p=self.attn_drop, dropout_seed=self.dropout_seed, pad_out=residual is not None)
        if residual is None:
            return self.proj_drop(K.linear(z, self.out_proj))
        stream, drop, sink = residual
        y = None
        if K.FUSE_RESIDUAL_DROPOUT and torch.is_grad_enabled() and (not self.proj_drop.training or self.proj_drop.p == 0):
            y = K.linear_plus_residual(z, self.out_proj, stream, sink, dropout=drop)
        if y is None:
            one = torch.ones(1, device=x.device)
            y = K.rezero_residual(stream, one, self.proj_drop(K.linear(z, self.out_proj)),
                                  sink=sink if isinstance(sink, K.GradSink) else None, dropout=drop)
        return y


class MLP(nn.Module):
    """msft_arch.py:19-33: Linear - GELU - Linear - Dropout; the Linears run in the projection kernels."""

    def __init__(self, in_features, hidden_features=None, out_features=None, ffn_scale=4, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or int(in_features * ffn_scale)
        self.mlp = nn.Sequential(nn.Linear(in_features, hidden_features), nn.GELU(),
                                 nn.Linear(hidden_features, out_features), nn.Dropout(drop))

    def forward(self, x, residual=None):
        """residual = (stream, dropout module or None, GradSink / PendingGrad or None): returns stream + dropout(mlp(x)),
        the residual of the caller's block (msft_arch.py:165), taken inside the second projection's kernel where that is
        possible."""
        if not x.is_cuda:
            return self.mlp(x)
        # the two Linears through the projection kernels (they read the LayerNorm's fp32 output where it lies), the
        # activation in their epilogues under bf16 autocast
        inner_drop = self.mlp[3].training and self.mlp[3].p > 0
        if residual is not None and K.FUSE_RESIDUAL_DROPOUT and torch.is_grad_enabled() and not inner_drop:
            y = K.mlp_gelu(x, self.mlp[0], self.mlp[2], residual=residual)
            if y is not None:
                return y
        y = K.mlp_gelu(x, self.mlp[0], self.mlp[2])
        if y is None:
            y = K.linear(K.gelu(K.linear(x, self.mlp[0], pad_out=True), pad_out=True), self.mlp[2])
        y = self.mlp[3](y)
        if residual is None:
            return y
        stream, drop, sink = residual
        return K.rezero_residual(stream, torch.ones(1, device=x.device), y, sink=sink if isinstance(sink, K.GradSink) else None, dropout=drop)


class ViTBlock(nn.Module):
    """msft_arch.py:146-167, the caller of WindowedMSA: NCHW in and out.  The NCHW->token copy
    is fused with ln1 and the token->NCHW copy is one kernel."""

    def __init__(self, window_size, dim, n_heads=4, ffn_scale=2, drop_path=0.0):
        super().__init__()
        self.msa = WindowedMSA(dim=dim, window_size=window_size, num_heads=n_heads)   # attn_drop 0.1, msft_arch.py:149
        self.mlp = MLP(dim, ffn_scale=ffn_scale)
        self.ln1 = nn.LayerNorm(dim)
        self.ln2 = nn.LayerNorm(dim)
        self.drop_path = nn.Dropout(drop_path)

    def forward(self, x):
        B, C, H, W = x.shape
        t = K.unshuffle_nhwc(K.supported_input(x), 1).view(B, H * W, C)
        # both LayerNorms feed projections: under bf16 autocast they write bf16 padded rows directly
        n1 = K.layer_norm(t, self.ln1, for_projection=True)
        t = self.drop_path(self.msa(n1, H, W)) + t
        t = self.drop_path(self.mlp(K.layer_norm(t, self.ln2, for_projection=True))) + t
        return K.shuffle_nchw(t.view(B, H, W, C), 1)

    def forward_tokens(self, t, H, W, grad_extra=None):
        """The block on the fp32 token stream t [B, H*W, C] (what forward computes between its two layout copies).
        The residual additions and the block's drop_path are one kernel each (the keep decisions of the dropout are
        a seeded hash of the position: no mask tensor); under bf16 autocast the stream stays fp32, as the LayerNorm
        outputs of the reference do.  With autograd, the two gradients of each residual's input (the residual's dy
        and the norm's) meet inside the norm's backward kernel, and grad_extra (a PendingGrad: a gradient for t's
        leading columns that is there when the backward pass reaches the first norm) is added by that kernel too."""
        one = self._one(t.device)
        sink = K.layer_norm_sink(t, self.ln1)
        n1 = K.layer_norm(t, self.ln1, for_projection=True, sink=sink, pending=grad_extra)
        if not torch.is_grad_enabled():
            t = K.rezero_residual(t, one, self.msa(n1, H, W), sink=sink, dropout=self.drop_path)
            return K.rezero_residual(t, one, self.mlp(K.layer_norm(t, self.ln2, for_projection=True)), dropout=self.drop_path)
        # training: the residual and its dropout inside the branch's closing projection (msft.WindowedMSA / MLP)
        t = self.msa(n1, H, W, residual=(t, self.drop_path, sink))
        sink = K.layer_norm_sink(t, self.ln2)
        return self.mlp(K.layer_norm(t, self.ln2, for_projection=True, sink=sink), residual=(t, self.drop_path, sink))

    def _one(self, device):
        o = getattr(self, "_one_buf", None)
        if o is None or o.device != device:
            o = self._one_buf = torch.ones(1, device=device)
        return o
