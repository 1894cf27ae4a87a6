"""CPU restatement of the MSFT hot path (reference: msft_arch.py:37-167).

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.  Plain torch ops in
the reference's op order; parameters travel as a dict with the reference's
``state_dict`` key names.  Pinned by ``tests/golden/msft_*.npz``, which were
produced by importing and running ``msft_arch.py`` itself.
"""
import torch
import torch.nn.functional as F


def window_partition(x, H, W, win, num_heads):
    """[B,N,C] -> [B*nW, h, win^2, C/h]  (msft_arch.py:94-102)."""
    B, N, C = x.shape
    t = x.view(B, H // win, win, W // win, win, C).permute(0, 1, 3, 2, 4, 5).contiguous()
    t = t.view(-1, win * win, num_heads, C // num_heads)
    return t.permute(0, 2, 1, 3).contiguous()


def reverse_window_partition(x, H, W, win):
    """[B*nW, h, win^2, d] -> [B, H*W, h*d]  (msft_arch.py:104-111)."""
    C = x.shape[1] * x.shape[3]
    t = x.permute(0, 2, 1, 3).contiguous().view(-1, win, win, C)
    t = t.view(-1, H // win, W // win, win, win, C).permute(0, 1, 3, 2, 4, 5).contiguous()
    return t.view(-1, H * W, C)


def locally_enhanced_pe(v, H, W, win, num_heads, conv_w, conv_b):
    """Depthwise 3x3 conv of V inside each window, zero padded at the window
    edge; returns V and the LePE term in head layout (msft_arch.py:113-125)."""
    B, N, C = v.shape
    t = v.transpose(-2, -1).contiguous().view(B, C, H // win, win, W // win, win)
    t = t.permute(0, 2, 4, 1, 3, 5).contiguous().view(-1, C, win, win)
    lepe = F.conv2d(t, conv_w, conv_b, 1, 1, groups=C)
    d = C // num_heads
    lepe = lepe.reshape(-1, num_heads, d, win * win).permute(0, 1, 3, 2).contiguous()
    vh = t.reshape(-1, num_heads, d, win * win).permute(0, 1, 3, 2).contiguous()
    return vh, lepe


def _sdpa(q, k, v, attn_drop, keep):
    """F.scaled_dot_product_attention(q, k, v, dropout_p) (msft_arch.py:136); with ``keep`` [N,h,L,L] bool the
    dropout draw is explicit (probabilities times keep / (1 - p), SDPA's own rule)."""
    if keep is None:
        return F.scaled_dot_product_attention(q, k, v, dropout_p=attn_drop)
    probs = ((q @ k.transpose(-2, -1)) * q.shape[-1] ** -0.5).softmax(dim=-1)
    return (probs * keep.to(probs.dtype) / (1.0 - attn_drop)) @ v


def wmsa_forward(P, x, H, W, win, num_heads, attn_drop=0.0, keep=None):
    """msft WindowedMSA.forward (msft_arch.py:127-143): qkv Linear, q|k|v
    thirds, SDPA (scale d^-0.5, no bias, no mask) + LePE, out_proj."""
    qkv = F.linear(x, P["qkv.weight"], P["qkv.bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    q = window_partition(q, H, W, win, num_heads)
    k = window_partition(k, H, W, win, num_heads)
    v, lepe = locally_enhanced_pe(v, H, W, win, num_heads, P["pe_encoder.weight"], P["pe_encoder.bias"])
    z = _sdpa(q, k, v, attn_drop, keep) + lepe
    z = reverse_window_partition(z, H, W, win)
    return F.linear(z, P["out_proj.weight"], P["out_proj.bias"])


def window_attention_lepe_core(qkv, H, W, win, num_heads, conv_w, conv_b):
    """The part of wmsa_forward between the two Linears: [B,N,3C] -> [B,N,C]."""
    q, k, v = qkv.chunk(3, dim=-1)
    q = window_partition(q, H, W, win, num_heads)
    k = window_partition(k, H, W, win, num_heads)
    v, lepe = locally_enhanced_pe(v, H, W, win, num_heads, conv_w, conv_b)
    z = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0) + lepe
    return reverse_window_partition(z, H, W, win)


def patch_merging(P, x, scale):
    """PixelUnshuffle -> NHWC -> LayerNorm -> Linear(no bias) -> NCHW
    (msft_arch.py:37-56).  x [B,c,H,W] with c*scale^2 == dim."""
    t = F.pixel_unshuffle(x, scale).permute(0, 2, 3, 1).contiguous()
    t = F.layer_norm(t, (t.shape[-1],), P["norm.weight"], P["norm.bias"])
    t = F.linear(t, P["linear.weight"])
    return t.permute(0, 3, 1, 2).contiguous()


def patch_unmerging(P, x, scale):
    """NHWC -> LayerNorm -> Linear(no bias) -> NCHW -> PixelShuffle
    (msft_arch.py:59-78)."""
    t = x.permute(0, 2, 3, 1).contiguous()
    t = F.layer_norm(t, (t.shape[-1],), P["norm.weight"], P["norm.bias"])
    t = F.linear(t, P["linear.weight"])
    return F.pixel_shuffle(t.permute(0, 3, 1, 2).contiguous(), scale)


def unshuffle_nhwc(x, scale):
    """PixelUnshuffle followed by the NCHW->NHWC copy (msft_arch.py:45-48)."""
    return F.pixel_unshuffle(x, scale).permute(0, 2, 3, 1).contiguous()


def shuffle_nchw(t, scale):
    """NHWC->NCHW copy followed by PixelShuffle (msft_arch.py:74-76)."""
    return F.pixel_shuffle(t.permute(0, 3, 1, 2).contiguous(), scale)


def vit_block(P, x, win, num_heads, attn_drop=0.0, keep=None):
    """msft ViTBlock.forward, residual dropout off (msft_arch.py:156-167): NCHW in/out,
    LayerNorm + WindowedMSA residual, LayerNorm + GELU MLP residual.  attn_drop / keep: the
    attention's own dropout (msft_arch.py:136, :149) with an explicit draw, as in wmsa_forward."""
    B, C, H, W = x.shape
    t = x.view(B, C, H * W).permute(0, 2, 1).contiguous()
    n1 = F.layer_norm(t, (C,), P["ln1.weight"], P["ln1.bias"])
    msa = {k[len("msa."):]: v for k, v in P.items() if k.startswith("msa.")}
    t = wmsa_forward(msa, n1, H, W, win, num_heads, attn_drop=attn_drop, keep=keep) + t
    n2 = F.layer_norm(t, (C,), P["ln2.weight"], P["ln2.bias"])
    m = F.linear(F.gelu(F.linear(n2, P["mlp.mlp.0.weight"], P["mlp.mlp.0.bias"])),
                 P["mlp.mlp.2.weight"], P["mlp.mlp.2.bias"])
    t = m + t
    return t.permute(0, 2, 1).contiguous().view(B, C, H, W)
