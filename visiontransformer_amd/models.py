"""The callers of the hot path: the DSWT and MSFT super-resolution networks.

Stock PyTorch layers around the kernel-backed attention modules, with the reference's module
names so that its checkpoints load (`state_dict` keys identical):
DSWT: DSwinT_arch.py:36-69, :340-505.   MSFT: msft_arch.py:169-210, :257-273, :318-435.
Nothing here is a kernel; it exists so that whole-model throughput (x4 SR megapixels/s,
training patches/s) can be measured and the drop-in can be checked end to end.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import dswin, msft
from . import functional as K


# ------------------------------------------------------------------------------------------------
# DSWT
# ------------------------------------------------------------------------------------------------
class LastChannelConv2D(nn.Module):
    """3x3 conv on an NHWC tensor (DSwinT_arch.py:36-44)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, groups=1):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, groups=groups)

    def forward(self, x, residual=None):
        """residual: a tensor to add to the result (the skip connection of the block this convolution closes);
        without autograd the addition happens in the convolution's epilogue"""
        if residual is not None:
            y = K.conv3x3_nhwc_residual(x, self.conv, residual)
            return y if y is not None else self.forward(x) + residual
        y = K.conv3x3_nhwc(x, self.conv)  # bf16 autocast: NHWC kernels (fwd and bwd), no layout copies
        if y is not None:
            return y
        return self.conv(x.permute(0, 3, 1, 2).contiguous()).permute(0, 2, 3, 1).contiguous()


class _Stacked:
    """weight / bias pair with the attribute names of an nn.Linear"""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias


class GatedMLP(nn.Module):
    """SiLU-gated MLP (DSwinT_arch.py:47-69); registered under the reference's name `MLP`."""

    def __init__(self, in_features, hidden_features=None, out_features=None, ffn_scale=4, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or int(in_features * ffn_scale)
        self.fc1_gate = nn.Linear(in_features, hidden_features)
        self.fc1_content = nn.Linear(in_features, hidden_features)
        self.act = nn.SiLU()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def _stacked(self):
        """rows of fc1_gate and fc1_content stacked; kept between calls when autograd is off and the
        parameters have not been modified"""
        ps = (self.fc1_gate.weight, self.fc1_content.weight, self.fc1_gate.bias, self.fc1_content.bias)
        if not K._may_keep_operands():   # autograd on, or stream capture (see functional._may_keep_operands)
            return _Stacked(torch.cat(ps[:2]), torch.cat(ps[2:]))
        key = tuple((p._version, p.data_ptr()) for p in ps)
        kept = getattr(self, "_kept_fc1", None)
        if kept is None or kept[0] != key:
            kept = (key, _Stacked(torch.cat(ps[:2]).detach(), torch.cat(ps[2:]).detach()))
            object.__setattr__(self, "_kept_fc1", kept)
        return kept[1]

    def _interleaved(self):
        """gate and content rows of fc1 in the layout of the fused projection + gated-SiLU kernel; kept between
        calls like _stacked()"""
        ps = (self.fc1_gate.weight, self.fc1_content.weight, self.fc1_gate.bias, self.fc1_content.bias)
        if not K._may_keep_operands():
            return K.interleave_gate_content(*ps)
        key = tuple((p._version, p.data_ptr()) for p in ps)
        kept = getattr(self, "_kept_fc1_inter", None)
        if kept is None or kept[0] != key:
            kept = (key, K.interleave_gate_content(*ps))
            object.__setattr__(self, "_kept_fc1_inter", kept)
        return kept[1]

    def _fused_operands(self, prenorm):
        """the operand images of the one-kernel MLP (functional.mlp_gated_operands), kept between calls like _stacked()"""
        ps = (self.fc1_gate.weight, self.fc1_content.weight, self.fc1_gate.bias, self.fc1_content.bias, self.fc2.weight, self.fc2.bias,
              prenorm.weight)
        if not K._may_keep_operands():
            return K.mlp_gated_operands(*ps)
        key = tuple(None if p is None else (p._version, p.data_ptr()) for p in ps)
        kept = getattr(self, "_kept_fused", None)
        if kept is None or kept[0] != key:
            kept = (key, K.mlp_gated_operands(*ps))
            object.__setattr__(self, "_kept_fused", kept)
        return kept[1]

    def forward(self, x, residual=None, prenorm=None):
        """residual = (stream, r), inference only: returns stream + r * mlp(x), the residual taken inside fc2's
        kernel (the caller's ReZero residual, DSwinT_arch.py:357).  prenorm (with residual): the nn.RMSNorm the caller
        applies in front of the module; x is then the stream itself and the norm runs inside fc1's kernel where that
        is possible."""
        if residual is not None:
            h = None
            Hd = self.fc1_gate.out_features
            if (prenorm is not None and residual[0] is x and residual[1] is not None and torch.is_tensor(residual[1]) and residual[1].numel() == 1
                    and isinstance(self.act, nn.SiLU) and K.mlp_gated_ok(x, prenorm, Hd, self.fc2.out_features)):
                # norm, both projections, the gate and the residual in one kernel: the hidden activations stay in registers
                return K.mlp_gated_residual(x, prenorm, self._fused_operands(prenorm), residual[1])
            fused_ok = (K.FUSE_GATED_SILU and (self.fc1_gate.bias is None) == (self.fc1_content.bias is None)
                        and (Hd + 31) // 32 * 32 == (Hd + 63) // 64 * 64 and self.fc1_gate.in_features <= 192)
            if prenorm is not None:
                if fused_ok:
                    h = K.linear_gated_silu(x, *self._interleaved(), self.fc1_gate.out_features, prenorm=prenorm)
                if h is None:
                    x = K.rms_norm(x, prenorm, for_projection=True)
            if h is None and fused_ok:
                h = K.linear_gated_silu(x, *self._interleaved(), self.fc1_gate.out_features)   # fc1 and the gate in one kernel
            if h is None:
                h = K.gated_silu(K.linear(x, self._stacked(), pad_out=True), pad_out=True)
            y = K.linear_residual(h, self.fc2, *residual)
            return y if y is not None else K.rezero_residual(residual[0], residual[1], K.linear(h, self.fc2))
        if not x.is_cuda:
            return self.drop(self.fc2(self.act(self.fc1_gate(x)) * self.fc1_content(x)))
        # gate and content share their input: one projection onto both (the rows of the two weights
        # stacked), read from wherever x lies, then the gated product and fc2 through the same kernels
        H = self.fc1_gate.out_features
        both = self._stacked()
        h = K.linear(x, both, pad_out=True)
        if isinstance(self.act, nn.SiLU):
            return self.drop(K.linear(K.gated_silu(h, pad_out=True), self.fc2))
        return self.drop(K.linear(self.act(h[..., :H]) * h[..., H:], self.fc2))


class DSWTViTBlock(nn.Module):
    """DSwinT_arch.py:340-358: RMSNorm + ReZero residuals around WindowedMSA, DS-MSA and the MLP."""

    def __init__(self, window_size, dim, n_heads=4, ffn_scale=2, drop_path=0.0):
        super().__init__()
        self.wmsa = dswin.WindowedMSA(dim=dim, window_size=window_size, num_heads=n_heads)
        self.sdwa = dswin.DeformableSparseWindowAttention(window_size, dim, num_heads=n_heads)
        self.mlp1 = GatedMLP(dim, ffn_scale=ffn_scale)
        self.ln1 = nn.RMSNorm(dim)
        self.ln2 = nn.RMSNorm(dim)
        self.ln3 = nn.RMSNorm(dim)
        self.rezero1 = nn.Parameter(torch.zeros(1))
        self.rezero2 = nn.Parameter(torch.zeros(1))
        self.rezero3 = nn.Parameter(torch.zeros(1))
        self.drop_path = nn.Dropout(drop_path)

    def forward(self, x):
        # x + rezero * branch(norm(x)): one pass over the fp32 stream for the residual, the branch read where
        # it lies, and the two gradients of x meeting in one buffer (K.prenorm_sink)
        fuse = (not torch.is_grad_enabled() and not self.training and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
                and K.autocast_dtype(None) == torch.bfloat16
                and isinstance(self.mlp1.act, nn.SiLU) and self.mlp1.drop.p == 0)
        for norm, rz, branch, proj in ((self.ln1, self.rezero1, self.wmsa, True), (self.ln3, self.rezero3, self.sdwa, False),
                                       (self.ln2, self.rezero2, self.mlp1, True)):
            if fuse:
                # inference: the branch's closing step (projection; DS-MSA's depthwise convolution) adds the residual
                # itself (no bf16 round trip of the branch, no residual pass)
                # (DS-MSA's input in bf16: its convolutions and its gather round it to bf16 as their first step, and
                # the map it writes back into is rounded to bf16 on its way into out_conv; half the bytes on every pass)
                if proj:    # the norm inside the branch's first projection where its kernel takes it
                    x = branch(x, residual=(x, rz), prenorm=norm)
                else:
                    x = branch(K.rms_norm(x, norm, dense_act=DSMSA_BF16_INPUT), residual=(x, rz))
                continue
            sink = K.prenorm_sink(x, norm, rz)
            b = branch(K.rms_norm(x, norm, for_projection=proj, sink=sink))
            if sink is not None and not (b.shape == x.shape and b.dtype in (torch.float32, torch.bfloat16)):
                raise RuntimeError("branch output of an unexpected form")
            x = K.rezero_residual(x, rz, b, sink=sink, dropout=self.drop_path)   # drop_path applied inside
        return x


class DSWTResidualBlock(nn.Module):
    """DSwinT_arch.py:361-374."""

    def __init__(self, n_sub_blocks, levels, window_size, dim, level_dim, n_heads, n_heads_fuse, ffn_scale=2,
                 drop_path=0.):
        super().__init__()
        self.layers = nn.Sequential(*[DSWTViTBlock(window_size=window_size, dim=dim, n_heads=n_heads,
                                                   ffn_scale=ffn_scale, drop_path=drop_path)
                                      for _ in range(n_sub_blocks)])
        self.out_layer = LastChannelConv2D(dim, dim, 3, 1, 1)

    def forward(self, x):
        return self.out_layer(self.layers(x), residual=x)


def _reconstruction_tail(dim, feature_dim, scale_factor, last_div):
    layers = [nn.Conv2d(dim, feature_dim, 3, 1, 1), nn.GELU(), nn.Conv2d(feature_dim, feature_dim, 3, 1, 1)]
    for _ in range(int(np.log2(scale_factor))):
        layers += [nn.Conv2d(feature_dim, feature_dim * 4, 3, 1, 1), nn.PixelShuffle(2), nn.GELU()]
    layers += [nn.Conv2d(feature_dim, feature_dim // last_div, 3, 1, 1), nn.GELU(),
               nn.Conv2d(feature_dim // last_div, 3, 3, 1, 1)]
    return nn.Sequential(*layers)


def _tail_plan(seq, x):
    """How the reconstruction tail runs on an NHWC tensor: a list of steps, or None when some
    module or size is not one the NHWC kernels take."""
    mods, steps, i = list(seq), [], 0
    B, H, W, C = x.shape
    dtype = x.dtype
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            up = (i + 2 < len(mods) and isinstance(mods[i + 1], nn.PixelShuffle) and mods[i + 1].upscale_factor == 2
                  and isinstance(mods[i + 2], nn.GELU))
            variant = "shuffle2" if up else ("pad_out4" if m.out_channels % 4 else "plain")
            if not K.conv3x3_nhwc_infer_ok((B, H, W, C), dtype, m, variant):
                return None
            steps.append(("conv", m, variant))
            C, dtype = m.out_channels + (-m.out_channels % 4), torch.bfloat16
            if up:
                steps.append(("shuffle_gelu", None, None))
                H, W, C = 2 * H, 2 * W, C // 4
                i += 2
        elif isinstance(m, nn.GELU) and m.approximate == "none":
            steps.append(("gelu", None, None))
        else:
            return None
        i += 1
    return steps


def _tail_nhwc(seq, x, steps):
    """The reconstruction tail (DSwinT_arch.py:426-437) on the NHWC stream, with or without autograd: every convolution on the
    matrix cores through the NHWC kernel, PixelShuffle as the addressing of the GELU pass that follows it (the
    convolution before it emits its channels in sub-pixel order), no NCHW copy before the last step.  Returns NCHW."""
    y = x
    for kind, m, variant in steps:
        if kind == "conv":
            y = K.conv3x3_nhwc_variant(y, m, variant)
        elif kind == "shuffle_gelu":
            y = K.shuffle2_gelu_nhwc(y)
        else:
            y = K.gelu(y)
    last = [m for m in seq if isinstance(m, nn.Conv2d)][-1]
    return y[..., :last.out_channels].permute(0, 3, 1, 2)


def _reflect_pad_to(x, divisor):
    _, _, h, w = x.shape
    ph = (divisor - h % divisor) % divisor
    pw = (divisor - w % divisor) % divisor
    return F.pad(x, (0, pw, 0, ph), "reflect") if (ph or pw) else x


def _init_weights(m):
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


class DSWTransformer(nn.Module):
    """The DSWT network, `MSFTransformer` of DSwinT_arch.py:376-505.

    One deliberate difference: the reference's `update_temperature` tests for an undefined class
    (`SparseDynamicWindowAttention`, DSwinT_arch.py:483) and raises NameError on every training
    forward; here the cosine annealing it describes is applied to the DS-MSA modules."""

    def __init__(self, block_type, n_blocks, n_sub_blocks, levels, window_size, dim, level_dim, n_heads,
                 n_heads_fuse, feature_dim, scale_factor, ffn_scale=2, drop_path=0.1, annealing_itr=250000):
        super().__init__()
        self.LHSA_levels = levels
        self.dim = dim
        self.feature_dim = feature_dim
        self.scale_factor = scale_factor
        self.window_size = window_size
        self.levels = levels
        self.annealing_itr = annealing_itr
        self.decay_rate = (0.1 / 1.0) ** (1 / annealing_itr)
        self.feature_extractor = nn.Conv2d(3, dim, 3, 1, 1)
        if block_type == "default":
            self.layers = nn.Sequential(*[DSWTResidualBlock(n_sub_blocks=n_sub_blocks, levels=levels, dim=dim,
                                                            level_dim=level_dim, window_size=window_size,
                                                            n_heads=n_heads, n_heads_fuse=n_heads_fuse,
                                                            ffn_scale=ffn_scale, drop_path=drop_path)
                                          for _ in range(n_blocks)])
        self.feature_transition = LastChannelConv2D(dim, dim, 3, 1, 1)
        self.img_reconstruction = _reconstruction_tail(dim, feature_dim, scale_factor, last_div=4)
        self.register_buffer("current_iteration", torch.tensor(0, dtype=torch.long))
        self.apply(_init_weights)

    def padding(self, x):
        divisor = self.window_size if self.levels <= 0 else self.window_size * (2 ** (self.levels - 1))
        return _reflect_pad_to(x, divisor)

    def update_temperature(self):
        if self.current_iteration < self.annealing_itr:
            progress = float(self.current_iteration) / self.annealing_itr
            temp = 0.1 + 0.5 * (1.0 - 0.1) * (1 + math.cos(math.pi * progress))
            for module in self.modules():
                if isinstance(module, dswin.DeformableSparseWindowAttention):
                    module.temperature = temp
        self.current_iteration += 1

    def forward(self, x):
        B, C, H, W = x.shape
        # a forward that runs inside a backward pass is activation checkpointing recomputing this one
        # (toolkit/model_helper2.py:115): the step has been counted already
        if self.training and torch._C._current_graph_task_id() == -1:
            self.update_temperature()
        x = self.padding(x)
        # Without autograd under bf16 autocast the stem and the reconstruction tail stay on NHWC tensors and on the
        # package's convolution kernel (NHWC_ENDS): no NCHW <-> NHWC copies around them, the stem writes the fp32
        # stream directly (3 input channels padded to 4), the tail never materialises an NCHW tensor.  With autograd
        # the same path runs (NHWC_ENDS_TRAINING): the weight gradients of the tail's 16- to 64-channel layers have a
        # kernel of their own (a three-row strip of x through LDS, all nine taps in one workgroup).
        nhwc_ends = (NHWC_ENDS and (NHWC_ENDS_TRAINING or not torch.is_grad_enabled()) and x.is_cuda and x.dtype == torch.float32
                     and K.autocast_dtype(x.dtype) == torch.bfloat16)
        xs = None
        if nhwc_ends:
            xs = F.pad(x.permute(0, 2, 3, 1), (0, -x.shape[1] % 4))
            if not K.conv3x3_nhwc_infer_ok(xs.shape, xs.dtype, self.feature_extractor, "pad_in4"):
                xs = None
            elif torch.is_grad_enabled():
                xs = K.conv3x3_nhwc_variant(xs, self.feature_extractor, "pad_in4").float()
            else:
                xs = K.conv3x3_nhwc_infer(xs, self.feature_extractor, "pad_in4", out_dtype=torch.float32)
        if xs is None:
            x = self.feature_extractor(x)
            x = x.permute(0, 2, 3, 1).contiguous()
            if x.is_cuda and x.dtype in (torch.bfloat16, torch.float16):
                # under autocast the stem hands over bf16; the residual stream between the blocks is kept in
                # fp32 (what it becomes after the first ReZero residual anyway, DSwinT_arch.py:355), so every
                # norm and residual of every block runs in its kernel and no gradient is cast at a block boundary
                x = x.float()
        else:
            x = xs
        x = self.feature_transition(self.layers(x), residual=x)
        steps = _tail_plan(self.img_reconstruction, x) if nhwc_ends else None
        if steps is not None:
            x = _tail_nhwc(self.img_reconstruction, x, steps)
        else:
            x = self.img_reconstruction(x.permute(0, 3, 1, 2).contiguous())
        return x[:, :, :H * self.scale_factor, :W * self.scale_factor]


# ------------------------------------------------------------------------------------------------
# MSFT
# ------------------------------------------------------------------------------------------------
DSMSA_BF16_INPUT = True   # inference under bf16 autocast: the norm in front of DS-MSA writes bf16 (False: fp32, as the reference's norm does)
NHWC_ENDS = True     # False: DSWTransformer's stem and reconstruction tail always run as stock NCHW convolutions
NHWC_ENDS_TRAINING = True    # False: with autograd the two ends run as stock NCHW convolutions (see DSWTransformer.forward)
NHWC_STREAM = True   # False: MSFTransformer runs its trunk module by module on NCHW tensors, as the reference does


def _pointwise(conv):
    """a 1x1 convolution as the projection of token rows it is: weight [Co, Ci, 1, 1] -> [Co, Ci]"""
    return _Stacked(conv.weight.view(conv.out_channels, conv.in_channels), conv.bias)


def _conv3x3_nhwc_plus(x, conv, resid):
    """conv(x) + resid on NHWC tensors: in the convolution's epilogue where the kernel takes the call"""
    y = K.conv3x3_nhwc_residual(x, conv, resid) if resid.dtype == torch.float32 else None
    return y if y is not None else _conv3x3_nhwc(x, conv) + resid


def _conv3x3_nhwc(x, conv):
    y = K.conv3x3_nhwc(x, conv)   # bf16 autocast: NHWC kernels (fwd and bwd), no layout copies
    if y is not None:
        return y
    return conv(x.permute(0, 3, 1, 2).contiguous()).permute(0, 2, 3, 1).contiguous()


class MSFBlock_3(nn.Module):
    """msft_arch.py:169-210: a pyramid of PatchMerging -> ViTBlock -> PatchUnMerging levels fused
    with the full-resolution map by one more ViTBlock."""

    def __init__(self, levels, window_size, dim, level_dim, n_heads, n_heads_fuse, ffn_scale=2, drop_path=0.0):
        super().__init__()
        self.levels = levels
        level_dim = dim // 2
        self.init_ds = nn.Conv2d(dim, level_dim, 1, 1)
        self.level_ds = nn.ModuleList([nn.Conv2d(level_dim // (4 ** i), level_dim // (4 ** (i + 1)), 1, 1)
                                       for i in range(levels - 1)])
        self.level_layer = nn.ModuleList([
            nn.Sequential(msft.PatchMerging(dim=level_dim, scale=2 ** i),
                          msft.ViTBlock(window_size, level_dim, n_heads=n_heads, ffn_scale=ffn_scale, drop_path=drop_path),
                          msft.PatchUnMerging(dim=level_dim, scale=2 ** i))
            for i in range(1, levels)])
        total = sum(level_dim // (4 ** i) for i in range(1, levels))
        self.post_level_fuse = nn.Conv2d(total, total, 3, 1, 1)
        self.fuse = nn.Sequential(
            msft.ViTBlock(window_size, dim + total, n_heads=n_heads_fuse, ffn_scale=ffn_scale, drop_path=drop_path),
            nn.Conv2d(dim + total, dim, 1, 1))
        self.rezero = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        maps = []
        z = self.init_ds(x)
        for i in range(self.levels - 1):
            z = self.level_ds[i](z)
            maps.append(self.level_layer[i](z))
        lv = self.post_level_fuse(torch.cat(maps, dim=1))
        return torch.addcmul(x, self.rezero, self.fuse(torch.cat([x, lv], dim=1)))

    def forward_nhwc(self, x):
        """The same function on the channels-last fp32 stream x [B, H, W, dim]: the 1x1 convolutions are
        projections of the token rows, the 3x3 one runs in the NHWC kernels, the transformer blocks work on the
        token rows where they lie, and the closing 1x1 convolution is fused with the ReZero residual when autograd
        is off.  No tensor changes layout inside the block."""
        B, H, W, C = x.shape
        maps = []
        # With autograd, x gets three gradients: the closing residual's dy, its share of the concatenation's and the
        # first projection's.  They meet inside kernels instead of two additions over the stream: dy is read by the
        # backward of the fusing block's first norm (which writes the concatenation's gradient), and the slice of
        # that map by the first projection's input-gradient kernel.
        train = K.GRAD_MEETING and torch.is_grad_enabled() and x.requires_grad
        pend_cat = K.PendingGrad() if train else None
        pend_res = K.PendingGrad() if train else None
        z = K.linear(x, _pointwise(self.init_ds), grad_extra=pend_cat)
        for i in range(self.levels - 1):
            z = K.linear(z, _pointwise(self.level_ds[i]))
            merge, vit, unmerge = self.level_layer[i]
            s = merge.scale
            t = merge.forward_nhwc(z).float()
            maps.append(unmerge.forward_tokens(vit.forward_tokens(t, H // s, W // s), H // s, W // s))
        lv = _conv3x3_nhwc(maps[0] if len(maps) == 1 else torch.cat(maps, dim=-1), self.post_level_fuse)
        t = self.fuse[0].forward_tokens(K.concat_pending(x, lv, pend_cat).view(B, H * W, -1), H, W, grad_extra=pend_res)
        close = _pointwise(self.fuse[1])
        y = K.linear_residual(t, close, x.view(B, H * W, C), self.rezero)
        if y is None:
            y = K.rezero_residual(x.view(B, H * W, C), self.rezero, K.linear(t, close), pending=pend_res)
        return y.view(B, H, W, C)


class MSFTResidualBlock(nn.Module):
    """msft_arch.py:257-273."""

    def __init__(self, n_sub_blocks, levels, window_size, dim, level_dim, n_heads, n_heads_fuse, ffn_scale=2,
                 drop_path=0.):
        super().__init__()
        self.layers = nn.Sequential(*[MSFBlock_3(levels=levels, level_dim=level_dim, dim=dim, window_size=window_size,
                                                 n_heads=n_heads, n_heads_fuse=n_heads_fuse, ffn_scale=ffn_scale,
                                                 drop_path=drop_path) for _ in range(n_sub_blocks)])
        self.out_layer = nn.Conv2d(dim, dim, 3, 1, 1)

    def forward(self, x):
        return self.out_layer(self.layers(x)) + x

    def forward_nhwc(self, x):
        y = x
        for blk in self.layers:
            y = blk.forward_nhwc(y)
        return _conv3x3_nhwc_plus(y, self.out_layer, x)


class MSFTransformer(nn.Module):
    """The MSFT network, `MSFTransformer` of msft_arch.py:318-435 (block_type "default")."""

    def __init__(self, block_type, n_blocks, n_sub_blocks, levels, window_size, dim, level_dim, n_heads,
                 n_heads_fuse, feature_dim, scale_factor, ffn_scale=4, drop_path=0.1):
        super().__init__()
        if block_type != "default":
            raise NotImplementedError("only block_type='default' is built (the reference's dense variant omits a "
                                      "required constructor argument, msft_arch.py:279-286, and cannot be built)")
        self.LHSA_levels = levels
        self.dim = dim
        self.feature_dim = feature_dim
        self.scale_factor = scale_factor
        self.window_size = window_size
        self.levels = levels
        self.feature_extractor = nn.Conv2d(3, dim, 3, 1, 1)
        self.layers = nn.Sequential(*[MSFTResidualBlock(n_sub_blocks=n_sub_blocks, levels=levels, dim=dim,
                                                        level_dim=level_dim, window_size=window_size, n_heads=n_heads,
                                                        n_heads_fuse=n_heads_fuse, ffn_scale=ffn_scale,
                                                        drop_path=drop_path) for _ in range(n_blocks)])
        self.feature_transition = nn.Conv2d(dim, dim, 3, 1, 1)
        self.img_reconstruction = _reconstruction_tail(dim, feature_dim, scale_factor, last_div=2)
        self.apply(_init_weights)

    def padding(self, x):
        divisor = self.window_size if self.levels <= 0 else self.window_size * (2 ** (self.levels - 1))
        return _reflect_pad_to(x, divisor)

    def forward(self, x):
        B, C, H, W = x.shape
        x = self.padding(x)
        nhwc = NHWC_STREAM and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and self.dim % 8 == 0
        # the stem and the reconstruction tail on NHWC tensors and on the package's convolution kernels, as in
        # DSWTransformer.forward (msft_arch.py builds the same two ends): under bf16 autocast, with or without autograd
        nhwc_ends = (nhwc and NHWC_ENDS and (NHWC_ENDS_TRAINING or not torch.is_grad_enabled()) and x.dtype == torch.float32
                     and K.autocast_dtype(x.dtype) == torch.bfloat16)
        xs = None
        if nhwc_ends:
            xs = F.pad(x.permute(0, 2, 3, 1), (0, -x.shape[1] % 4))
            if not K.conv3x3_nhwc_infer_ok(xs.shape, xs.dtype, self.feature_extractor, "pad_in4"):
                xs = None
            elif torch.is_grad_enabled():
                xs = K.conv3x3_nhwc_variant(xs, self.feature_extractor, "pad_in4").float()
            else:
                xs = K.conv3x3_nhwc_infer(xs, self.feature_extractor, "pad_in4", out_dtype=torch.float32)
        if xs is None:
            x = self.feature_extractor(x)
        if nhwc:
            # the trunk on a channels-last fp32 stream (the attention modules work on token rows: every NCHW <-> token
            # copy of the blocks falls away, and the convolutions between them run in the NHWC kernels)
            if xs is None:
                xs = x.permute(0, 2, 3, 1).float().contiguous()
            y = xs
            for blk in self.layers:
                y = blk.forward_nhwc(y)
            y = _conv3x3_nhwc_plus(y, self.feature_transition, xs)
            steps = _tail_plan(self.img_reconstruction, y) if nhwc_ends else None
            if steps is not None:
                x = _tail_nhwc(self.img_reconstruction, y, steps)
                return x[:, :, :H * self.scale_factor, :W * self.scale_factor]
            x = y.permute(0, 3, 1, 2)
            if not torch.is_autocast_enabled("cuda"):
                x = x.contiguous()
        else:
            x = self.feature_transition(self.layers(x)) + x
        x = self.img_reconstruction(x)
        return x[:, :, :H * self.scale_factor, :W * self.scale_factor]


def dswt_from_config(cfg):
    """Build DSWT from the `model:` section of the reference's configs/config.yaml."""
    m = cfg["model"] if "model" in cfg else cfg
    return DSWTransformer(block_type=m["block_type"], n_blocks=m["n_blocks"], n_sub_blocks=m["n_sub_blocks"],
                          levels=m["levels"], window_size=m["window_size"], dim=m["dim"], level_dim=m["level_dim"],
                          n_heads=m["n_heads"], n_heads_fuse=m["n_heads_fuse"], feature_dim=m["feature_dim"],
                          scale_factor=m["scale_factor"])


DSWT_DEFAULT = dict(block_type="default", n_blocks=8, n_sub_blocks=3, levels=1, window_size=16, dim=180,
                    level_dim=96, n_heads=6, n_heads_fuse=4, feature_dim=64, scale_factor=4)
MSFT_CONFIG_C = dict(block_type="default", n_blocks=8, n_sub_blocks=3, levels=2, window_size=8, dim=192,
                     level_dim=96, n_heads=6, n_heads_fuse=4, feature_dim=64, scale_factor=4)
