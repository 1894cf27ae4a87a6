"""CPU restatement of the DSWT hot path (reference: DSwinT_arch.py:95-338).

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.  Plain torch ops, the
same ops in the same order as the reference so that dtype promotion and
rounding follow it; differentiable through torch autograd, so the same
functions give the gradient reference.

Parameters travel as a dict that uses the reference's ``state_dict`` key names
(``qkv.weight``, ``pe.relative_position_bias_table``, ``sparse_discriminator.0.weight`` ...).

parity unpinned: ``roi_align`` below (third-party torchvision op, see package
docstring).  Everything else is pinned by ``tests/golden/*.npz``.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- #
# Relative positional encoding  (DSwinT_arch.py:99-125)
# --------------------------------------------------------------------------- #
def relative_position_index(q_size, k_size=None):
    """int64 [Lq, Lk] lookup into the (2k-1)^2 bias table (DSwinT_arch.py:105-119).

    For q_size == k_size == w this is (qi-ki+w-1)*(2w-1) + (qj-kj+w-1)."""
    k_size = q_size if k_size is None else k_size
    assert q_size <= k_size
    ii, jj = torch.meshgrid(torch.arange(k_size), torch.arange(k_size), indexing="ij")
    grid = torch.stack([ii, jj])                                  # [2, k, k]
    lo = (k_size - q_size) // 2
    cq = grid[:, lo:lo + q_size, lo:lo + q_size].reshape(2, -1)   # centred queries
    ck = grid.reshape(2, -1)
    rel = (cq[:, :, None] - ck[:, None, :]).permute(1, 2, 0).contiguous() + (k_size - 1)
    return rel[..., 0] * (2 * k_size - 1) + rel[..., 1]


def rpe_bias(table, index, scale=1):
    """[1, h, Lq, Lk] bias from the [(2w-1)^2, h] table (DSwinT_arch.py:122-125)."""
    b = scale * table[index]
    return b.permute(2, 0, 1).contiguous().unsqueeze(0)


# --------------------------------------------------------------------------- #
# WindowedAttention helpers  (DSwinT_arch.py:135-172)
# --------------------------------------------------------------------------- #
def window_partition(x, win, keep_batch=False):
    """[B,H,W,C] -> [B*nW, win^2, C]  (DSwinT_arch.py:135-141)."""
    B, H, W, C = x.shape
    t = x.view(B, H // win, win, W // win, win, C).permute(0, 1, 3, 2, 4, 5).contiguous()
    nw = (H // win) * (W // win)
    return t.view(B, nw, win * win, C) if keep_batch else t.view(B * nw, win * win, C)


def reverse_window_partition(x, win, H, W, with_batch=False):
    """inverse of window_partition (DSwinT_arch.py:143-150)."""
    if with_batch:
        x = x.reshape(-1, win * win, x.shape[-1])
    N, _, C = x.shape
    t = x.view(N, win, win, C).view(-1, H // win, W // win, win, win, C)
    return t.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, H, W, C)


def headify(x, num_heads):
    """[N,L,C] -> [N,h,L,C/h]  (DSwinT_arch.py:152-160)."""
    N, L, C = x.shape
    return x.view(N, L, num_heads, C // num_heads).permute(0, 2, 1, 3).contiguous()


def deheadify(x):
    """[N,h,L,d] -> [N,L,h*d]  (DSwinT_arch.py:162-169)."""
    N, _, L, _ = x.shape
    return x.permute(0, 2, 1, 3).contiguous().view(N, L, -1)


def _pair_hash(pair, seed):
    """the 32-bit hash of pair indices (uint32 array) under a 64-bit seed: (pair + seed_lo) ^ seed_hi through the
    murmur3 finaliser (include/dswin_hip.h, dsw_win_attn_fwd / dsw_rezero_dropout_seeded_fwd)"""
    import numpy as np
    seed_lo, seed_hi = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        x = (pair + seed_lo) ^ seed_hi
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x85EBCA6B)
        x ^= x >> np.uint32(13)
        x *= np.uint32(0xC2B2AE35)
        x ^= x >> np.uint32(16)
    return x


def dropout_keep_mask(n_win, num_heads, L, p, seed):
    """The keep decisions of the build's in-kernel attention dropout, [n_win, h, L, L] bool, from the rule
    include/dswin_hip.h states for dsw_win_attn_fwd (torch's own dropout draws cannot be reproduced across
    devices, so - like the Gumbel noise - the randomness is made explicit): element (window, head, q, k) has pair
    index ((window*h + head)*L + q)*L/2 + k/2 mod 2^32; x = (pair + seed_lo) ^ seed_hi goes through the murmur3
    finaliser; key k takes bits 0-15 (k even) or 16-31 (k odd) of x and is kept when they are >= round(p*65536)."""
    import numpy as np
    thr = min(int(round(p * 65536.0)), 65535)
    rows = np.arange(n_win * num_heads * L, dtype=np.uint64).reshape(-1, 1)
    pair = ((rows * np.uint64(L // 2) + np.arange(L // 2, dtype=np.uint64).reshape(1, -1)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x = _pair_hash(pair, seed)
    keep = np.empty((rows.shape[0], L), dtype=bool)
    keep[:, 0::2] = (x & np.uint32(0xFFFF)) >= thr
    keep[:, 1::2] = (x >> np.uint32(16)) >= thr
    return torch.from_numpy(keep).view(n_win, num_heads, L, L)


def elementwise_keep_mask(numel, p, seed):
    """The keep decisions of the seeded dropout on a dense tensor of `numel` elements (the drop_path of the
    reference's ViTBlock, DSwinT_arch.py:352, applied inside the residual kernel): elements 2i, 2i+1 share the hash
    of pair index i mod 2^32, low / high 16 bits, kept when >= round(p*65536).  -> flat bool tensor."""
    import numpy as np
    thr = min(int(round(p * 65536.0)), 65535)
    pair = (np.arange((numel + 1) // 2, dtype=np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x = _pair_hash(pair, seed)
    keep = np.empty(2 * pair.shape[0], dtype=bool)
    keep[0::2] = (x & np.uint32(0xFFFF)) >= thr
    keep[1::2] = (x >> np.uint32(16)) >= thr
    return torch.from_numpy(keep[:numel])


def attention(q, k, v, C, num_heads, attn_drop=0.0, pe=0, training=False, keep=None):
    """softmax(q k^T * (C//h)^-0.5 + pe) v  (DSwinT_arch.py:171-172).  ``keep`` [N,h,L,L] bool makes the
    dropout draw explicit: probabilities are multiplied by keep / (1 - attn_drop), what F.dropout does
    with its own draw."""
    logits = (q @ k.transpose(-2, -1)) * (C // num_heads) ** -0.5 + pe
    probs = logits.softmax(dim=-1)
    if keep is not None:
        return (probs * keep.to(probs.dtype) / (1.0 - attn_drop)) @ v
    return F.dropout(probs, p=attn_drop, training=training) @ v


# --------------------------------------------------------------------------- #
# roi_align  (call site DSwinT_arch.py:297; third party - parity unpinned)
# --------------------------------------------------------------------------- #
def _bilinear_terms(y, x, H, W):
    """torchvision's bilinear_interpolate for sample points y,x (float tensors
    of one shape): the four tap indices, the four weights, and a validity mask
    (a sample outside [-1,H] x [-1,W] contributes 0)."""
    valid = ~((y < -1.0) | (y > H) | (x < -1.0) | (x > W))
    y = y.clamp(min=0.0)
    x = x.clamp(min=0.0)
    y_low = y.long()
    x_low = x.long()
    top = y_low >= H - 1
    y_high = torch.where(top, torch.full_like(y_low, H - 1), y_low + 1)
    y_low = torch.where(top, torch.full_like(y_low, H - 1), y_low)
    y = torch.where(top, y_low.to(y.dtype), y)
    right = x_low >= W - 1
    x_high = torch.where(right, torch.full_like(x_low, W - 1), x_low + 1)
    x_low = torch.where(right, torch.full_like(x_low, W - 1), x_low)
    x = torch.where(right, x_low.to(x.dtype), x)
    ly = y - y_low.to(y.dtype)
    lx = x - x_low.to(x.dtype)
    hy = 1.0 - ly
    hx = 1.0 - lx
    return (y_low, x_low, y_high, x_high), (hy * hx, hy * lx, ly * hx, ly * lx), valid


def roi_align(input, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """RoIAlign as torchvision publishes it, any boxes, differentiable in
    ``input`` only (torchvision has no box gradient either).

    input [B,C,H,W]; boxes [K,5] = (batch, x1, y1, x2, y2); returns
    [K,C,ph,pw].  Under autocast torchvision computes in fp32; callers here
    pass fp32."""
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    ph, pw = output_size
    B, C, H, W = input.shape
    outs = []
    flat = input.reshape(B, C, H * W)
    for n in range(boxes.shape[0]):
        b = int(boxes[n, 0].item())
        off = 0.5 if aligned else 0.0
        x1 = float(boxes[n, 1]) * spatial_scale - off
        y1 = float(boxes[n, 2]) * spatial_scale - off
        x2 = float(boxes[n, 3]) * spatial_scale - off
        y2 = float(boxes[n, 4]) * spatial_scale - off
        rw, rh = x2 - x1, y2 - y1
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / ph, rw / pw
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        count = max(gh * gw, 1)
        py = torch.arange(ph, dtype=torch.float32).view(ph, 1, 1, 1)
        px = torch.arange(pw, dtype=torch.float32).view(1, pw, 1, 1)
        iy = torch.arange(gh, dtype=torch.float32).view(1, 1, gh, 1)
        ix = torch.arange(gw, dtype=torch.float32).view(1, 1, 1, gw)
        ys = (y1 + py * bh + (iy + 0.5) * bh / gh).expand(ph, pw, gh, gw)
        xs = (x1 + px * bw + (ix + 0.5) * bw / gw).expand(ph, pw, gh, gw)
        (yl, xl, yh, xh), wts, valid = _bilinear_terms(ys, xs, H, W)
        acc = 0
        for (yy, xx), wt in zip(((yl, xl), (yl, xh), (yh, xl), (yh, xh)), wts):
            taps = flat[b][:, (yy * W + xx).reshape(-1)].view(C, ph, pw, gh, gw)
            acc = acc + taps * (wt * valid).to(taps.dtype)
        outs.append(acc.sum(dim=(-1, -2)) / count)
    if not outs:
        return input.new_zeros((0, C, ph, pw))
    return torch.stack(outs)


def gather_windows(x_nchw, b_idx, h1, w1, w):
    """roi_align specialised to the only way the reference calls it
    (DSwinT_arch.py:295-297): integer-aligned w x w boxes, output (w,w),
    spatial_scale 1, sampling_ratio -1, aligned False  =>  bin 1 px, one
    sample per output at (h1+i+.5, w1+j+.5).  Vectorised; [Np,C,w,w]."""
    B, C, H, W = x_nchw.shape
    Np = b_idx.numel()
    i = torch.arange(w, dtype=torch.float32)
    ys = (h1.view(-1, 1, 1).float() + i.view(1, w, 1) + 0.5).expand(Np, w, w)
    xs = (w1.view(-1, 1, 1).float() + i.view(1, 1, w) + 0.5).expand(Np, w, w)
    (yl, xl, yh, xh), wts, valid = _bilinear_terms(ys, xs, H, W)
    flat = x_nchw.reshape(B, C, H * W)
    sel = flat[b_idx]                                              # [Np,C,HW]
    out = 0
    for (yy, xx), wt in zip(((yl, xl), (yl, xh), (yh, xl), (yh, xh)), wts):
        idx = (yy * W + xx).view(Np, 1, w * w).expand(Np, C, w * w)
        out = out + sel.gather(2, idx).view(Np, C, w, w) * (wt * valid).unsqueeze(1).to(sel.dtype)
    return out


# --------------------------------------------------------------------------- #
# DS-MSA pieces
# --------------------------------------------------------------------------- #
def gumbel_noise(shape, generator=None, dtype=torch.float32):
    """g = -log(Exp(1)) as F.gumbel_softmax draws it (torch/nn/functional.py,
    called from DSwinT_arch.py:211)."""
    return -torch.empty(shape, dtype=dtype).exponential_(generator=generator).log()


def candidate_indexes(scores, num_win_selected, temperature, training, noise=None, forced_idx=None):
    """Top-k window candidates + straight-through value (DSwinT_arch.py:201-225).

    scores [B,1,Hw,Ww]; ``noise`` is the explicit Gumbel draw [B,Hw*Ww] used
    when ``training`` (the reference lets F.gumbel_softmax draw it).  Returns
    b_idx, h_idx, w_idx, scores_delta (all flat [B*k]) and the soft probs.
    ``forced_idx`` [B,k] (tests of reduced-precision runs): take these cells as
    the top-k instead of ranking, so that values can be compared on the same
    windows when a rounding difference swaps two near-equal candidates."""
    B, _, Hw, Ww = scores.shape
    k = int((Hw * Ww) * num_win_selected)
    flat = scores.view(B, -1)
    # softmax is on CUDA autocast's fp32 list, so under the trainer's autocast it runs in
    # fp32 whatever the activation dtype; with fp32 inputs the casts below change nothing.
    if training:
        probs = ((flat + noise) / temperature).float().softmax(dim=-1)    # gumbel_softmax, hard=False
    else:
        probs = F.softmax((flat / temperature).float(), dim=-1)
    if forced_idx is None:
        top_p, top_i = torch.topk(probs, k, dim=1)
    else:
        top_i = forced_idx.long().view(B, k)
        top_p = probs.gather(1, top_i)
    scores_delta = (torch.ones_like(top_p) - top_p).detach() + top_p
    b_idx = torch.arange(B).view(-1, 1).expand(-1, k).reshape(-1)
    h_idx = (top_i // Ww).reshape(-1)
    w_idx = (top_i % Ww).reshape(-1)
    return b_idx, h_idx, w_idx, scores_delta.view(-1), probs


def cords(b_idx, h_idx, w_idx, offsets, w, H, W):
    """Window top-left corners after the learned shift (DSwinT_arch.py:227-237).
    ``.long()`` truncates toward zero and carries no gradient."""
    h_off = offsets[b_idx, 0, h_idx, w_idx] * w - (w / 2.)
    w_off = offsets[b_idx, 1, h_idx, w_idx] * w - (w / 2.)
    h1 = torch.clamp(h_idx * w + h_off.long(), min=0, max=H - w)
    w1 = torch.clamp(w_idx * w + w_off.long(), min=0, max=W - w)
    return h1, w1, h1 + w, w1 + w


def integrate(shape, delta, b_idx, h1, w1, w):
    """Mean of overlapping window deltas at integer pixels; pixels no window
    covers get 0 (DSwinT_arch.py:239-253)."""
    B, C, H, W = shape
    gi, gj = torch.meshgrid(torch.arange(w), torch.arange(w), indexing="ij")
    pix = ((h1.view(-1, 1, 1) + gi) * W + (w1.view(-1, 1, 1) + gj)).flatten(1)      # [Np, L]
    base = (b_idx * C * H * W).view(-1, 1) + torch.arange(C).view(1, C) * H * W     # [Np, C]
    index = (base.unsqueeze(2) + pix.unsqueeze(1)).view(-1)
    out = torch.zeros(B * C * H * W, dtype=delta.dtype)
    out.scatter_reduce_(0, index, delta.reshape(-1), reduce="mean", include_self=False)
    return out.view(B, C, H, W)


def sparse_discriminator(P, x_nchw, w, prefix="sparse_discriminator."):
    """conv C->4 3x3, SiLU, conv 4->4 kxk stride k, SiLU, conv 4->4 3x3
    (DSwinT_arch.py:180-186)."""
    t = F.conv2d(x_nchw, P[prefix + "0.weight"], P[prefix + "0.bias"], 1, 1)
    t = F.silu(t)
    t = F.conv2d(t, P[prefix + "2.weight"], P[prefix + "2.bias"], stride=w)
    t = F.silu(t)
    return F.conv2d(t, P[prefix + "4.weight"], P[prefix + "4.bias"], 1, 1)


def dsmsa_forward(P, x, w, num_heads, training=False, noise=None, temperature=1.0,
                  num_win_selected=0.25, aux=None, forced=None):
    """DeformableSparseWindowAttention.forward (DSwinT_arch.py:255-317).

    x [B,H,W,C] -> [B,H,W,C].  ``aux`` (a dict) receives the integer
    decisions: candidate (b,h,w), pass mask, passing (b,h,w), h1, w1.
    ``forced`` (a dict, tests of reduced-precision runs only): integer decisions
    to take over instead of deriving them - "cand_idx" [B,k] flat cells,
    "pass_mask" [B*k] bool, "h1"/"w1" [n] corners of the passing windows."""
    x = x.permute(0, 3, 1, 2).contiguous()
    B, C, H, W = x.shape
    sd = sparse_discriminator(P, x, w)
    scores = sd[:, :1]
    gates = sd[:, 1:2].sigmoid()
    offsets = sd[:, 2:].sigmoid()
    forced = forced or {}
    cb, ch, cw, scores_delta, probs = candidate_indexes(scores, num_win_selected, temperature,
                                                        training, noise, forced.get("cand_idx"))
    if aux is not None:
        aux.update(sd=sd.detach(), cand_b=cb, cand_h=ch, cand_w=cw, probs=probs.detach())
    if cb.numel() == 0:
        return x.permute(0, 2, 3, 1).contiguous()
    cand_gates = gates[cb, 0, ch, cw]
    hard = (cand_gates > 0.5).float()
    if "pass_mask" in forced:
        hard = forced["pass_mask"].float()
    gates_delta = (hard - cand_gates).detach() + cand_gates
    keep = hard.bool()
    b_idx, h_idx, w_idx = cb[keep], ch[keep], cw[keep]
    gates_delta, scores_delta = gates_delta[keep], scores_delta[keep]
    if aux is not None:
        aux.update(pass_mask=keep, b_idx=b_idx, h_idx=h_idx, w_idx=w_idx)
    if b_idx.numel() == 0:
        return x.permute(0, 2, 3, 1).contiguous()
    h1, w1, h2, w2 = cords(b_idx, h_idx, w_idx, offsets, w, H, W)
    if "h1" in forced:
        h1, w1 = forced["h1"].long(), forced["w1"].long()
    if aux is not None:
        aux.update(h1=h1, w1=w1)
    before = gather_windows(x, b_idx, h1, w1, w)                   # roi_align, DSwinT_arch.py:297
    tok = F.linear(before.permute(0, 2, 3, 1).contiguous(), P["in_proj.weight"], P["in_proj.bias"])
    q, k, v = tok.view(-1, w * w, 3 * C).chunk(3, dim=-1)
    q, k, v = headify(q, num_heads), headify(k, num_heads), headify(v, num_heads)
    pe = rpe_bias(P["pe.relative_position_bias_table"], P["pe.relative_position_index"])
    after = attention(q, k, v, C, num_heads, pe=pe)
    after = F.linear(deheadify(after).view(-1, w, w, C), P["out_proj.weight"], P["out_proj.bias"])
    after = after.permute(0, 3, 1, 2).contiguous()
    delta = after - before
    gated = delta * gates_delta.view(-1, 1, 1, 1) * scores_delta.view(-1, 1, 1, 1)
    x = x + integrate(x.shape, gated, b_idx, h1, w1, w)
    y = F.conv2d(x, P["out_conv.weight"], P["out_conv.bias"], 1, 1, groups=C)
    return y.permute(0, 2, 3, 1).contiguous()


def wmsa_forward(P, x, w, num_heads, attn_drop=0.0, training=False, keep=None):
    """DSWT WindowedMSA.forward (DSwinT_arch.py:331-338).  The 3C projection is
    headified first and then split per head: head h owns channels
    [h*3d, (h+1)*3d) and q,k,v are the thirds of that slice."""
    B, H, W, C = x.shape
    z = window_partition(x, w)
    qkv = F.linear(z, P["qkv.weight"], P["qkv.bias"])
    q, k, v = headify(qkv, num_heads).chunk(3, dim=-1)
    pe = rpe_bias(P["pe.relative_position_bias_table"], P["pe.relative_position_index"])
    z = attention(q, k, v, C, num_heads, attn_drop=attn_drop, pe=pe, training=training, keep=keep)
    z = F.linear(deheadify(z), P["out_proj.weight"], P["out_proj.bias"])
    return reverse_window_partition(z, w, H, W)


# --------------------------------------------------------------------------- #
# Pieces the HIP kernels are compared against one by one
# --------------------------------------------------------------------------- #
def window_attention_core(qkv, table, w, num_heads, layout, grid=None, attn_drop=0.0, keep=None):
    """Attention core on a packed 3C projection, output de-headified.

    layout "chunked": channels are q|k|v thirds, each split into heads
    (DS-MSA, DSwinT_arch.py:303-304; msft_arch.py:129).
    layout "interleaved": head h owns [h*3d,(h+1)*3d) = q|k|v (DSwinT_arch.py:335).
    qkv is [N, L, 3C] (windows already gathered) or, with ``grid=(B,H,W)``,
    [B,H,W,3C] that is partitioned here; the result has the matching shape
    with C channels.  ``table`` None means no bias (msft)."""
    C = qkv.shape[-1] // 3
    if grid is not None:
        B, H, W = grid
        qkv = window_partition(qkv.reshape(B, H, W, 3 * C), w)
    if layout == "chunked":
        q, k, v = (headify(t, num_heads) for t in qkv.chunk(3, dim=-1))
    else:
        q, k, v = headify(qkv, num_heads).chunk(3, dim=-1)
    pe = 0 if table is None else rpe_bias(table, relative_position_index(w).to(table.device))
    out = deheadify(attention(q, k, v, C, num_heads, attn_drop=attn_drop, pe=pe, keep=keep))
    if grid is not None:
        out = reverse_window_partition(out, w, H, W)
    return out


def select_windows(sd, sig, w, H, W, num_win_selected, temperature, training, noise=None):
    """Candidate top-k, hard gate, compaction and corner arithmetic in one
    place (DSwinT_arch.py:201-237, 280-295), on a given sparse-discriminator
    output ``sd`` [B,4,Hw,Ww] and its sigmoid ``sig`` = sd[:,1:].sigmoid()
    ([B,3,Hw,Ww]: gate, offset_h, offset_w)."""
    cb, ch, cw, scores_delta, probs = candidate_indexes(sd[:, :1], num_win_selected, temperature,
                                                        training, noise)
    cand_gates = sig[cb, 0, ch, cw]
    hard = (cand_gates > 0.5).float()
    gates_delta = (hard - cand_gates).detach() + cand_gates
    keep = hard.bool()
    b_idx, h_idx, w_idx = cb[keep], ch[keep], cw[keep]
    h1, w1, _, _ = cords(b_idx, h_idx, w_idx, sig[:, 1:], w, H, W)
    return dict(cand_b=cb, cand_h=ch, cand_w=cw, pass_mask=keep, b_idx=b_idx, h_idx=h_idx,
                w_idx=w_idx, h1=h1, w1=w1, gates_delta=gates_delta[keep],
                scores_delta=scores_delta[keep], probs=probs)


def scatter_mean_residual(x_nhwc, after, gates_delta, scores_delta, b_idx, h1, w1, w):
    """x + mean over covering windows of (after - roi_align(x)) * g * s, NHWC
    in/out: the DSwinT_arch.py:309-315 block with the layout copies removed.
    ``after`` is [Np, L, C] (token-major, as out_proj leaves it)."""
    B, H, W, C = x_nhwc.shape
    x = x_nhwc.permute(0, 3, 1, 2).contiguous()
    before = gather_windows(x, b_idx, h1, w1, w)
    aft = after.view(-1, w, w, C).permute(0, 3, 1, 2).contiguous()
    gated = (aft - before) * gates_delta.view(-1, 1, 1, 1) * scores_delta.view(-1, 1, 1, 1)
    x = x + integrate(x.shape, gated, b_idx, h1, w1, w)
    return x.permute(0, 2, 3, 1).contiguous()
