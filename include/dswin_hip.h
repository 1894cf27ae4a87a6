/*
 * dswin_hip.h - C ABI of libdswin_hip.so: hand-written gfx950 (MI355X) kernels for
 * the deformable-sparse-window-attention / windowed-MSA hot path of
 * jackwoodleigh/VisionTransformer.
 *
 * The reference is pure PyTorch and has no FFI of its own; each entry point below
 * replaces the torch / torchvision call sequence cited next to it.  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (nothing is allocated,
 *     freed or retained by the library); `stream` is a hipStream_t passed as void*;
 *   - every function only enqueues work on `stream` and returns at once, never
 *     synchronises, keeps no mutable global state and may be called concurrently
 *     from several host threads (forward thread and autograd worker threads);
 *   - return value 0 = DSW_OK, otherwise a DSW_E* code; dswin_last_error_string()
 *     gives the calling thread's last message;
 *   - activations are DSW_F32 or DSW_BF16 as stated per argument; integer tensors
 *     are int32; images are NHWC ([B,H,W,C], C fastest) unless stated.
 */
#ifndef DSWIN_HIP_H
#define DSWIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { DSW_F32 = 0, DSW_BF16 = 1 };

enum {
    DSW_OK = 0,
    DSW_EINVAL = 1,      /* bad shape / null pointer / inconsistent arguments      */
    DSW_EUNSUPPORTED = 2, /* window size, head dim or dtype outside the built range */
    DSW_ELAUNCH = 3      /* hipGetLastError() after a launch was not hipSuccess     */
};

/* qkv channel layouts of the packed [.., 3C] projection */
enum {
    DSW_QKV_CHUNKED = 0,     /* q|k|v thirds, each split in heads: DSwinT_arch.py:303-304, msft_arch.py:129 */
    DSW_QKV_INTERLEAVED = 1  /* head h owns [h*3d,(h+1)*3d) = q|k|v:  DSwinT_arch.py:335                   */
};

int dswin_version(void);
const char *dswin_last_error_string(void);

/* ------------------------------------------------------------------------------------------
 * (1) Window selection: softmax / Gumbel-softmax over the windows of each image, top-k,
 *     straight-through value, hard gate, compaction, window-corner arithmetic.
 *     Replaces DSwinT_arch.py:201-225 (candidate_indexes), :280-289 (gate STE + boolean-mask
 *     compaction, which costs the reference two device->host syncs) and :227-237 (cords).
 *
 * sd    [B,4,Hw,Ww] sparse-discriminator output (channel 0 = scores), dtype sd_dtype
 * sig   [B,3,Hw,Ww] sigmoid of channels 1..3 as torch computed it (gate, offset_h, offset_w),
 *       same dtype; the comparisons and the offset arithmetic are done in that dtype with one
 *       rounding per op, like the reference
 * noise [B,Hw*Ww] fp32 Gumbel noise -log(Exp(1)) (training) or NULL (eval: softmax(s/tau))
 * k     windows kept per image = int(Hw*Ww*num_win_selected), 1 <= k <= Hw*Ww
 * outputs: probs [B,Hw*Ww] fp32; cand_idx [B,k] flat window index by descending probability
 *   (equal logits: lower index first), cand_p [B,k] fp32 its probability; cand_pass [B,k] 0/1 =
 *   gate > 0.5;
 *   n_pass [1]; win_b / win_cell / win_h1 / win_w1 [B*k] for the passing windows in the
 *   reference's order (image ascending, then top-k rank), entries >= n_pass set to zero (a
 *   harmless window for callers that run on all B*k slots without reading n_pass on the host);
 *   scores_delta / gates_delta [B*k] fp32 straight-through values ((1-p)+p and (1-g)+g), zero
 *   from n_pass on;
 *   cellmap [B,Hw,Ww] = compact window id of the window that started from that cell, or -1.
 */
int dsw_select_fwd(const void *sd, const void *sig, int sd_dtype, const float *noise, float tau,
                   int B, int Hw, int Ww, int k, int w, int H, int W,
                   float *probs, int *cand_idx, float *cand_p, int *cand_pass, int *n_pass,
                   int *win_b, int *win_cell, int *win_h1, int *win_w1,
                   float *scores_delta, float *gates_delta, int *cellmap, void *stream);

/* Backward of (1): the straight-through gradients.  d_scores_delta / d_gates_delta [n_win]
 * fp32 are the incoming gradients of the two straight-through values; d_scores [B,Hw*Ww] fp32
 * receives the softmax-Jacobian product (probs*(dp - sum(probs*dp)))/tau, d_gate [B,Hw*Ww]
 * fp32 the gradient of sig channel 0 (zero for windows that did not pass).  The offsets get no
 * gradient, as in the reference (.long() at DSwinT_arch.py:232-233).  n_win is the number of window
 * slots handed over: the pass count read on the host, or all B*k slots with n_live (may be NULL) = the
 * device counter n_pass of dsw_select_fwd - slots from *n_live on are ignored, so a caller that never
 * reads the count on the host (DSwinT_arch.py:291 does, through numel()) gets the same gradients. */
int dsw_select_bwd(const float *probs, const float *d_scores_delta, const float *d_gates_delta,
                   const int *win_b, const int *win_cell, int n_win, float tau,
                   int B, int Hw, int Ww, float *d_scores, float *d_gate, const int *n_live, void *stream);

/* dsw_select_fwd accepts sig = NULL: the three sigmoids of sd[:, 1:] (DSwinT_arch.py:269-270, :280)
 * are then taken inside the kernels, rounded to the activation dtype like the tensor op.
 * dsw_select_bwd_sd is the backward for that case, continued through the sigmoid: it writes the
 * gradient of sd itself, d_sd [B,4,Hw,Ww] in sd's dtype (channel 0: logits, channel 1: gate,
 * channels 2,3: zero - the offsets get no gradient, DSwinT_arch.py:232-233). */
int dsw_select_bwd_sd(const float *probs, const float *d_scores_delta, const float *d_gates_delta,
                      const int *win_b, const int *win_cell, int n_win, float tau, int B, int Hw,
                      int Ww, const void *sd, int sd_dtype, void *d_sd, const int *n_live, void *stream);

/* ------------------------------------------------------------------------------------------
 * (2) Bilinear window gather.  Replaces the NHWC->NCHW copy (DSwinT_arch.py:259),
 *     torchvision.ops.roi_align(x, rois, (w,w)) (:297) and the NCHW->NHWC copy (:303):
 *     out[n, i*w+j, :] = 2x2 box mean of x[b] at (h1+i, w1+j) with torchvision's border rule
 *     (last row / column: 2x1 mean, corner: the pixel).
 * x [B,H,W,C] x_dtype -> out [n_win, w*w, C] out_dtype.
 * Row strides (ld_*, here and in (4)): elements between consecutive token rows of the window
 * matrices, a multiple of 4 and >= C; 0 = dense.  Outputs with ld > C get zero padding columns.
 */
int dsw_gather_fwd(const void *x, int x_dtype, const int *win_b, const int *win_h1,
                   const int *win_w1, int n_win, int B, int H, int W, int C, int w,
                   void *out, int out_dtype, int ld_out, void *stream);

/* Backward of (2), atomic-free and bitwise reproducible: every pixel gathers from the <= 4
 * windows that cover each of its <= 4 sample positions (torchvision's roi_align backward
 * scatters with atomicAdd).  dout [n_win, w*w, C] -> dx [B,H,W,C].
 * accumulate = 0: every element of dx is written.  accumulate != 0: the result is added into the
 * dx the caller passes (the gradient another consumer of x already produced) and pixels that no
 * window touches are neither read nor written. */
int dsw_gather_bwd(const void *dout, int dout_dtype, const int *cellmap, const int *win_h1,
                   const int *win_w1, int n_win, int B, int H, int W, int C, int w, int ld_dout,
                   void *dx, int dx_dtype, int accumulate, void *stream);

/* ------------------------------------------------------------------------------------------
 * (4) Gated delta, mean of overlapping windows, residual add.  Replaces DSwinT_arch.py:308-315:
 *     delta = after - roi_align(x); gated = delta * gates_delta * scores_delta;
 *     y = x + scatter_reduce(mean) of gated at integer pixels (uncovered pixels: y = x).
 *     Pixel-centric, no atomics: a pixel is covered by at most 4 windows (2 per axis), found
 *     through cellmap; they are summed in ascending window id, then divided by their count.
 * x [B,H,W,C]; after [n_win, w*w, C] (token-major, as out_proj leaves it) -> y [B,H,W,C].
 */
int dsw_scatter_mean_fwd(const void *x, int x_dtype, const void *after, int after_dtype,
                         const float *gates_delta, const float *scores_delta,
                         const int *cellmap, const int *win_h1, const int *win_w1, int n_win,
                         int B, int H, int W, int C, int w, int ld_after, void *y, int y_dtype,
                         void *stream);

/* The same for a selection whose window count lives on the device (n_live, one int): when it holds 0 nothing is
 * written - y stays as it was - and the pass over the map is saved; the caller's next kernel must then read x instead
 * (dsw_dwconv3x3_fwd_src / dsw_dwconv3x3_residual_fwd_src with x_if_zero = x), which is what DS-MSA's early return
 * (DSwinT_arch.py:291-292: no window passed, return x) amounts to without a host read. */
int dsw_scatter_mean_fwd_live(const void *x, int x_dtype, const void *after, int after_dtype,
                              const float *gates_delta, const float *scores_delta,
                              const int *cellmap, const int *win_h1, const int *win_w1, int n_win,
                              int B, int H, int W, int C, int w, int ld_after, void *y, int y_dtype,
                              const int *n_live, void *stream);

/* Backward of (4).  dy [B,H,W,C] -> dx [B,H,W,C] (identity path + the -roi_align path),
 * d_after [n_win, w*w, C], d_gates_delta / d_scores_delta [n_win] fp32.
 * partial: caller-provided scratch of 2*n_win*w floats (per window-row partial sums, reduced
 * in a fixed order).  accumulate_dx != 0: dx is added into the buffer passed instead of written.
 * n_live (may be NULL): device counter of the passing windows when n_win counts all B*k slots (see
 * dsw_select_bwd); the slots from *n_live on get zero d_after rows and zero d_gates / d_scores. */
int dsw_scatter_mean_bwd(const void *dy, int dy_dtype, const void *x, int x_dtype,
                         const void *after, int after_dtype, const float *gates_delta,
                         const float *scores_delta, const int *cellmap, const int *win_b,
                         const int *win_h1, const int *win_w1, int n_win,
                         int B, int H, int W, int C, int w, int ld_after,
                         void *dx, int dx_dtype, int accumulate_dx, void *d_after, int d_after_dtype,
                         int ld_d_after, float *d_gates_delta, float *d_scores_delta, float *partial,
                         const int *n_live, void *stream);

/* The same backward with dx = dx_add + (the gradient): dx_add [B,H,W,C], of dx's dtype and not dx itself, is a
 * gradient that reaches the same tensor by another route - the residual around the module,
 * DSwinT_arch.py:356 - read here instead of being added by a pass of its own; it is left untouched. */
int dsw_scatter_mean_bwd_add(const void *dy, int dy_dtype, const void *x, int x_dtype,
                             const void *after, int after_dtype, const float *gates_delta,
                             const float *scores_delta, const int *cellmap, const int *win_b,
                             const int *win_h1, const int *win_w1, int n_win,
                             int B, int H, int W, int C, int w, int ld_after,
                             void *dx, int dx_dtype, const void *dx_add, void *d_after, int d_after_dtype,
                             int ld_d_after, float *d_gates_delta, float *d_scores_delta, float *partial,
                             const int *n_live, void *stream);

/* ------------------------------------------------------------------------------------------
 * (3)+(5) Fused window attention on MFMA: softmax(q k^T * scale + bias) v per (window, head),
 *     with window partition / headify / chunk / deheadify / reverse done as address arithmetic.
 *     Replaces DSwinT_arch.py:135-172 + :122-125 (bias expansion) as used at :303-308 and
 *     :333-337, and msft_arch.py:94-125 + :136 (SDPA + LePE) as used at :129-139.
 *
 * qkv    [rows, 3C]; rows = n_img*H*W tokens.  grid mode (H,W > 0): the tokens of image b are
 *        its H*W pixels row-major and windows are cut out by address arithmetic; packed mode
 *        (H = W = 0): n_img counts windows and each window's w*w tokens are contiguous.
 * bias_table [(2w-1)^2, heads] fp32 relative-position table or NULL (msft: no bias).
 * lepe_w [C,9] / lepe_b [C] fp32 depthwise 3x3 (zero padded at the window edge) applied to V
 *        and added to the output (msft_arch.py:113-125, :136), or NULL.
 * out    [rows, C] same token order as qkv, dtype = qkv dtype.
 * lse    [n_windows, heads, w*w] fp32 log-sum-exp of every softmax row (saved for backward).
 * ld_qkv, ld_out: row strides in elements of qkv (>= 3C) and out (>= C), even; 0 = dense rows.
 *        With ld_out > C the columns C..ld_out-1 of every out row are set to zero (rows padded
 *        to the widths the GEMM library is fast at go to the next projection as they are).
 * dropout_p, dropout_seed: attention dropout on the softmax probabilities (DSwinT_arch.py:172: F.dropout in
 *        training; msft_arch.py:136: SDPA's dropout_p, applied in eval() too).  No mask tensor: the keep
 *        decision of element (window, head, query q, key k) is a hash of the seed and the element's position
 *        that forward and backward both evaluate - pair index ((window*heads + head)*L + q)*L/2 + k/2 mod
 *        2^32, x = (pair + seed_lo) ^ seed_hi put through the murmur3 finaliser (x ^= x>>16; x *= 0x85EBCA6B;
 *        x ^= x>>13; x *= 0xC2B2AE35; x ^= x>>16); key k takes bits 0-15 (k even) or 16-31 (k odd) and is kept
 *        when they are >= round(p * 65536); kept probabilities are scaled by 1/(1-p).  p in [0,1); 0 = none.
 *        The backward must be given the p and seed of its forward.
 * Supported: w in {8,16}; head dim C/heads <= 64; dtype f32 (exact f32 MFMA) or bf16.
 */
int dsw_win_attn_fwd(const void *qkv, int dtype, int layout, int n_img, int H, int W, int w,
                     int C, int heads, int ld_qkv, int ld_out, const float *bias_table,
                     float scale, const float *lepe_w, const float *lepe_b, void *out,
                     float *lse, float dropout_p, unsigned long long dropout_seed, const int *n_live,
                     void *stream);
/* n_live (here and below; NULL = all): a counter on the device; packed windows (H = W = 0) with index >= *n_live
 * are switched off - DS-MSA hands over a slot for every top-k candidate and leaves the number of windows that
 * passed the gate on the device (DSwinT_arch.py:280-292 reads it on the host instead).  A switched-off window
 * costs no arithmetic: its rows of the result are set to zero (so that later readers find finite values) and it
 * adds nothing to any sum.  dsw_linear / dsw_linear_wgrad take the same counter as live_units with the rows per
 * unit. */

/* bytes of scratch dsw_win_attn_bwd needs (per-(window,head) partial bias-table and LePE
 * weight gradients, reduced in a fixed order by a second kernel inside the call). */
size_t dsw_win_attn_bwd_workspace(int n_img, int H, int W, int w, int C, int heads,
                                  int has_bias, int has_lepe);

/* Backward of (3).  dout [rows, C] -> dqkv [rows, 3C] (same layout as qkv), d_bias_table
 * [(2w-1)^2, heads] fp32 (NULL if no bias), d_lepe_w [C,9] / d_lepe_b [C] fp32 (NULL if no
 * LePE).  `out` is the forward output as it was returned (with LePE the pre-pass removes that
 * term itself).  ld_qkv: row stride of qkv AND dqkv; ld_out, ld_dout: row strides of out and
 * dout; 0 = dense.  With ld_qkv > 3C the padding columns of every dqkv row are set to zero. */
int dsw_win_attn_bwd(const void *qkv, const void *out, const void *dout, int dtype, int layout,
                     int n_img, int H, int W, int w, int C, int heads, int ld_qkv, int ld_out,
                     int ld_dout, const float *bias_table, float scale, const float *lepe_w,
                     const float *lepe_b, const float *lse, void *dqkv, float *d_bias_table,
                     float *d_lepe_w, float *d_lepe_b, void *workspace, size_t workspace_bytes,
                     float dropout_p, unsigned long long dropout_seed, const int *n_live, void *stream);

/* ------------------------------------------------------------------------------------------
 * (5) MSFT multi-scale reshapes, one pass each, no intermediate copies.
 *
 * dsw_unshuffle_nhwc_ln_fwd: PixelUnshuffle(s) + NCHW->NHWC copy + LayerNorm
 *   (msft_arch.py:45-50; with s = 1 also the NCHW->token copy + LayerNorm of
 *   msft_arch.py:68-70 and :160-162).  x [B,c,H,W] x_dtype -> y [B,H/s,W/s,c*s*s] y_dtype,
 *   normalised over the last dim with gamma/beta (fp32) and eps; mean / rstd
 *   [B*(H/s)*(W/s)] fp32 are saved for the backward.
 * dsw_unshuffle_nhwc_fwd: the same reshape without the LayerNorm.
 * dsw_shuffle_nchw_fwd: NHWC->NCHW copy + PixelShuffle(s) (msft_arch.py:54, :74-76, :166):
 *   t [B,H,W,C] -> y [B,C/(s*s),H*s,W*s].
 * The two reshapes are each other's adjoint, so the backward of one is the forward of the
 * other; dsw_unshuffle_nhwc_ln_bwd adds the LayerNorm backward: dy [B,H/s,W/s,c*s*s] and the
 * forward input x -> dx [B,c,H,W], d_gamma / d_beta [c*s*s] fp32.  partial: caller scratch of
 * dsw_unshuffle_blocks(B,H,W,s) * 2 * c*s*s floats (per-block sums, reduced in a fixed order).
 */
int dsw_unshuffle_nhwc_ln_fwd(const void *x, int x_dtype, int B, int c, int H, int W, int s,
                              const float *gamma, const float *beta, float eps, void *y,
                              int y_dtype, float *mean, float *rstd, void *stream);
int dsw_unshuffle_nhwc_ln_bwd(const void *dy, const void *x, int dtype, int B, int c, int H,
                              int W, int s, const float *gamma, const float *mean,
                              const float *rstd, void *dx, float *d_gamma, float *d_beta,
                              float *partial, void *stream);
int dsw_unshuffle_blocks(int B, int H, int W, int s);
int dsw_shuffle_nchw_fwd(const void *t, int dtype, int B, int H, int W, int C, int s, void *y,
                         void *stream);
int dsw_unshuffle_nhwc_fwd(const void *x, int dtype, int B, int c, int H, int W, int s, void *y,
                           void *stream);

/* ------------------------------------------------------------------------------------------
 * Convolutions that sit on the DS-MSA path, NHWC, so that the reference's NHWC->NCHW round trip
 * (DSwinT_arch.py:259, :317) disappears.  Weights and their gradients are fp32.
 *
 * identity_if_zero (forward, may be NULL): a device counter, e.g. dsw_select_fwd's n_pass; when it holds 0
 * the kernel copies x to y instead (the early return of DSwinT_arch.py:291-292 without a host read); the
 * backward then passes dy through as dx and gives zero d_weight / d_bias.
 * dsw_dwconv3x3_fwd/bwd: depthwise 3x3, stride 1, zero padding 1 (out_conv, DSwinT_arch.py:198,
 *   :317).  weight [C,9], bias [C] or NULL.  bwd writes dx, d_weight [C,9], d_bias [C];
 *   partial: caller scratch of (dsw_dwconv3x3_bwd_blocks(B,H,W) + 1) * C * 10 floats.
 * dsw_sparse_disc_fwd/bwd: the sparse discriminator (DSwinT_arch.py:180-186, :266) =
 *   conv C->4 3x3 p1 (w0 [4,C,3,3], b0) + SiLU + conv 4->4 k=w stride w (w2 [4,4,w,w], b2) + SiLU
 *   + conv 4->4 3x3 p1 (w4 [4,4,3,3], b4).  x [B,H,W,C] -> sd [B,4,H/w,W/w] (NCHW, sd_dtype);
 *   t0 [B,H,W,4] fp32 and u [B,4,H/w,W/w] fp32 (the two pre-activation tensors) are saved for
 *   the backward.  bwd: dsd -> dx [B,H,W,C] and the six parameter gradients; scratch: caller
 *   buffer of dsw_sparse_disc_bwd_scratch(B,H,W,C,w) floats.
 */
int dsw_dwconv3x3_fwd(const void *x, int x_dtype, const float *weight, const float *bias, int B,
                      int H, int W, int C, void *y, int y_dtype, const int *identity_if_zero, void *stream);

/* Inference: the same convolution with the ReZero residual of the block around DS-MSA (DSwinT_arch.py:356,
 * x + rezero3 * sdwa(ln3(x))) taken in its pass: y = resid + *rscale * dwconv(x) (or resid + *rscale * x when
 * *identity_if_zero == 0), fp32 [B][H][W][C]; resid and y 16-byte aligned, rscale one fp32 value on the device. */
int dsw_dwconv3x3_residual_fwd(const void *x, int x_dtype, const float *weight, const float *bias, int B, int H, int W,
                               int C, const float *resid, const float *rscale, float *y, const int *identity_if_zero,
                               void *stream);
/* Both with a second source for the identity case: when *identity_if_zero == 0 the kernels copy x_if_zero (same
 * dtype and shape as x, 16-byte aligned) instead of x - x is then never read (see dsw_scatter_mean_fwd_live). */
int dsw_dwconv3x3_fwd_src(const void *x, int x_dtype, const float *weight, const float *bias, int B, int H, int W, int C,
                          void *y, int y_dtype, const int *identity_if_zero, const void *x_if_zero, void *stream);
int dsw_dwconv3x3_residual_fwd_src(const void *x, int x_dtype, const float *weight, const float *bias, int B, int H, int W,
                                   int C, const float *resid, const float *rscale, float *y, const int *identity_if_zero,
                                   const void *x_if_zero, void *stream);
int dsw_dwconv3x3_bwd_blocks(int B, int H, int W);
int dsw_dwconv3x3_bwd(const void *dy, int dy_dtype, const void *x, int x_dtype, const float *weight,
                      int B, int H, int W, int C, void *dx, int dx_dtype, float *d_weight,
                      float *d_bias, float *partial, const int *identity_if_zero, void *stream);
int dsw_sparse_disc_fwd(const void *x, int x_dtype, const float *w0, const float *b0,
                        const float *w2, const float *b2, const float *w4, const float *b4, int B,
                        int H, int W, int C, int w, float *t0, float *u, void *sd, int sd_dtype,
                        void *stream);
int dsw_sparse_disc_bwd_blocks(int B, int H);
size_t dsw_sparse_disc_bwd_scratch(int B, int H, int W, int C, int w);
int dsw_sparse_disc_bwd(const void *dsd, int dsd_dtype, const void *x, int x_dtype, const float *w0,
                        const float *w2, const float *w4, const float *t0, const float *u, int B,
                        int H, int W, int C, int w, void *dx, int dx_dtype, int accumulate_dx, float *dw0,
                        float *db0, float *dw2, float *db2, float *dw4, float *db4, float *scratch,
                        void *stream);

/* Column sums of a row-major [rows][cols] matrix, out[c] = sum_r x[r][c] in fp32: the bias gradient of
 * the path's Linear layers (torch.nn.Linear backward behind DSwinT_arch.py:188,190,325,327 and
 * msft_arch.py:88,92), summed in a fixed order (bitwise reproducible).  cols % 4 == 0.
 *   partial: caller scratch of dsw_colsum_blocks(rows) * cols floats. */
int dsw_colsum_blocks(long long rows);
int dsw_colsum(const void *x, int dtype, long long rows, int cols, float *out, float *partial, void *stream);

/* dsw_colsum without the caller scratch.  No counterpart in the reference: it exists for callers that cannot
 * allocate device memory around a call - code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  The per-block partial sums live in a static device array inside the library
 * (dsw_colsum_static_max_floats() floats = 4 MiB, held by every GPU the library runs on until the process
 * ends); the call allocates and frees nothing.  dsw_colsum_blocks(rows) * cols must not exceed that size
 * (DSW_EUNSUPPORTED otherwise, nothing launched).  The device the buffers live on must be the current device.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU race; calls in stream order do
 * not.  Same kernels and summation order as dsw_colsum, so the same bits. */
size_t dsw_colsum_static_max_floats(void);
int dsw_colsum_static(const void *x, int dtype, long long rows, int cols, float *out, void *stream);

/* Exponential moving average of a list of parameter tensors, in place:
 *   ema[i][e] = ema[i][e] * beta + one_minus_beta * cur[i][e]
 * replaces the per-parameter Python loop of toolkit/param_ema.py:9-17 (three elementwise launches per
 * parameter) by one launch per 128 tensors, with the same rounding (both products rounded to the tensor
 * dtype, then added): bit-identical to the reference's expression.
 *   ema, cur, numel: HOST arrays of n device pointers / element counts (read during the call only);
 *   every tensor dense, of the one dtype given; beta and one_minus_beta as the reference passes them
 *   (one_minus_beta = the double 1 - beta rounded to float). */
int dsw_ema_update(int n, void *const *ema, const void *const *cur, const long long *numel, int dtype, float beta,
                   float one_minus_beta, void *stream);

/* fc1 of the gated MLP with the gate applied in the projection's epilogue (DSwinT_arch.py:69:
 * fc2(act(fc1_gate(x)) * fc1_content(x)), act = SiLU), inference: out = silu(x Wg^T + bg) * (x Wc^T + bc) in one
 * launch; the [tokens, 2H] pre-activations are never written.  Same bits as dsw_linear on the stacked weight
 * followed by dsw_gated_silu_fwd (both pre-activations are rounded to bf16 before the gate, as the separate
 * projection would have stored them).
 *   a, a_dtype, lda, M, K   as for dsw_linear; K <= 192.
 *   w_p     prepared weight (dsw_weight_prep layout, bf16, rows of ldw elements) of wrows = 2 * ceil32(H) rows in
 *           blocks of 64: gate rows 32c..32c+31, then the content rows of the same hidden columns; zero rows past H.
 *   bias    fp32 [wrows] in the same order (zeros past H), or NULL.
 *   out     bf16 [M][ldo], ldo >= wrows / 2, a multiple of 8; columns 0..wrows/2-1 are written (those past H come
 *           out as zero because their weight rows and biases are zero). */
int dsw_linear_gated_silu(const void *a, int a_dtype, long long lda, long long M, int K, const void *w_p, int wrows, int ldw,
                          const float *bias, void *out, long long ldo, void *stream);

/* dsw_linear / dsw_linear_gated_silu with the nn.RMSNorm in front of them folded in (DSwinT_arch.py:343-345 applied
 * at :355 and :357: wmsa(ln1(x)), mlp1(ln2(x))), inference: c = (a W'^T) / sqrt(mean_k a^2 + eps) + bias per row,
 * where W' = W diag(norm weight) has been prepared by the caller in place of W.  The mean runs over the K values
 * of the row as they lie in memory (fp32 for the fp32 stream), the product over their bf16 roundings: the result
 * differs from norm-then-project by bf16 roundings only (the norm's output is rounded to bf16 there, the scaled
 * weight here).  K <= 192; other arguments as for the plain entry points; no a_copy, no live-row counter. */
int dsw_linear_rmsnorm(const void *a, int a_dtype, long long lda, long long M, int K, float eps, const void *w_p,
                       int wrows, int ldw, const float *bias, int N, void *c, int c_dtype, long long ldc, int ncols,
                       void *stream);
int dsw_linear_rmsnorm_gated_silu(const void *a, int a_dtype, long long lda, long long M, int K, float eps,
                                  const void *w_p, int wrows, int ldw, const float *bias, void *out, long long ldo,
                                  void *stream);

/* The gated MLP of DSWT's transformer block with the RMSNorm in front of it and the ReZero residual around it
 * (DSwinT_arch.py:47-69 applied at :357 with the norm of :345), inference, one launch:
 *   y = x + r * (fc2(silu(fc1_gate(n)) * fc1_content(n)) + b2),  n = x / sqrt(mean_k x^2 + eps) * g   (norm_eps < 0: n = x)
 * x, y: fp32 rows of C values (C % 4 == 0, C <= 192; row strides ldx, ldy in elements), y may not alias x.  The hidden
 * activations never reach memory.  Operand images prepared by the caller (bf16, 16-byte aligned), the hidden dimension
 * in `hidden_chunks` chunks of 32 units (<= 16), zero rows / columns beyond the matrices:
 *   w1c [chunks][64][192]: rows 0..31 of a chunk = the gate rows of its hidden units, rows 32..63 = their content rows,
 *                          each row = W[h][0..C) * g[0..C) (the norm's weight folded in);
 *   b1c [chunks][64] fp32, the same order;
 *   w2c [chunks][192][32]: W2[out][hidden of the chunk], every 16 hidden units in the order 0-3, 8-11, 4-7, 12-15
 *                          (the order in which a lane's MFMA accumulators hold the gated values);
 *   b2 [192] fp32.  r: one float on the device.
 * The gated product is taken on the fp32 pre-activations and rounded to bf16 once, as the second product's operand
 * (the separate kernels, dsw_linear_rmsnorm_gated_silu and dsw_linear_residual, also round gate and content to bf16
 * as they store them: 4e-3 between the two forms). */
/* Its operand images from the modules' fp32 parameters in one launch: gate_w, content_w [H][C], fc2_w [N][H] row-major,
 * biases and norm_w (the RMSNorm weight [C]) may be NULL (zero / one); w1c, b1c, w2c, b2 as described above with
 * ceil(H / 32) chunks. */
int dsw_mlp_gated_prep(const float *gate_w, const float *content_w, const float *gate_b, const float *content_b,
                       const float *fc2_w, const float *fc2_b, const float *norm_w, int H, int C, int N, void *w1c,
                       float *b1c, void *w2c, float *b2, void *stream);
int dsw_mlp_gated_residual(const float *x, long long ldx, long long M, int C, float norm_eps, const void *w1c,
                           const float *b1c, const void *w2c, const float *b2, int hidden_chunks, const float *r,
                           float *y, long long ldy, void *stream);

/* The first projection of msft's MLP with its activation (msft_arch.py:22-25 applied at :33: Linear, then
 * nn.GELU() in its erf form): out = gelu(a W^T + bias) in one launch, bf16.  The pre-activation is rounded to bf16
 * first - what dsw_linear would have stored - so out has the bits of dsw_linear followed by dsw_gelu_fwd.
 *   a ... ncols, a_copy, lda_copy   as for dsw_linear (any K); ncols, ldo multiples of 8.
 *   pre     bf16 [M][ld_pre] or NULL: receives the pre-activation (columns 0..ncols-1), which the backward needs.
 * dsw_linear_gelu_bwd is the other half, on the way out of the second projection's input gradient
 * (the backward of msft_arch.py:26 followed by that of :25): out = (a W^T) * gelu'(pre), bf16, where a is the
 * gradient of the MLP's output, W the prepared transpose of the second projection's weight and pre what the
 * forward call stored: the gradient of the pre-activation, with the bits of dsw_linear followed by dsw_gelu_bwd. */
int dsw_linear_gelu(const void *a, int a_dtype, long long lda, long long M, int K, const void *w_p, int wrows, int ldw,
                    const float *bias, int N, void *out, long long ldo, int ncols, void *pre, long long ld_pre,
                    void *a_copy, long long lda_copy, void *stream);
int dsw_linear_gelu_bwd(const void *a, int a_dtype, long long lda, long long M, int K, const void *w_p, int wrows,
                        int ldw, const void *pre, long long ld_pre, void *out, long long ldo, int ncols, void *a_copy,
                        long long lda_copy, float drop_p, unsigned long long drop_seed, void *stream);

/* The residual dropout of the callers' blocks (msft_arch.py:163-165: x = self.drop_path(branch) + x, an nn.Dropout on
 * the branch; DSwinT_arch.py:355-357) inside the projections on either side of it, with the seeded decisions of
 * dsw_rezero_dropout_seeded_fwd / _bwd (a hash of the seed and the element's index in the dense [M][N] branch):
 *   dsw_linear_residual_dropout: out = resid + *scale * dropout(a W^T + bias), fp32 - the projection that closes the
 *     branch, the dropout and the residual in one launch (arguments as dsw_linear_residual, then drop_p, drop_seed);
 *   dsw_linear_dropout_in: c = dropout(a) W^T for an fp32 a of dense rows [M][K] - the backward of the same: a is the
 *     gradient of the block's output, masked and scaled on its way in (a_copy receives the masked bf16 values: the
 *     operand of dsw_linear_wgrad), w_p the prepared transpose of the closing projection's weight;
 *   dsw_linear_gelu_bwd takes drop_p / drop_seed for the same purpose when a is fp32 (0: no dropout). */
int dsw_linear_dropout_in(const void *a, long long lda, long long M, int K, const void *w_p, int wrows, int ldw, void *c,
                          int c_dtype, long long ldc, int ncols, void *a_copy, long long lda_copy, float drop_p,
                          unsigned long long drop_seed, void *stream);
int dsw_linear_residual_dropout(const void *a, int a_dtype, long long lda, long long M, int K, const void *w_p, int wrows,
                                int ldw, const float *bias, int N, const float *resid, long long ld_resid,
                                const float *scale, float *out, long long ldo, int ncols, float drop_p,
                                unsigned long long drop_seed, void *stream);

/* Window attention with the q|k|v projection inside the kernel, forward only (inference): the nn.Linear of
 * DSwinT_arch.py:334 (WindowedMSA.qkv) or :303 (DeformableSparseWindowAttention.in_proj) followed by the
 * attention of :135-172 as ONE launch; the [tokens, 3C] q|k|v matrix is never written to memory.
 *   out[token][head*d + e] = sum_k softmax_k(q.k * scale + table) v,   q|k|v = x W^T + bias  per token
 * with the roundings of the two separate calls (dsw_linear to bf16, then dsw_win_attn_fwd): projection
 * accumulated in fp32, bias added, rounded to bf16; q scaled and rounded again.
 *   x        bf16 activations, one row per token, row stride ldx elements (a multiple of 8, >= ldw); the
 *            columns C..ldw-1 of every row must hold zeros (finite values suffice: they meet zero weights).
 *            Token addressing as in dsw_win_attn_fwd: H = W = 0 means n_img packed windows of w*w rows, else
 *            rows are the pixels of n_img images of H x W and the windows are cut out by address arithmetic.
 *   w_p      the projection weight as dsw_weight_prep leaves it: bf16 [w_rows >= 3C][ldw], ldw a multiple of
 *            64, zero past column C; row n = output feature n in the q|k|v `layout` (DSW_QKV_*).
 *   bias     fp32 [3C] or NULL;  bias_table fp32 [(2w-1)^2][heads] or NULL;  lse fp32 [n_win][heads][w*w] or NULL.
 *   out      bf16 rows of ld_out elements (0 = C); columns C..ld_out-1 are set to zero.
 * bf16 only, w in {8, 16}, head dim C/heads <= 32: DSW_EUNSUPPORTED otherwise (callers then use the two separate
 * entry points).  No dropout and no LePE term: the modules that need them (msft_arch.py:94-139) keep the
 * separate calls. */
int dsw_win_attn_proj_fwd(const void *x, long long ldx, const void *w_p, int w_rows, int ldw, const float *bias,
                          int layout, int n_img, int H, int W, int w, int C, int heads, int ld_out,
                          const float *bias_table, float scale, void *out, float *lse, const int *n_live, void *stream);

/* One Adam step over a list of fp32 parameter tensors, in place: replaces the optimizer step of the
 * reference's trainer (torch.optim.Adam created at toolkit/init.py:133 with betas (0.9, 0.99) and stepped at
 * toolkit/model_helper2.py:125), which torch runs as eight elementwise passes over the tensor lists, by one
 * pass (4 reads + 3 writes per element), one launch per 64 tensors.  Per element, fp32, in the order and
 * rounding of those passes as torch's GPU kernels compute them (bit-identical results):
 *   m <- fma(one_minus_beta1, g - m, m);  v <- v * beta2;  v <- fma(one_minus_beta2, g * g, v);
 *   d <- sqrt(v) / bias2_sqrt + eps;      p <- fma(step_size, m / d, p)
 * with step_size = -lr / (1 - beta1^t) and bias2_sqrt = sqrt(1 - beta2^t) computed by the caller in
 * double and rounded to float, as torch passes them (t = the step number, from 1).
 *   param, grad, exp_avg, exp_avg_sq, numel: HOST arrays of n device pointers / element counts (read
 *   during the call only); every tensor dense fp32.  No weight decay, amsgrad or maximize. */
int dsw_adam_update(int n, void *const *param, const void *const *grad, void *const *exp_avg, void *const *exp_avg_sq,
                    const long long *numel, float one_minus_beta1, float beta2, float one_minus_beta2,
                    float bias2_sqrt, float eps, float step_size, void *stream);

/* Row copy with dtype conversion between two row strides (in elements): dst[r][c] = src[r][c] for
 * c < cols; with zero_pad != 0 the columns cols..dst_ld-1 of every dst row are set to zero.  Used
 * around the projection GEMMs (the nn.Linear calls at DSwinT_arch.py:303,308,334,337 and
 * msft_arch.py:128,141) to hand the GEMM library rows padded to multiples of 64 elements and to
 * take padded results apart, in place of torch's cast + fill + strided copy.  cols and both row
 * strides are multiples of 4 elements. */
int dsw_cast_rows(const void *src, int src_dtype, long long src_ld, void *dst, int dst_dtype,
                  long long dst_ld, long long rows, long long src_rows, int cols, int zero_pad,
                  void *stream);

/* The projections of the attention modules (the nn.Linear calls at DSwinT_arch.py:303, :308, :334,
 * :337 and msft_arch.py:128, :141) and their input gradients as one bf16 MFMA kernel:
 *
 *   c[m][n] = sum_{k < K} a[m][k] * w_p[n][k]  (+ bias[n] for n < N),   m < M, n < ncols
 *
 * dsw_weight_prep turns an nn.Linear weight [N][K] (fp32 or bf16) into the operands: w_p
 * [ceil64(N)][ceil64(K)] and its transpose w_pt [ceil64(K)][ceil64(N)], bf16, zero outside N x K
 * (either pointer may be NULL).  The forward projection multiplies by w_p, the input gradient
 * dx = dy . W multiplies by w_pt.
 *   a:      [M][lda] fp32 or bf16; fp32 values are rounded to bf16 on load.  K and lda multiples
 *           of 4 elements, base aligned to 4 elements (bf16 rows that are 16-byte aligned, K and lda
 *           multiples of 8, load in half as many instructions).
 *   w_p:    bf16 [wrows][ldw], ldw >= ceil64(K); rows >= wrows read as zero.
 *   bias:   fp32 [N] or NULL.
 *   c:      [M][ldc] fp32 or bf16, columns 0..ncols-1 written (ncols % 4 == 0): with ncols =
 *           ceil64(N) and a zero-padded w_p the padding columns come out as zeros.  accumulate != 0
 *           (fp32 c only): the product is added to what c holds.
 *   a_copy: NULL, or bf16 [M][lda_copy]: receives a rounded to bf16 with the columns K..ceil64(K)-1
 *           zeroed (the operand the weight-gradient GEMM wants), saving a separate cast pass.
 * K <= 192 takes any ncols; K > 192 is run in slabs of 64/128/192 columns of a. */
int dsw_weight_prep(const void *w, int w_dtype, int N, int K, void *w_p, void *w_pt, void *stream);
/* dsw_weight_prep for n weights in ceil(n / 48) launches (the weights of a network after an optimizer step: the
 * Linear layers named above in every block).  w, w_dtype, N, K, w_p, w_pt: HOST arrays of n entries (read during
 * the call only), each entry as the argument of the same name of dsw_weight_prep. */
int dsw_weight_prep_batch(int n, const void *const *w, const int *w_dtype, const int *N, const int *K,
                          void *const *w_p, void *const *w_pt, void *stream);
int dsw_linear(const void *a, int a_dtype, long long lda, long long M, int K, const void *w_p, int wrows,
               int ldw, const float *bias, int N, void *c, int c_dtype, long long ldc, int ncols,
               int accumulate, void *a_copy, long long lda_copy, const int *live_units, int unit_rows, void *stream);
/* live_units (NULL = all rows), unit_rows: rows at and beyond *live_units * unit_rows are switched off (see n_live
 * of dsw_win_attn_fwd): workgroups that hold none of the live rows compute nothing and leave zeros in their rows
 * of c and a_copy (with accumulate: leave c as it is). */
/* The projection that closes a branch of the reference's ViTBlock together with the ReZero residual around it
 * (DSwinT_arch.py:337 or :69 with :355-357, inference): out = resid + scale[0] * (A W^T + bias), fp32 [M][ncols]
 * with row stride ldo, resid fp32 with row stride ld_resid (it may be out itself), scale one fp32 value on the
 * device.  Operands as for dsw_linear.  Saves the bf16 round trip of the branch and the residual pass. */
int dsw_linear_residual(const void *a, int a_dtype, long long lda, long long M, int K, const void *w_p, int wrows,
                        int ldw, const float *bias, int N, const float *resid, long long ld_resid, const float *scale,
                        float *out, long long ldo, int ncols, void *stream);

/* Weight and bias gradient of those projections (torch.nn.Linear backward behind the same call
 * sites): dw[n][k] = sum_m dy[m][n] x[m][k], db[n] = sum_m dy[m][n], fp32, summed in a fixed order.
 *   dy: bf16 [M][ldy], ldy >= ceil64(N), columns N.. zero;  x: bf16 [M][ldx], ldx >= ceil64(K),
 *   columns K.. zero (the a_copy output of dsw_linear has this form).  K % 4 == 0.
 *   dw: fp32 [N][K] dense;  db: fp32 [N] or NULL (needs K < ceil64(K): the sum rides in the first
 *   padding column of x);  workspace: dsw_linear_wgrad_workspace(M, N, K) floats. */
size_t dsw_linear_wgrad_workspace(long long M, int N, int K);
int dsw_linear_wgrad(const void *dy, long long ldy, const void *x, long long ldx, long long M, int N, int K,
                     float *dw, float *db, float *workspace, const int *live_units, int unit_rows, void *stream);
/* live_units, unit_rows as for dsw_linear: the switched-off rows are not read and add nothing to dw / db. */

/* The first convolution of the DS-MSA sparse discriminator (nn.Conv2d(dim, 4, 3, 1, 1),
 * DSwinT_arch.py:181) as a GEMM on the matrix cores, for bf16 runs: with Wf[4 t + o][c] = w0[o][c][t]
 * (t = 3 ky + kx), P = x . Wf^T by dsw_linear gives the 36 products of every pixel,
 * dsw_sd_stencil_fwd gathers them, t0[px][o] = b0[o] + sum_t P[px + off(t)][4 t + o] (zero padding), and
 * dsw_sparse_disc_tail_fwd runs the rest of the discriminator (:182-185) on t0.  Backward:
 * dsw_sparse_disc_tail_bwd returns dt0 and the gradients of the two small convolutions,
 * dsw_sd_gather_bwd lays out G[px][4 t + o] = dt0[px - off(t)][o] (bf16 [pixels][64], columns 36.. zero),
 * and then dx = G . Wf (dsw_linear on Wf^T), dWf = G^T . x and db0[o] = column 16 + o of G summed
 * (dsw_linear_wgrad).
 *   P: fp32 [B*H*W][ldp], ldp >= 36, ldp % 4 == 0.  t0, dt0: fp32 [B][H][W][4].  u: fp32 [B][4][H/w][W/w].
 *   du: scratch of B*4*(H/w)*(W/w) floats. */
int dsw_sd_stencil_fwd(const float *P, int ldp, const float *b0, int B, int H, int W, float *t0, void *stream);
int dsw_sd_gather_bwd(const float *dt0, int B, int H, int W, void *G, void *stream);
int dsw_sparse_disc_tail_fwd(const float *t0, const float *w2, const float *b2, const float *w4, const float *b4,
                             int B, int H, int W, int w, float *u, void *sd, int sd_dtype, void *stream);
int dsw_sparse_disc_tail_bwd(const void *dsd, int dsd_dtype, const float *w2, const float *w4, const float *t0,
                             const float *u, int B, int H, int W, int w, float *du, float *dt0, float *dw2,
                             float *db2, float *dw4, float *db4, void *stream);

/* RMSNorm over channel rows of the fp32 stream (the nn.RMSNorm layers of the reference's ViTBlock,
 * DSwinT_arch.py:343-345, applied at :355-357): out[m][c] = x[m][c] * rstd[m] * weight[c] with
 * rstd[m] = rsqrt(mean_c x[m][c]^2 + eps), written as fp32 or bf16 with row stride ldo >= C (columns
 * C..ldo-1 are set to zero: the projection kernels take such rows as they are).  C % 4 == 0, C <= 256.
 *   rstd: fp32 [M] (NULL when no backward follows).
 * Backward: dx (fp32 [M][C]; with dx_add != NULL, fp32 [M][C], dx = dx_add + the norm's gradient: dx_add is
 * the gradient that reached x through the residual connection around the normalised branch) and per-block partial sums of d(weight): partial is
 * dsw_rmsnorm_blocks(M) x C floats, to be summed over its rows (e.g. dsw_colsum). */
int dsw_rmsnorm_blocks(long long M);
int dsw_rmsnorm_fwd(const float *x, const float *weight, float eps, long long M, int C, void *out, int out_dtype,
                    long long ldo, float *rstd, void *stream);
int dsw_rmsnorm_bwd(const void *dy, int dy_dtype, long long ld_dy, const float *x, const float *weight,
                    const float *rstd, long long M, int C, float *dx, const float *dx_add, float *partial,
                    void *stream);

/* The gated activation of the reference's MLP (DSwinT_arch.py:47-69, fc2(act(fc1_gate(x)) *
 * fc1_content(x))) on h = [gate | content] rows (the two fc1 projections computed as one product):
 * out[m][j] = silu(h[m][j]) * h[m][H + j], j < H; columns H..ldo-1 of out are zeroed.  Backward:
 * dh[m][j] and dh[m][H + j] from dout and h, row stride ldh, columns 2H..ldh-1 zeroed.
 * H and all row strides are multiples of 4 elements; fp32 or bf16 on either side. */
int dsw_gated_silu_fwd(const void *h, int h_dtype, long long ldh, int H, long long M, void *out, int out_dtype,
                       long long ldo, void *stream);
int dsw_gated_silu_bwd(const void *dout, int dout_dtype, long long ld_dout, const void *h, int h_dtype,
                       long long ldh, int H, long long M, void *dh, void *stream);

/* The activation of msft's MLP (msft_arch.py:25: nn.GELU(), the erf form) between its two projections, on rows with
 * their own stride: y[m][j] = gelu(x[m][j]), j < C, columns C..ldy-1 zeroed; backward dx = dy * gelu'(x), columns
 * C..ld_dx-1 zeroed.  One dtype (fp32 or bf16) for all operands; C and the row strides are multiples of 4. */
int dsw_gelu_fwd(const void *x, int dtype, long long ldx, int C, long long M, void *y, long long ldy, void *stream);
int dsw_gelu_bwd(const void *dy, long long ld_dy, const void *x, int dtype, long long ldx, int C, long long M, void *dx,
                 long long ld_dx, void *stream);

/* One up-sampling stage of the reference's reconstruction tail (DSwinT_arch.py:426-437: Conv2d(f, 4f, 3) ->
 * PixelShuffle(2) -> GELU) on NHWC tensors: x [B][H][W][4c] is the convolution's result with its output channels
 * ordered (r1, r2, channel) - nn.PixelShuffle's channel index channel*4 + r1*2 + r2 re-ordered, which the caller does
 * on the weight rows - and y [B][2H][2W][c] = gelu(erf form) of the shuffled tensor.  One dtype (fp32 or bf16); c is a
 * multiple of 4, both tensors dense. */
int dsw_shuffle2_gelu_nhwc(const void *x, int dtype, int B, int H, int W, int c, void *y, void *stream);
/* its backward: dx [B][H][W][4c] from dy [B][2H][2W][c] and the forward's input x (same dtype for all three) */
int dsw_shuffle2_gelu_nhwc_bwd(const void *dy, const void *x, int dtype, int B, int H, int W, int c, void *dx, void *stream);

/* The ReZero residual of the reference's ViTBlock (DSwinT_arch.py:355-357), y = x + r * branch, in one
 * pass: x, y fp32 [M][C]; branch fp32 or bf16 with row stride ldb; r one fp32 value on the device.
 * Backward (dx = dy needs no kernel): d_branch = r * dy in branch's dtype with row stride ld_db (columns
 * C..ld_db-1 zeroed) and partial[i] = block i's share of dr = sum dy * branch; partial holds
 * dsw_rezero_blocks(M, ld_db) floats, summed by the caller.  C and the strides are multiples of 4. */
int dsw_rezero_blocks(long long M, long long ld);
int dsw_rezero_fwd(const float *x, const void *branch, int b_dtype, long long ldb, const float *r, long long M,
                   int C, float *y, void *stream);
int dsw_rezero_bwd(const float *dy, const void *branch, int b_dtype, long long ldb, const float *r, long long M,
                   int C, void *d_branch, long long ld_db, float *partial, void *stream);
/* The same residual with the reference's dropout on the branch in between (`drop_path`, an nn.Dropout,
 * DSwinT_arch.py:352, applied at :355-357): y = x + r * (branch * mask * keep_scale).  mask: dense bytes
 * [M][C], 0 = dropped (the caller draws it, e.g. Bernoulli(1 - p) from torch's generator, so that activation
 * checkpointing replays it); keep_scale = 1 / (1 - p).  Backward as above with dy masked and scaled. */
int dsw_rezero_dropout_fwd(const float *x, const void *branch, int b_dtype, long long ldb, const float *r,
                           const unsigned char *mask, float keep_scale, long long M, int C, float *y, void *stream);
int dsw_rezero_dropout_bwd(const float *dy, const void *branch, int b_dtype, long long ldb, const float *r,
                           const unsigned char *mask, float keep_scale, long long M, int C, void *d_branch,
                           long long ld_db, float *partial, void *stream);
/* The same with the keep decisions taken from a 64-bit seed instead of a mask tensor (no mask is drawn, written or
 * read): elements 2i, 2i+1 of the dense [M][C] matrix share the 32-bit hash of pair index i (mod 2^32) - the
 * murmur3 finaliser of (i + seed_lo) ^ seed_hi - the even element takes its bits 0-15, the odd one bits 16-31, an
 * element is kept when its 16 bits are >= round(p * 65536), and kept values are scaled by 1 / (1 - p); p in [0, 1).
 * Forward and backward evaluate the same function of (seed, position). */
int dsw_rezero_dropout_seeded_fwd(const float *x, const void *branch, int b_dtype, long long ldb, const float *r, float p,
                                  unsigned long long seed, long long M, int C, float *y, void *stream);
int dsw_rezero_dropout_seeded_bwd(const float *dy, const void *branch, int b_dtype, long long ldb, const float *r, float p,
                                  unsigned long long seed, long long M, int C, void *d_branch, long long ld_db, float *partial,
                                  void *stream);

/* Dense 3x3 convolution, stride 1, zero padding 1, on an NHWC tensor (the LastChannelConv2D layers of the
 * reference's model assembly, DSwinT_arch.py:36-44, used at :372 and :420) on the matrix cores: the product kernel of dsw_linear with one k slab per tap, the rows of tap t read shifted by
 * (t / 3 - 1, t % 3 - 1) pixels and as zero outside the image.
 *   x:   [B][H][W][ldx] fp32 or bf16 (rounded to bf16 on load), below 2 GiB.
 *   w_p: bf16 [wrows][ldw] with w_p[co][t * ceil64(C) + ci] = weight[co][ci][t / 3][t % 3], zero elsewhere
 *        (dsw_weight_prep of the tap-major, channel-padded matrix).
 *   out: [B*H*W][ldo] fp32 or bf16, columns 0..ncols-1 written; bias fp32 [N] or NULL.
 *   x_copy: NULL or bf16 [B*H*W][ld_copy]: x rounded to bf16, channels padded to 64 with zeros. */
int dsw_conv3x3_nhwc(const void *x, int x_dtype, long long ldx, int B, int H, int W, int C, const void *w_p,
                     int wrows, int ldw, const float *bias, int N, void *out, int out_dtype, long long ldo,
                     int ncols, void *x_copy, long long ld_copy, void *stream);

/* The same convolution with the residual add behind it (DSwinT_arch.py:373-374, out_layer(layers(x)) + x; :501) in its
 * epilogue: out = resid + conv(x) + bias as fp32; resid rows ld_resid >= ncols floats apart, 16-byte aligned. */
int dsw_conv3x3_nhwc_residual(const void *x, int x_dtype, long long ldx, int B, int H, int W, int C, const void *w_p, int wrows,
                              int ldw, const float *bias, int N, const float *resid, long long ld_resid, float *out,
                              long long ldo, int ncols, void *stream);
/* The same with the bf16 copy of x that dsw_conv3x3_nhwc leaves for the weight gradient (x_copy, ld_copy as there):
 * the training forward of a residual block's closing convolution. */
int dsw_conv3x3_nhwc_residual_copy(const void *x, int x_dtype, long long ldx, int B, int H, int W, int C, const void *w_p,
                                   int wrows, int ldw, const float *bias, int N, const float *resid, long long ld_resid,
                                   float *out, long long ldo, int ncols, void *x_copy, long long ld_copy, void *stream);
/* Its weight and bias gradient: dw[co][t][ci] = sum_px dy[px][co] x[px + off(t)][ci] (tap-major, fp32
 * [Co][9][Ci]), db[co] = sum_px dy[px][co]; dy, x bf16 in zero-padded rows as for dsw_linear_wgrad (x_copy
 * of the forward call has that form).  The input gradient is the same convolution applied to dy with the
 * flipped, transposed weight.  workspace: dsw_conv3x3_wgrad_workspace(B, H, W, Co, Ci) floats.
 * Ci <= 64 on images whose width is a multiple of 64 (the 16- to 64-channel layers of the reconstruction tail,
 * DSwinT_arch.py:430-437, at 128 x 128 and 256 x 256 pixels per patch) takes a kernel of its own: a workgroup owns
 * dw[64 output channels][9 taps][64] and reads a three-row strip of x through LDS once for all nine taps. */
size_t dsw_conv3x3_wgrad_workspace(int B, int H, int W, int Co, int Ci);
int dsw_conv3x3_wgrad(const void *dy, long long ldy, const void *x, long long ldx, int B, int H, int W, int Co,
                      int Ci, float *dw, float *db, float *workspace, void *stream);

/* LayerNorm over channel rows of the fp32 stream (the nn.LayerNorm layers of msft's ViTBlock,
 * msft_arch.py:152-153, applied at :163-165): out = (x - mean) * rstd * weight + bias, written as fp32 or
 * bf16 with row stride ldo >= C (columns C..ldo-1 zeroed), like dsw_rmsnorm_fwd.  C % 4 == 0, C <= 256;
 * mean, rstd: fp32 [M], both or neither.  Backward: dx fp32 [M][C]; partial holds dsw_rmsnorm_blocks(M)
 * rows of [d(weight) | d(bias)] (2 C floats each) to be summed over rows.  dx_add (fp32 [M][C], or NULL) and
 * dx_add2 (fp32, row stride ld_add2, covering the first cols2 columns; or NULL) are gradients that reached x
 * by other ways - the residual around the block (msft_arch.py:163-165), the concatenation in front of the
 * fusing block (msft_arch.py:210) - and are added as dx is written, in place of passes of their own. */
int dsw_layernorm_fwd(const float *x, const float *weight, const float *bias, float eps, long long M, int C,
                      void *out, int out_dtype, long long ldo, float *mean, float *rstd, void *stream);
int dsw_layernorm_bwd(const void *dy, int dy_dtype, long long ld_dy, const float *x, const float *weight,
                      const float *mean, const float *rstd, long long M, int C, float *dx, const float *dx_add,
                      const float *dx_add2, long long ld_add2, int cols2, float *partial, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DSWIN_HIP_H */
