// NHWC convolutions that sit on the DS-MSA path next to the attention kernels:
//   * depthwise 3x3 (out_conv, DSwinT_arch.py:198, :317), forward / data gradient / weight gradient;
//   * the sparse discriminator (DSwinT_arch.py:180-186, :266): conv C->4 3x3 + SiLU + conv 4->4
//     k=w stride w + SiLU + conv 4->4 3x3, forward and backward, without the reference's
//     NHWC->NCHW copy of the whole activation.
// All are bandwidth-bound: every kernel streams the activation once, 4 channels per lane, with the
// small weight sets in LDS.  Parameter gradients use per-block partial sums reduced in a fixed
// order (no atomics, bitwise reproducible).
#include <mutex>

#include "common.hpp"

namespace dsw {

constexpr int CV_THREADS = 256;

__device__ __forceinline__ float silu(float v) { return v / (1.f + __expf(-v)); }
__device__ __forceinline__ float silu_grad(float v) {
    const float s = 1.f / (1.f + __expf(-v));
    return s * (1.f + v * (1.f - s));
}
// four fused multiply-adds as two packed ones (v_pk_fma_f32 does two lanes' worth per issue slot)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    const f32x2_t lo = __builtin_elementwise_fma(f32x2_t{a.x, a.y}, f32x2_t{b.x, b.y}, f32x2_t{c.x, c.y});
    const f32x2_t hi = __builtin_elementwise_fma(f32x2_t{a.z, a.w}, f32x2_t{b.z, b.w}, f32x2_t{c.z, c.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 fma4s(float4 a, float s, float4 c) {
    const f32x2_t ss = {s, s};
    const f32x2_t lo = __builtin_elementwise_fma(f32x2_t{a.x, a.y}, ss, f32x2_t{c.x, c.y});
    const f32x2_t hi = __builtin_elementwise_fma(f32x2_t{a.z, a.w}, ss, f32x2_t{c.z, c.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}

// ---------------------------------------------------------------------------------------------
// depthwise 3x3, stride 1, zero padding 1.  FLIP = data gradient (taps mirrored, no bias).
// weights [C][9] fp32 in global -> LDS as [9][C]
// ---------------------------------------------------------------------------------------------
// An image row is W*C/4 consecutive float4; thread i of a row segment owns float4 i (pixel i / (C/4),
// channel quad i % (C/4)) and walks down CW_ROWS image rows with a 3 x 3 register window, so every
// load instruction of a wave is one contiguous run, a row is fetched (CW_ROWS + 2) / CW_ROWS times,
// and the left / right neighbours come from the vector cache.  The 9 tap vectors stay in registers.
constexpr int CW_ROWS = 8;
// RES (forward, fp32 result): y = resid + *rscale * conv(x), the ReZero residual of the caller's block
// (DSwinT_arch.py:356) taken in the same pass.
template <typename TI, typename TO, bool FLIP, bool RES = false>
__global__ __launch_bounds__(CV_THREADS) void dwconv_stream_kernel(const TI *__restrict__ x, const float *__restrict__ wgt,
                                                                   const float *__restrict__ bias, int H, int W, int C,
                                                                   int segs, TO *__restrict__ y, const int *__restrict__ identity_if_zero,
                                                                   const float *__restrict__ resid = nullptr,
                                                                   const float *__restrict__ rscale = nullptr,
                                                                   const TI *__restrict__ x_if_zero = nullptr) {
    // identity_if_zero: a device counter; when it holds 0 the kernel copies its input instead of convolving
    // (DS-MSA returns its input untouched when no window passed the gate, DSwinT_arch.py:291-292, and this
    // lets a caller that never reads the count on the host keep that behaviour)
    const bool ident = identity_if_zero != nullptr && *identity_if_zero == 0;
    const int CV = C >> 2, rowq = W * CV, hblocks = (H + CW_ROWS - 1) / CW_ROWS;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);  // an XCD takes whole images: the halo rows of vertical neighbours are L2 hits
    // (wave-uniform quotients of the vector unit's division, handed back to scalar registers)
    const int bh = __builtin_amdgcn_readfirstlane(bid / segs), seg = bid - bh * segs;
    const int b = __builtin_amdgcn_readfirstlane(bh / hblocks), hb = bh - b * hblocks;
    const int i = seg * CV_THREADS + threadIdx.x;
    if (i >= rowq) return;
    const int xx = i / CV, c = (i - xx * CV) << 2;
    // The 9 taps of the thread's four channels are 36 consecutive floats of the [C][9] weight (144 bytes, 16-byte
    // aligned): nine 16-byte loads straight into registers and a transposition by register naming - no LDS copy of the
    // weight, no barrier in front of the stream.
    float wr[36];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(wgt + (size_t)c * 9 + 4 * q);
        wr[4 * q] = v.x, wr[4 * q + 1] = v.y, wr[4 * q + 2] = v.z, wr[4 * q + 3] = v.w;
    }
    float4 wq[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tt = FLIP ? 8 - t : t;
        wq[t] = make_float4(wr[tt], wr[9 + tt], wr[18 + tt], wr[27 + tt]);
    }
    const float4 bq = (bias != nullptr && !FLIP) ? *reinterpret_cast<const float4 *>(bias + c) : f4zero();
    const size_t rs = (size_t)W * C;
    // x_if_zero: what the identity copies when the counter holds 0 (the producer of x skipped its pass in that case)
    const TI *src = (ident && x_if_zero != nullptr ? x_if_zero : x) + (size_t)b * H * rs + (size_t)i * 4;
    TO *dst = y + (size_t)b * H * rs + (size_t)i * 4;
    const bool has_l = xx > 0, has_r = xx < W - 1;
    const int h0 = hb * CW_ROWS, h1 = min(h0 + CW_ROWS, H);
    float4 r0[3], r1[3], r2[3];
    auto load_row = [&](int yy, float4(&r)[3]) {
        if (yy < 0 || yy >= H) {
            r[0] = r[1] = r[2] = f4zero();
        } else {
            const TI *pr = src + (size_t)yy * rs;
            r[1] = io<TI>::ld4(pr);
            r[0] = has_l ? io<TI>::ld4(pr - C) : f4zero();
            r[2] = has_r ? io<TI>::ld4(pr + C) : f4zero();
        }
    };
    float rsv = 1.f;
    const float *res = nullptr;
    if constexpr (RES) {
        rsv = *rscale;
        res = resid + (size_t)b * H * rs + (size_t)i * 4;
    }
    load_row(h0 - 1, r0);
    load_row(h0, r1);
#pragma unroll 2
    for (int h = h0; h < h1; ++h) {
        load_row(h + 1, r2);
        float4 rv = f4zero();
        if constexpr (RES) rv = *reinterpret_cast<const float4 *>(res + (size_t)h * rs);
        float4 acc = bq;
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
            acc = fma4(r0[bb], wq[bb], acc);
            acc = fma4(r1[bb], wq[3 + bb], acc);
            acc = fma4(r2[bb], wq[6 + bb], acc);
        }
        float4 v = ident ? r1[1] : acc;
        if constexpr (RES) v = rv + v * rsv;
        io<TO>::st4(dst + (size_t)h * rs, v);
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) r0[bb] = r1[bb], r1[bb] = r2[bb];
    }
}

// weight / bias gradient partials of the depthwise convolution: one block per DW_ROWS image rows, summed over the
// blocks by partial_reduce_kernel.
constexpr int DW_ROWS = 2;
constexpr int WG_THREADS = 448;  // sd_bwd_w0_kernel: 3 parts x (45 quads x 3 tap rows) at C = 180
// Thread = (pixel part, channel quad), all nine taps in registers: walking along the block's two image rows the
// three tap columns are a sliding window of x, so a pixel costs four x loads (the rows above, of and below the pair)
// and two dy loads.  The parts are summed through LDS in part order.  Measured at 32 x 64 x 64 x 180: 36 us with 16
// pixels per thread (192 threads) and the pixel loop not unrolled (about 100 registers); 45 us unrolled by two,
// 53 us with 8 pixels per thread, 41 / 39 us with 128 / 256 threads; a thread per tap row (half the registers,
// twice the loads) 53 us.
constexpr int WR_THREADS = 192;  // 4 parts x 45 quads at C = 180: 16 pixels of the row pair per thread
template <typename TG, typename TX>
__global__ __launch_bounds__(WR_THREADS) void dwconv_wgrad_rows_kernel(const TG *__restrict__ dy, const TX *__restrict__ x,
                                                                       int H, int W, int C, float *__restrict__ partial,
                                                                       const int *__restrict__ identity_if_zero) {
    static_assert(DW_ROWS == 2, "the row pair is written out below");
    if (identity_if_zero != nullptr && *identity_if_zero == 0) return;
    extern __shared__ __attribute__((aligned(16))) float red[];  // [CV][10] float4: nine taps + dy sum, x 4 channels
    const int CV = C >> 2, PP = max(1, min(WR_THREADS / CV, W));
    const int rows_per_img = (H + DW_ROWS - 1) / DW_ROWS;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = __builtin_amdgcn_readfirstlane(bid / rows_per_img), h0 = (bid - b * rows_per_img) * DW_ROWS;
    const int cv = threadIdx.x % CV, part = threadIdx.x / CV, c = cv << 2;
    const int chunk = (W + PP - 1) / PP, p0 = part * chunk, p1 = min(p0 + chunk, W);
    float4 acc[3][3], gsum = f4zero();
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) acc[a][bb] = f4zero();
    if (part < PP && p0 < p1) {
        const TG *gimg = dy + (size_t)b * H * W * C + c;
        const TX *ximg = x + (size_t)b * H * W * C + c;
        const TX *xrow[4];
        bool xok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yy = h0 - 1 + j;
            xok[j] = yy >= 0 && yy < H;
            xrow[j] = ximg + (size_t)(xok[j] ? yy : 0) * W * C;
        }
        const bool g1ok = h0 + 1 < H;
        const TG *grow0 = gimg + (size_t)h0 * W * C, *grow1 = gimg + (size_t)(g1ok ? h0 + 1 : h0) * W * C;
        float4 xm[4], xc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xm[j] = (p0 > 0 && xok[j]) ? io<TX>::ld4(xrow[j] + (size_t)(p0 - 1) * C) : f4zero();
            xc[j] = xok[j] ? io<TX>::ld4(xrow[j] + (size_t)p0 * C) : f4zero();
        }
#pragma unroll 1
        for (int p = p0; p < p1; ++p) {
            float4 xp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xp[j] = (p + 1 < W && xok[j]) ? io<TX>::ld4(xrow[j] + (size_t)(p + 1) * C) : f4zero();
            const float4 g0 = io<TG>::ld4(grow0 + (size_t)p * C);
            const float4 g1 = g1ok ? io<TG>::ld4(grow1 + (size_t)p * C) : f4zero();
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                acc[a][0] = fma4(g0, xm[a], acc[a][0]);
                acc[a][1] = fma4(g0, xc[a], acc[a][1]);
                acc[a][2] = fma4(g0, xp[a], acc[a][2]);
                acc[a][0] = fma4(g1, xm[a + 1], acc[a][0]);
                acc[a][1] = fma4(g1, xc[a + 1], acc[a][1]);
                acc[a][2] = fma4(g1, xp[a + 1], acc[a][2]);
            }
            gsum = gsum + g0 + g1;
#pragma unroll
            for (int j = 0; j < 4; ++j) xm[j] = xc[j], xc[j] = xp[j];
        }
    }
    for (int pt = 0; pt < PP; ++pt) {
        if (part == pt) {
            float4 *dst = reinterpret_cast<float4 *>(red) + (size_t)cv * 10;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int bb = 0; bb < 3; ++bb) dst[a * 3 + bb] = pt == 0 ? acc[a][bb] : dst[a * 3 + bb] + acc[a][bb];
            dst[9] = pt == 0 ? gsum : dst[9] + gsum;
        }
        __syncthreads();
    }
    // partial[blk][c][k]: k = a*3 + bb, k = 9: bias
    for (int i = threadIdx.x; i < C * 10; i += WR_THREADS) {
        const int ch = i / 10, k = i - ch * 10;
        partial[(size_t)bid * C * 10 + i] = red[((size_t)(ch >> 2) * 10 + k) * 4 + (ch & 3)];
    }
}

// out[i] = sum over blocks of partial[blk][i]: PR_GROUPS interleaved block groups (block index mod PR_GROUPS), each
// summed by a thread per column with four running sums, then the groups' sums in a fixed order.  A block takes
// PR_COLS adjacent columns: 8 (32 bytes of every partial row) measured 6.3 us on 1024 x 1800 partials, 4 columns
// 10.2 us, 16 columns 7.4 us.
constexpr int PR_COLS = 8, PR_GROUPS = 32;
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float *__restrict__ partial, int nblocks, int n,
                                                             float *__restrict__ out) {
    __shared__ float part[PR_GROUPS][PR_COLS + 1];
    const int e = threadIdx.x & (PR_COLS - 1), grp = threadIdx.x / PR_COLS;
    const int i = blockIdx.x * PR_COLS + e;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
        int bl = grp;
        for (; bl + 3 * PR_GROUPS < nblocks; bl += 4 * PR_GROUPS) {
            a0 += partial[(size_t)bl * n + i];
            a1 += partial[(size_t)(bl + PR_GROUPS) * n + i];
            a2 += partial[(size_t)(bl + 2 * PR_GROUPS) * n + i];
            a3 += partial[(size_t)(bl + 3 * PR_GROUPS) * n + i];
        }
        for (; bl < nblocks; bl += PR_GROUPS) a0 += partial[(size_t)bl * n + i];
    }
    part[grp][e] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (grp == 0 && i < n) {
        float t = 0.f;
#pragma unroll 8
        for (int k = 0; k < PR_GROUPS; ++k) t += part[k][e];
        out[i] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// column sums of a row-major [rows][cols] matrix (the bias gradients of the path's Linear layers):
// block = CS_ROWS consecutive rows, thread = (row lane, column quad), partial sums per block reduced
// by partial_reduce_kernel in a fixed order.
// ---------------------------------------------------------------------------------------------
constexpr int CS_ROWS = 128, CS_THREADS = 512;
template <typename T>
__global__ __launch_bounds__(CS_THREADS) void colsum_kernel(const T *__restrict__ x, long long rows, int cols,
                                                            float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [RL][cols]
    const int CQ = cols >> 2, RL = CS_THREADS / CQ;  // row lanes
    const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ;
    const long long r0 = (long long)blockIdx.x * CS_ROWS, r1 = min(r0 + CS_ROWS, rows);
    float4 acc = f4zero();
    if (rl < RL) {
        const T *p = x + (size_t)(r0 + rl) * cols + (cq << 2);
        const size_t step = (size_t)RL * cols;
        long long r = r0 + rl;
        for (; r + 3 * RL < r1; r += 4 * RL, p += 4 * step) {  // four loads in flight
            const float4 a = io<T>::ld4(p), b = io<T>::ld4(p + step), c = io<T>::ld4(p + 2 * step), d4 = io<T>::ld4(p + 3 * step);
            acc = acc + ((a + b) + (c + d4));
        }
        for (; r < r1; r += RL, p += step) acc = acc + io<T>::ld4(p);
        *reinterpret_cast<float4 *>(red + (size_t)rl * cols + (cq << 2)) = acc;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += CS_THREADS) {
        float s = 0.f;
        for (int k = 0; k < RL; ++k) s += red[(size_t)k * cols + c];
        partial[(size_t)blockIdx.x * cols + c] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// row copy with dtype conversion between two row strides: dst[r][c] = src[r][c] for c < cols and
// r < src_rows, zero in the other rows and, when zero_pad is set, in the columns cols..dst_ld-1.  Feeds the projection GEMMs with rows padded to
// the widths the GEMM library is fast at (and takes their padded results apart again), in one
// vectorised pass instead of a cast, a fill and a strided copy.
// ---------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_rows_kernel(const TS *__restrict__ src, long long src_ld, TD *__restrict__ dst,
                                                        long long dst_ld, long long rows, long long src_rows, int cols,
                                                        int width) {
    // width = columns written per row (cols, or dst_ld with zero fill), a multiple of 4
    const int WQ = width >> 2;
    const long long items = rows * WQ;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const long long r = it / WQ;
        const int c = (int)(it - r * WQ) << 2;
        const float4 v = (c < cols && r < src_rows) ? io<TS>::ld4(src + r * src_ld + c) : f4zero();
        io<TD>::st4(dst + r * dst_ld + c, v);
    }
}

// ---------------------------------------------------------------------------------------------
// sparse discriminator
//   stage A (per window block): t0 = conv0(x) [saved, pre-SiLU], u = conv2(silu(t0)) [saved, pre-SiLU]
//   stage B (tiny):            sd = conv4(silu(u))
// weights: w0 [4][C][3][3], b0 [4]; w2 [4][4][w][w], b2 [4]; w4 [4][4][3][3], b4 [4]  (fp32)
// ---------------------------------------------------------------------------------------------
// conv0: t0[b,y,x,o] = b0[o] + sum_{ci,taps} w0[o][ci][tap] * x[b,y+a-1,x+bb-1,ci].
// 16 lanes per strip of 4 consecutive pixels of an image row; each lane covers a strided share of
// the channel quads, so one set of weight reads serves 4 pixels and 6 input columns serve 3 taps.
// A strip is 3 tap rows x ceil(C/64) channel rounds; the 6 input vectors of the next round are in
// flight while the current round's 192 FMAs run.  Blocks loop over strips (weights staged once).
// LDS copy of w0 as [tap][ci][o]: one float4 read gives the 4 outputs of (tap, ci).
template <typename TX>
__global__ __launch_bounds__(CV_THREADS) void sd_stage_a_kernel(const TX *__restrict__ x, const float *__restrict__ w0,
                                                                const float *__restrict__ b0, int B, int H, int W, int C,
                                                                float *__restrict__ t0) {
    extern __shared__ __attribute__((aligned(16))) float w0s[];  // [9][C][4]
    for (int i = threadIdx.x; i < 4 * C; i += CV_THREADS) {
        const int o = i / C, ci = i - o * C;
#pragma unroll
        for (int t = 0; t < 9; ++t) w0s[(t * C + ci) * 4 + o] = w0[(size_t)i * 9 + t];
    }
    __syncthreads();
    const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15, CV = C >> 2;
    const int strips_per_row = W >> 2, nit = (CV + 15) >> 4, rounds = 3 * nit;
    const long long nstrips = (long long)B * H * strips_per_row;
    const float4 bq = make_float4(b0[0], b0[1], b0[2], b0[3]);
    for (long long s0 = (long long)blockIdx.x * (CV_THREADS / 16); s0 < nstrips; s0 += (long long)gridDim.x * (CV_THREADS / 16)) {
        // out-of-range groups run on a clamped strip (the shuffles below need every lane) and skip the store
        const bool live = s0 + grp < nstrips;
        const long long sc = live ? s0 + grp : nstrips - 1;
        const int xs = (int)(sc % strips_per_row) << 2;
        const int y = (int)((sc / strips_per_row) % H), b = (int)(sc / ((long long)strips_per_row * H));
        const TX *img = x + (size_t)b * H * W * C;
        auto load_round = [&](int r, float4(&v)[6]) {
            const int a = r / nit, cv = gl + ((r - a * nit) << 4), yy = y + a - 1;
            const bool ok = yy >= 0 && yy < H && cv < CV;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int xc = xs + k - 1;
                v[k] = (ok && xc >= 0 && xc < W) ? io<TX>::ld4(img + ((size_t)yy * W + xc) * C + (cv << 2)) : f4zero();
            }
        };
        float4 acc[4] = {f4zero(), f4zero(), f4zero(), f4zero()};
        float4 vc[6], vn[6];
        load_round(0, vc);
        for (int r = 0; r < rounds; ++r) {
            if (r + 1 < rounds) load_round(r + 1, vn);
            const int a = r / nit, cv = min(gl + ((r - a * nit) << 4), CV - 1);  // past-the-end lanes hold zeros in vc
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) {
                const float4 *wq = reinterpret_cast<const float4 *>(w0s + ((size_t)(a * 3 + bb) * C + (cv << 2)) * 4);
                const float4 q0 = wq[0], q1 = wq[1], q2 = wq[2], q3 = wq[3];
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const float4 vv = vc[pp + bb];
                    acc[pp] = fma4s(q0, vv.x, acc[pp]);
                    acc[pp] = fma4s(q1, vv.y, acc[pp]);
                    acc[pp] = fma4s(q2, vv.z, acc[pp]);
                    acc[pp] = fma4s(q3, vv.w, acc[pp]);
                }
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) vc[k] = vn[k];
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                acc[pp].x += __shfl_xor(acc[pp].x, o, 64);
                acc[pp].y += __shfl_xor(acc[pp].y, o, 64);
                acc[pp].z += __shfl_xor(acc[pp].z, o, 64);
                acc[pp].w += __shfl_xor(acc[pp].w, o, 64);
            }
        }
        if (live && gl < 4) {
            // lane gl stores pixel gl of the strip (every lane holds all four sums)
            const float4 r = gl == 0 ? acc[0] : gl == 1 ? acc[1] : gl == 2 ? acc[2] : acc[3];
            *reinterpret_cast<float4 *>(t0 + (((size_t)b * H + y) * W + xs + gl) * 4) = r + bq;
        }
    }
}

// conv2 (kernel = stride = window): u[b,o,wr,wc] = b2[o] + sum_{c2,pix} w2[o][c2][pix] * silu(t0[b,pix,c2]);
// one block per window, fixed-order block reduction.
__global__ __launch_bounds__(256) void sd_stage_u_kernel(const float *__restrict__ t0, const float *__restrict__ w2,
                                                         const float *__restrict__ b2, int H, int W, int w,
                                                         float *__restrict__ u) {
    __shared__ float red[4];
    const int Hw = H / w, Ww = W / w, L = w * w;
    const int wc = blockIdx.x % Ww, wr = (blockIdx.x / Ww) % Hw, b = blockIdx.x / (Ww * Hw);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int pix = threadIdx.x; pix < L; pix += 256) {
        const int py = pix / w, px = pix - py * w;
        const float4 t = *reinterpret_cast<const float4 *>(t0 + (((size_t)b * H + wr * w + py) * W + wc * w + px) * 4);
        const float sv[4] = {silu(t.x), silu(t.y), silu(t.z), silu(t.w)};
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) acc[o] = fmaf(w2[(size_t)(o * 4 + c2) * L + pix], sv[c2], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const float tsum = block_sum(acc[o], red);
        if (threadIdx.x == 0) u[((size_t)(b * 4 + o) * Hw + wr) * Ww + wc] = tsum + b2[o];
        __syncthreads();
    }
}

// sd[b,o,r,c] = b4[o] + sum_{c2,taps} w4[o][c2][tap] * silu(u[b,c2,r+a-1,c+bb-1]);  thread per output
template <typename TO>
__global__ void sd_stage_b_kernel(const float *__restrict__ u, const float *__restrict__ w4, const float *__restrict__ b4,
                                  int B, int Hw, int Ww, TO *__restrict__ sd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 4 * Hw * Ww) return;
    const int c = i % Ww, r = (i / Ww) % Hw, o = (i / (Ww * Hw)) % 4, b = i / (Ww * Hw * 4);
    // all 36 taps are requested at once (a position outside the map reads the thread's own cell and is not added:
    // with a branch per tap the loads went out one after the other, 36 trips to L2 in a row), then added in tap order
    float uv[36];
    bool in[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) {
        const int c2 = k / 9, a = (k % 9) / 3, bb = k % 3;
        const int rr = r + a - 1, cc = c + bb - 1;
        in[k] = rr >= 0 && rr < Hw && cc >= 0 && cc < Ww;
        uv[k] = u[((size_t)(b * 4 + c2) * Hw + (in[k] ? rr : r)) * Ww + (in[k] ? cc : c)];
    }
    float acc = b4[o];
#pragma unroll
    for (int k = 0; k < 36; ++k)
        if (in[k]) acc = fmaf(w4[o * 36 + k], silu(uv[k]), acc);
    io<TO>::st(sd + i, acc);
}

// ---- backward ------------------------------------------------------------------------------------------
// stage B backward, thread per (b, c2, r, c): du = silu'(u) * sum_{o,taps} w4[o][c2][tap] * dsd[b,o,r-(a-1),c-(bb-1)]
template <typename TG>
__global__ void sd_bwd_b_kernel(const TG *__restrict__ dsd, const float *__restrict__ u, const float *__restrict__ w4, int B,
                                int Hw, int Ww, float *__restrict__ du) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 4 * Hw * Ww) return;
    const int c = i % Ww, r = (i / Ww) % Hw, c2 = (i / (Ww * Hw)) % 4, b = i / (Ww * Hw * 4);
    float gv[36];      // all taps requested at once, as in sd_stage_b_kernel
    bool in[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) {
        const int o = k / 9, a = (k % 9) / 3, bb = k % 3;
        const int rr = r - (a - 1), cc = c - (bb - 1);
        in[k] = rr >= 0 && rr < Hw && cc >= 0 && cc < Ww;
        gv[k] = io<TG>::ld(dsd + ((size_t)(b * 4 + o) * Hw + (in[k] ? rr : r)) * Ww + (in[k] ? cc : c));
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 36; ++k) {
        const int o = k / 9, t = k % 9;
        if (in[k]) acc = fmaf(w4[(o * 4 + c2) * 9 + t], gv[k], acc);
    }
    du[i] = acc * silu_grad(u[i]);
}

// d(w4) [4][4][9] and d(b4) [4]: one block per entry, fixed-order block reduction over all cells
template <typename TG>
__global__ __launch_bounds__(256) void sd_bwd_w4_kernel(const TG *__restrict__ dsd, const float *__restrict__ u, int B, int Hw,
                                                        int Ww, float *__restrict__ dw4, float *__restrict__ db4) {
    __shared__ float red[4];
    const int i = blockIdx.x, ncell = B * Hw * Ww;
    float acc = 0.f;
    if (i < 144) {
        const int o = i / 36, c2 = (i / 9) % 4, a = (i % 9) / 3, bb = i % 3;
        for (int j = threadIdx.x; j < ncell; j += 256) {
            const int c = j % Ww, r = (j / Ww) % Hw, b = j / (Ww * Hw);
            const int rr = r + a - 1, cc = c + bb - 1;
            if (rr < 0 || rr >= Hw || cc < 0 || cc >= Ww) continue;
            acc = fmaf(io<TG>::ld(dsd + ((size_t)(b * 4 + o) * Hw + r) * Ww + c), silu(u[((size_t)(b * 4 + c2) * Hw + rr) * Ww + cc]), acc);
        }
    } else {
        const int o = i - 144;
        for (int j = threadIdx.x; j < ncell; j += 256) {
            const int b = j / (Ww * Hw), cell = j - b * Ww * Hw;
            acc += io<TG>::ld(dsd + (size_t)(b * 4 + o) * Hw * Ww + cell);
        }
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) {
        if (i < 144) dw4[i] = t;
        else db4[i - 144] = t;
    }
}

// stage A backward, window side: dt0[b,y,x,c2] = silu'(t0) * sum_o w2[o][c2][pix] * du[b,o,win];  thread per pixel
__global__ void sd_bwd_a_pix_kernel(const float *__restrict__ du, const float *__restrict__ t0, const float *__restrict__ w2,
                                    int B, int H, int W, int w, float *__restrict__ dt0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H * W) return;
    const int xx = i % W, y = (i / W) % H, b = i / (W * H);
    const int Hw = H / w, Ww = W / w, L = w * w;
    const int wr = y / w, wc = xx / w, pix = (y - wr * w) * w + (xx - wc * w);
    float d[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) d[o] = du[((size_t)(b * 4 + o) * Hw + wr) * Ww + wc];
    const float4 t = *reinterpret_cast<const float4 *>(t0 + (size_t)i * 4);
    const float tv[4] = {t.x, t.y, t.z, t.w};
    float out[4];
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
        float acc = 0.f;
#pragma unroll
        for (int o = 0; o < 4; ++o) acc = fmaf(w2[(size_t)(o * 4 + c2) * L + pix], d[o], acc);
        out[c2] = acc * silu_grad(tv[c2]);
    }
    *reinterpret_cast<float4 *>(dt0 + (size_t)i * 4) = make_float4(out[0], out[1], out[2], out[3]);
}

// d(w2) [4][4][L] and d(b2) [4].  Block = SW2_ENT (c2, pix) entries x SW2_GRP window groups; every thread sums
// its windows for the 4 outputs at once, then the group partials are added in order.  (4 entries x 64 groups: 257
// blocks at w = 16 whose threads walk 8 windows each at 512 windows; 16 x 16 - 65 blocks, 32 windows per thread -
// took 15.9 us for 2 MB of operands.)
constexpr int SW2_ENT = 4, SW2_GRP = 256 / SW2_ENT;
__global__ __launch_bounds__(256) void sd_bwd_w2_kernel(const float *__restrict__ du, const float *__restrict__ t0, int B, int H,
                                                        int W, int w, float *__restrict__ dw2, float *__restrict__ db2) {
    __shared__ float part[SW2_GRP][SW2_ENT][5];
    const int L = w * w, Hw = H / w, Ww = W / w, nwin = B * Hw * Ww;
    const int e = threadIdx.x % SW2_ENT, grp = threadIdx.x / SW2_ENT;
    const int ent = blockIdx.x * SW2_ENT + e;  // c2 * L + pix; entries >= 4L: the four d(b2) sums
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (ent < 4 * L) {
        const int c2 = ent / L, pix = ent - c2 * L, py = pix / w, px = pix - py * w;
        for (int wi = grp; wi < nwin; wi += SW2_GRP) {
            const int wc = wi % Ww, wr = (wi / Ww) % Hw, b = wi / (Ww * Hw);
            const float tv = silu(t0[(((size_t)b * H + wr * w + py) * W + wc * w + px) * 4 + c2]);
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o] = fmaf(du[((size_t)(b * 4 + o) * Hw + wr) * Ww + wc], tv, acc[o]);
        }
    } else if (ent == 4 * L) {
        for (int wi = grp; wi < nwin; wi += SW2_GRP) {
            const int cell = wi % (Ww * Hw), b = wi / (Ww * Hw);
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o] += du[(size_t)(b * 4 + o) * Hw * Ww + cell];
        }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) part[grp][e][o] = acc[o];
    __syncthreads();
    if (grp == 0 && ent <= 4 * L) {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float t = 0.f;
            for (int k = 0; k < SW2_GRP; ++k) t += part[k][e][o];
            if (ent < 4 * L) dw2[(size_t)o * 4 * L + ent] = t;
            else db2[o] = t;
        }
    }
}

// dx[b,y,x,ci] = sum_{o,taps} w0[o][ci][tap] * dt0[b, y-(a-1), x-(bb-1), o].  Same row-segment / walk-down
// mapping as the depthwise kernel; the channel quad's 36 weight vectors live in registers and the
// dt0 window (16 B per pixel, shared by all channel quads of the pixel) comes from the vector cache.
// (staging the 36 weight vectors costs a block about as much as 16 rows of work: long walks)
constexpr int DX_ROWS = 32;
template <typename TO, bool ACC>
__global__ __launch_bounds__(CV_THREADS) void sd_bwd_dx_kernel(const float *__restrict__ dt0, const float *__restrict__ w0,
                                                               int H, int W, int C, int segs, TO *__restrict__ dx) {
    extern __shared__ __attribute__((aligned(16))) float w0s[];  // [tap][o][C]
    for (int i = threadIdx.x; i < 4 * C; i += CV_THREADS) {
        const int o = i / C, ci = i - o * C;
#pragma unroll
        for (int t = 0; t < 9; ++t) w0s[(t * 4 + o) * C + ci] = w0[(size_t)i * 9 + t];
    }
    __syncthreads();
    const int CV = C >> 2, rowq = W * CV, hblocks = (H + DX_ROWS - 1) / DX_ROWS;
    const int seg = blockIdx.x % segs, hb = (blockIdx.x / segs) % hblocks, b = blockIdx.x / (segs * hblocks);
    const int i = seg * CV_THREADS + threadIdx.x;
    if (i >= rowq) return;
    const int xx = i / CV, c = (i - xx * CV) << 2;
    float4 wq[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int o = 0; o < 4; ++o) wq[t][o] = *reinterpret_cast<const float4 *>(w0s + (size_t)(t * 4 + o) * C + c);
    const float *gimg = dt0 + (size_t)b * H * W * 4 + (size_t)xx * 4;
    TO *dst = dx + (size_t)b * H * W * C + (size_t)i * 4;
    const size_t rs = (size_t)W * C;
    const bool has_l = xx > 0, has_r = xx < W - 1;
    const int h0 = hb * DX_ROWS, h1 = min(h0 + DX_ROWS, H);
    // g[k]: dt0 at columns xx-1, xx, xx+1; rows y-1 (r0), y (r1), y+1 (r2).  Tap (a, bb) pairs with row y+1-a, column xx+1-bb.
    float4 r0[3], r1[3], r2[3];
    auto load_row = [&](int yy, float4(&r)[3]) {
        if (yy < 0 || yy >= H) {
            r[0] = r[1] = r[2] = f4zero();
        } else {
            const float *pr = gimg + (size_t)yy * W * 4;
            r[1] = *reinterpret_cast<const float4 *>(pr);
            r[0] = has_l ? *reinterpret_cast<const float4 *>(pr - 4) : f4zero();
            r[2] = has_r ? *reinterpret_cast<const float4 *>(pr + 4) : f4zero();
        }
    };
    load_row(h0 - 1, r0);
    load_row(h0, r1);
    for (int h = h0; h < h1; ++h) {
        load_row(h + 1, r2);
        float4 acc = f4zero();
        if constexpr (ACC) acc = io<TO>::ld4(dst + (size_t)h * rs);  // add into an existing dx
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
            const float4 g2 = r2[2 - bb], g1 = r1[2 - bb], g0 = r0[2 - bb];  // a = 0, 1, 2
            acc = fma4s(wq[bb][0], g2.x, acc), acc = fma4s(wq[bb][1], g2.y, acc);
            acc = fma4s(wq[bb][2], g2.z, acc), acc = fma4s(wq[bb][3], g2.w, acc);
            acc = fma4s(wq[3 + bb][0], g1.x, acc), acc = fma4s(wq[3 + bb][1], g1.y, acc);
            acc = fma4s(wq[3 + bb][2], g1.z, acc), acc = fma4s(wq[3 + bb][3], g1.w, acc);
            acc = fma4s(wq[6 + bb][0], g0.x, acc), acc = fma4s(wq[6 + bb][1], g0.y, acc);
            acc = fma4s(wq[6 + bb][2], g0.z, acc), acc = fma4s(wq[6 + bb][3], g0.w, acc);
        }
        io<TO>::st4(dst + (size_t)h * rs, acc);
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) r0[bb] = r1[bb], r1[bb] = r2[bb];
    }
}

// d(w0) / d(b0) partials: block = SD_ROWS image rows; thread = (pixel part, channel quad, tap row a).
// Walking along an image row the three tap columns are a sliding window of x, so each pixel costs one
// new x load and one dt0 load for 48 FMAs.  12 float4 accumulators: [bb][o] x 4 channels, plus the
// dt0 sum for d(b0).  The parts are summed through LDS in part order.
constexpr int SD_ROWS = 4;
template <typename TX>
__global__ __launch_bounds__(WG_THREADS) void sd_bwd_w0_kernel(const float *__restrict__ dt0, const TX *__restrict__ x, int H,
                                                               int W, int C, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [CV*3][13][4]
    const int CV = C >> 2, NI = CV * 3, PP = max(1, WG_THREADS / NI), rows_per_img = (H + SD_ROWS - 1) / SD_ROWS;
    const int b = blockIdx.x / rows_per_img, h0 = (blockIdx.x - b * rows_per_img) * SD_ROWS;
    const float *gimg = dt0 + (size_t)b * H * W * 4;
    float *dst = partial + (size_t)blockIdx.x * (4 * C * 9 + 4);
    const int item = threadIdx.x % NI, part = threadIdx.x / NI;
    const int cv = item / 3, a = item - cv * 3, c = cv << 2;
    const int chunk = (W + PP - 1) / PP, p0 = part * chunk, p1 = min(p0 + chunk, W);
    float4 acc[3][4], gsum = f4zero();
#pragma unroll
    for (int bb = 0; bb < 3; ++bb)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[bb][o] = f4zero();
    if (part < PP && p0 < p1) {
        const TX *ximg = x + (size_t)b * H * W * C + c;
        for (int h = h0; h < min(h0 + SD_ROWS, H); ++h) {
            const int yy = h + a - 1;
            if (yy < 0 || yy >= H) continue;
            const TX *xrow = ximg + (size_t)yy * W * C;
            const float *grow = gimg + (size_t)h * W * 4;
            float4 xm = p0 > 0 ? io<TX>::ld4(xrow + (size_t)(p0 - 1) * C) : f4zero();
            float4 xc = io<TX>::ld4(xrow + (size_t)p0 * C);
#pragma unroll 4
            for (int p = p0; p < p1; ++p) {
                const float4 xp = (p + 1 < W) ? io<TX>::ld4(xrow + (size_t)(p + 1) * C) : f4zero();
                const float4 g = *reinterpret_cast<const float4 *>(grow + (size_t)p * 4);
                const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    acc[0][o] = fma4s(xm, gv[o], acc[0][o]);
                    acc[1][o] = fma4s(xc, gv[o], acc[1][o]);
                    acc[2][o] = fma4s(xp, gv[o], acc[2][o]);
                }
                gsum = gsum + g;
                xm = xc;
                xc = xp;
            }
        }
    }
    for (int pt = 0; pt < PP; ++pt) {
        if (part == pt) {
            float4 *rd = reinterpret_cast<float4 *>(red + (size_t)item * 52);
#pragma unroll
            for (int bb = 0; bb < 3; ++bb)
#pragma unroll
                for (int o = 0; o < 4; ++o) rd[bb * 4 + o] = pt == 0 ? acc[bb][o] : rd[bb * 4 + o] + acc[bb][o];
            rd[12] = pt == 0 ? gsum : rd[12] + gsum;
        }
        __syncthreads();
    }
    // dst[o][ci][tap]
    for (int i = threadIdx.x; i < 4 * C * 9; i += WG_THREADS) {
        const int o = i / (C * 9), r = i - o * C * 9, ci = r / 9, t = r - ci * 9, a2 = t / 3, bb = t - a2 * 3;
        dst[i] = red[((size_t)((ci >> 2) * 3 + a2) * 13 + bb * 4 + o) * 4 + (ci & 3)];
    }
    // d(b0): the dt0 sum of the (channel quad 0, a = 1) thread, whose rows are all valid
    if (threadIdx.x < 4) dst[4 * C * 9 + threadIdx.x] = red[(size_t)(0 * 3 + 1) * 52 + 48 + threadIdx.x];
}

// packed [C][10] -> d_weight [C][9], d_bias [C]
// zero_if_zero: a device counter; when it holds 0 the gradients are zero (the forward was the identity)
__global__ void split10_kernel(const float *__restrict__ packed, int C, float *__restrict__ dw, float *__restrict__ db,
                               const int *__restrict__ zero_if_zero) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * 10) return;
    const int c = i / 10, k = i - c * 10;
    const float v = (zero_if_zero != nullptr && *zero_if_zero == 0) ? 0.f : packed[i];
    if (k < 9) dw[c * 9 + k] = v;
    else db[c] = v;
}
// packed [n0 + n1] -> a [n0], b [n1]
__global__ void copy2_kernel(const float *__restrict__ packed, int n0, int n1, float *__restrict__ a, float *__restrict__ b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n0) a[i] = packed[i];
    else if (i < n0 + n1) b[i - n0] = packed[i];
}

// the depthwise kernels read the [C][9] weight and the bias 16 bytes at a time
static int check_conv_params(const char *who, const float *weight, const float *bias) {
    if (((uintptr_t)weight & 15) || (bias != nullptr && ((uintptr_t)bias & 15)))
        return fail(DSW_EUNSUPPORTED, "%s: weight and bias must be 16-byte aligned", who);
    return DSW_OK;
}

static int check_conv(const char *who, int B, int H, int W, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return fail(DSW_EINVAL, "%s: bad sizes B=%d H=%d W=%d C=%d", who, B, H, W, C);
    if (C % 4) return fail(DSW_EUNSUPPORTED, "%s: channel count %d must be a multiple of 4", who, C);
    if (C > 1024) return fail(DSW_EUNSUPPORTED, "%s: channel count %d > 1024", who, C);
    return DSW_OK;
}

// grid of the walk-down kernels: row segments of CV_THREADS float4 x groups of CW_ROWS image rows
static int walk_segs(int W, int C) { return (W * (C / 4) + CV_THREADS - 1) / CV_THREADS; }
static unsigned walk_blocks(int B, int H, int segs) { return (unsigned)B * ((H + CW_ROWS - 1) / CW_ROWS) * segs; }

template <typename K> static int set_lds_attr(K kern, size_t lds, const char *who) {
    if (lds > 160 * 1024) return fail(DSW_EUNSUPPORTED, "%s: needs %zu B of LDS", who, lds);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DSW_ELAUNCH, "%s: LDS attribute: %s", who, hipGetErrorString(e));
    }
    return DSW_OK;
}

// ---- first conv of the sparse discriminator as a GEMM (bf16 runs) -------------------------------
// conv C -> 4, 3x3 = for every pixel the 36 products P[px][4 t + o] = sum_c x[px][c] w0[o][c][t] (one
// [pixels, C] x [C, 36] GEMM on the matrix cores, dsw_linear) followed by a 9-point gather of them:
//   t0[px][o] = b0[o] + sum_t P[px + off(t)][4 t + o],   off(t) = (t / 3 - 1, t % 3 - 1), zero outside.
__global__ __launch_bounds__(256) void sd_stencil_fwd_kernel(const float *__restrict__ P, int ldp, const float *__restrict__ b0, int B,
                                                              int H, int W, float *__restrict__ t0) {
    const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
    if (px >= (long long)B * H * W) return;
    const int xw = (int)(px % W), yh = (int)((px / W) % H);
    float4 acc = *reinterpret_cast<const float4 *>(b0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        if (yh + dy >= 0 && yh + dy < H && xw + dx >= 0 && xw + dx < W)
            acc = acc + *reinterpret_cast<const float4 *>(P + (size_t)(px + dy * W + dx) * ldp + 4 * t);
    }
    *reinterpret_cast<float4 *>(t0 + (size_t)px * 4) = acc;
}

// The transposed gather for the backward pass: G[px][4 t + o] = dt0[px - off(t)][o] (zero outside the
// image), bf16 rows of 64 with columns 36.. zero.  Then dx = G . Wf and dWf = G^T . x are plain GEMMs
// (Wf[4 t + o][c] = w0[o][c][t]), and column 16 + o of G summed over pixels is db0[o].
__global__ __launch_bounds__(256) void sd_gather_bwd_kernel(const float *__restrict__ dt0, int B, int H, int W, bf16_t *__restrict__ G) {
    // thread = (pixel, 16-byte piece of its 128-byte row): piece i < 4 holds taps 2 i and 2 i + 1, piece 4 tap 8 and
    // zeros, pieces 5..7 zeros - eight neighbouring lanes write one whole row (a thread per pixel writing its 128
    // bytes in eight stores touched 64 lines per store instruction: 11.5 us for 17 MB)
    const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long px = gi >> 3;
    const int piece = (int)(gi & 7);
    if (px >= (long long)B * H * W) return;
    const int xw = (int)(px % W), yh = (int)((px / W) % H);
    uint32_t out[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int t = 2 * piece + k;
        if (t < 9) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            float4 v = f4zero();
            if (yh - dy >= 0 && yh - dy < H && xw - dx >= 0 && xw - dx < W) v = *reinterpret_cast<const float4 *>(dt0 + (size_t)(px - dy * W - dx) * 4);
            out[2 * k] = pack_bf16(v.x, v.y);
            out[2 * k + 1] = pack_bf16(v.z, v.w);
        }
    }
    *reinterpret_cast<uint4 *>(G + (size_t)px * 64 + piece * 8) = make_uint4(out[0], out[1], out[2], out[3]);
}

}  // namespace dsw

using namespace dsw;

static int dwconv3x3_fwd_impl(const void *x, int x_dtype, const float *weight, const float *bias, int B, int H, int W, int C, void *y,
                              int y_dtype, const int *identity_if_zero, const void *x_if_zero, void *stream) {
    int rc = check_conv("dsw_dwconv3x3_fwd", B, H, W, C);
    if (rc) return rc;
    DSW_REQUIRE(x && weight && y, "dsw_dwconv3x3_fwd: null pointer");
    if (x_if_zero != nullptr && (identity_if_zero == nullptr || ((uintptr_t)x_if_zero & 15)))
        return fail(DSW_EINVAL, "dsw_dwconv3x3_fwd_src: x_if_zero needs the counter and a 16-byte aligned base");
    if ((rc = check_conv_params("dsw_dwconv3x3_fwd", weight, bias))) return rc;
    const int segs = walk_segs(W, C);
    DSW_DISPATCH_DTYPE(x_dtype, TI, DSW_DISPATCH_DTYPE(y_dtype, TO, {
        hipLaunchKernelGGL((dwconv_stream_kernel<TI, TO, false>), dim3(walk_blocks(B, H, segs)), dim3(CV_THREADS), 0,
                           as_stream(stream), (const TI *)x, weight, bias, H, W, C, segs, (TO *)y, identity_if_zero,
                           (const float *)nullptr, (const float *)nullptr, (const TI *)x_if_zero);
    }));
    return check_launch("dwconv_stream_kernel");
}

static int dwconv3x3_residual_fwd_impl(const void *x, int x_dtype, const float *weight, const float *bias, int B, int H, int W, int C,
                                       const float *resid, const float *rscale, float *y, const int *identity_if_zero,
                                       const void *x_if_zero, void *stream) {
    int rc = check_conv("dsw_dwconv3x3_residual_fwd", B, H, W, C);
    if (rc) return rc;
    DSW_REQUIRE(x && weight && y && resid && rscale, "dsw_dwconv3x3_residual_fwd: null pointer");
    if (((uintptr_t)resid | (uintptr_t)y) & 15) return fail(DSW_EUNSUPPORTED, "dsw_dwconv3x3_residual_fwd: resid and y must be 16-byte aligned");
    if (x_if_zero != nullptr && (identity_if_zero == nullptr || ((uintptr_t)x_if_zero & 15)))
        return fail(DSW_EINVAL, "dsw_dwconv3x3_residual_fwd_src: x_if_zero needs the counter and a 16-byte aligned base");
    if ((rc = check_conv_params("dsw_dwconv3x3_residual_fwd", weight, bias))) return rc;
    const int segs = walk_segs(W, C);
    DSW_DISPATCH_DTYPE(x_dtype, TI, {
        hipLaunchKernelGGL((dwconv_stream_kernel<TI, float, false, true>), dim3(walk_blocks(B, H, segs)), dim3(CV_THREADS), 0,
                           as_stream(stream), (const TI *)x, weight, bias, H, W, C, segs, y, identity_if_zero, resid, rscale,
                           (const TI *)x_if_zero);
    });
    return check_launch("dwconv_stream_kernel<residual>");
}

extern "C" int dsw_dwconv3x3_fwd(const void *x, int x_dtype, const float *weight, const float *bias, int B, int H, int W,
                                 int C, void *y, int y_dtype, const int *identity_if_zero, void *stream) {
    return dwconv3x3_fwd_impl(x, x_dtype, weight, bias, B, H, W, C, y, y_dtype, identity_if_zero, nullptr, stream);
}

extern "C" int dsw_dwconv3x3_fwd_src(const void *x, int x_dtype, const float *weight, const float *bias, int B, int H, int W, int C,
                                     void *y, int y_dtype, const int *identity_if_zero, const void *x_if_zero, void *stream) {
    return dwconv3x3_fwd_impl(x, x_dtype, weight, bias, B, H, W, C, y, y_dtype, identity_if_zero, x_if_zero, stream);
}

extern "C" int dsw_dwconv3x3_residual_fwd(const void *x, int x_dtype, const float *weight, const float *bias, int B, int H, int W,
                                          int C, const float *resid, const float *rscale, float *y, const int *identity_if_zero,
                                          void *stream) {
    return dwconv3x3_residual_fwd_impl(x, x_dtype, weight, bias, B, H, W, C, resid, rscale, y, identity_if_zero, nullptr, stream);
}

extern "C" int dsw_dwconv3x3_residual_fwd_src(const void *x, int x_dtype, const float *weight, const float *bias, int B, int H, int W,
                                              int C, const float *resid, const float *rscale, float *y, const int *identity_if_zero,
                                              const void *x_if_zero, void *stream) {
    return dwconv3x3_residual_fwd_impl(x, x_dtype, weight, bias, B, H, W, C, resid, rscale, y, identity_if_zero, x_if_zero, stream);
}

extern "C" int dsw_dwconv3x3_bwd_blocks(int B, int H, int W) {
    (void)W;
    return B > 0 && H > 0 ? B * ((H + DW_ROWS - 1) / DW_ROWS) : 0;
}

extern "C" int dsw_dwconv3x3_bwd(const void *dy, int dy_dtype, const void *x, int x_dtype, const float *weight, int B,
                                 int H, int W, int C, void *dx, int dx_dtype, float *d_weight, float *d_bias,
                                 float *partial, const int *identity_if_zero, void *stream) {
    int rc = check_conv("dsw_dwconv3x3_bwd", B, H, W, C);
    if (rc) return rc;
    DSW_REQUIRE(dy && x && weight && dx && d_weight && d_bias && partial, "dsw_dwconv3x3_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    if ((rc = check_conv_params("dsw_dwconv3x3_bwd", weight, nullptr))) return rc;
    const int segs = walk_segs(W, C);
    DSW_DISPATCH_DTYPE(dy_dtype, TG, DSW_DISPATCH_DTYPE(dx_dtype, TO, {
        hipLaunchKernelGGL((dwconv_stream_kernel<TG, TO, true>), dim3(walk_blocks(B, H, segs)), dim3(CV_THREADS), 0, st,
                           (const TG *)dy, weight, (const float *)nullptr, H, W, C, segs, (TO *)dx, identity_if_zero);
    }));
    rc = check_launch("dwconv_stream_kernel<flip>");
    if (rc) return rc;
    const int CVq = C / 4;
    if (CVq > WR_THREADS) return fail(DSW_EUNSUPPORTED, "dsw_dwconv3x3_bwd: %d channels > %d", C, WR_THREADS * 4);
    const int nblocks = dsw_dwconv3x3_bwd_blocks(B, H, W);
    const size_t lds2 = (size_t)CVq * 10 * 4 * sizeof(float);
    DSW_DISPATCH_DTYPE(dy_dtype, TG, DSW_DISPATCH_DTYPE(x_dtype, TX, {
        auto kern = dwconv_wgrad_rows_kernel<TG, TX>;
        rc = set_lds_attr(kern, lds2, "dsw_dwconv3x3_bwd");
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(WR_THREADS), lds2, st, (const TG *)dy, (const TX *)x, H, W, C, partial, identity_if_zero);
    }));
    rc = check_launch("dwconv_wgrad_rows_kernel");
    if (rc) return rc;
    // partial rows are [C][10] = 9 taps + bias; reduce into a packed [C*10] staging area at the
    // end of `partial`, then split (the split is a strided copy done by the caller's view)
    float *packed = partial + (size_t)nblocks * C * 10;
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((C * 10 + PR_COLS - 1) / PR_COLS), dim3(256), 0, st, partial, nblocks, C * 10, packed);
    rc = check_launch("partial_reduce_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(split10_kernel, dim3((C * 10 + 255) / 256), dim3(256), 0, st, packed, C, d_weight, d_bias, identity_if_zero);
    return check_launch("split10_kernel");
}

extern "C" int dsw_sd_stencil_fwd(const float *P, int ldp, const float *b0, int B, int H, int W, float *t0, void *stream) {
    if (B <= 0 || H <= 0 || W <= 0 || ldp < 36 || ldp % 4) return fail(DSW_EINVAL, "dsw_sd_stencil_fwd: bad sizes B=%d H=%d W=%d ldp=%d", B, H, W, ldp);
    DSW_REQUIRE(P && b0 && t0, "dsw_sd_stencil_fwd: null pointer");
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(sd_stencil_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), P, ldp, b0, B, H, W, t0);
    return check_launch("sd_stencil_fwd_kernel");
}

extern "C" int dsw_sd_gather_bwd(const float *dt0, int B, int H, int W, void *G, void *stream) {
    if (B <= 0 || H <= 0 || W <= 0) return fail(DSW_EINVAL, "dsw_sd_gather_bwd: bad sizes B=%d H=%d W=%d", B, H, W);
    DSW_REQUIRE(dt0 && G, "dsw_sd_gather_bwd: null pointer");
    const long long n = (long long)B * H * W;
    if (n > 0x0fffffffLL) return fail(DSW_EUNSUPPORTED, "dsw_sd_gather_bwd: %lld pixels", n);
    hipLaunchKernelGGL(sd_gather_bwd_kernel, dim3((unsigned)((n * 8 + 255) / 256)), dim3(256), 0, as_stream(stream), dt0, B, H, W, (bf16_t *)G);
    return check_launch("sd_gather_bwd_kernel");
}

extern "C" int dsw_sparse_disc_tail_fwd(const float *t0, const float *w2, const float *b2, const float *w4, const float *b4, int B,
                                        int H, int W, int w, float *u, void *sd, int sd_dtype, void *stream) {
    if (B <= 0 || H <= 0 || W <= 0) return fail(DSW_EINVAL, "dsw_sparse_disc_tail_fwd: bad sizes B=%d H=%d W=%d", B, H, W);
    DSW_REQUIRE(w > 0 && H % w == 0 && W % w == 0, "dsw_sparse_disc_tail_fwd: H,W (%d,%d) must be multiples of %d", H, W, w);
    if (w % 4) return fail(DSW_EUNSUPPORTED, "dsw_sparse_disc_tail_fwd: window size %d must be a multiple of 4", w);
    DSW_REQUIRE(t0 && w2 && b2 && w4 && b4 && u && sd, "dsw_sparse_disc_tail_fwd: null pointer");
    hipStream_t st = as_stream(stream);
    const int Hw = H / w, Ww = W / w;
    hipLaunchKernelGGL(sd_stage_u_kernel, dim3(B * Hw * Ww), dim3(256), 0, st, t0, w2, b2, H, W, w, u);
    int rc = check_launch("sd_stage_u_kernel");
    if (rc) return rc;
    const int n = B * 4 * Hw * Ww;
    DSW_DISPATCH_DTYPE(sd_dtype, TO, {
        hipLaunchKernelGGL((sd_stage_b_kernel<TO>), dim3((n + 255) / 256), dim3(256), 0, st, u, w4, b4, B, Hw, Ww, (TO *)sd);
    });
    return check_launch("sd_stage_b_kernel");
}

extern "C" int dsw_sparse_disc_fwd(const void *x, int x_dtype, const float *w0, const float *b0, const float *w2,
                                   const float *b2, const float *w4, const float *b4, int B, int H, int W, int C, int w,
                                   float *t0, float *u, void *sd, int sd_dtype, void *stream) {
    int rc = check_conv("dsw_sparse_disc_fwd", B, H, W, C);
    if (rc) return rc;
    DSW_REQUIRE(w > 0 && H % w == 0 && W % w == 0, "dsw_sparse_disc_fwd: H,W (%d,%d) must be multiples of %d", H, W, w);
    if (w % 4) return fail(DSW_EUNSUPPORTED, "dsw_sparse_disc_fwd: window size %d must be a multiple of 4", w);
    DSW_REQUIRE(x && w0 && b0 && w2 && b2 && w4 && b4 && t0 && u && sd, "dsw_sparse_disc_fwd: null pointer");
    hipStream_t st = as_stream(stream);
    const size_t lds = (size_t)(9 * C * 4) * sizeof(float);
    const long long nstrips = (long long)B * H * (W / 4);
    DSW_DISPATCH_DTYPE(x_dtype, TX, {
        auto kern = sd_stage_a_kernel<TX>;
        rc = set_lds_attr(kern, lds, "dsw_sparse_disc_fwd");
        if (rc) return rc;
        const long long want = (nstrips + CV_THREADS / 16 - 1) / (CV_THREADS / 16);
        hipLaunchKernelGGL(kern, dim3((unsigned)(want < 1024 ? want : 1024)), dim3(CV_THREADS), lds, st,
                           (const TX *)x, w0, b0, B, H, W, C, t0);
    });
    rc = check_launch("sd_stage_a_kernel");
    if (rc) return rc;
    return dsw_sparse_disc_tail_fwd(t0, w2, b2, w4, b4, B, H, W, w, u, sd, sd_dtype, stream);
}

extern "C" int dsw_sparse_disc_tail_bwd(const void *dsd, int dsd_dtype, const float *w2, const float *w4, const float *t0,
                                        const float *u, int B, int H, int W, int w, float *du, float *dt0, float *dw2, float *db2,
                                        float *dw4, float *db4, void *stream) {
    if (B <= 0 || H <= 0 || W <= 0) return fail(DSW_EINVAL, "dsw_sparse_disc_tail_bwd: bad sizes B=%d H=%d W=%d", B, H, W);
    DSW_REQUIRE(w > 0 && H % w == 0 && W % w == 0, "dsw_sparse_disc_tail_bwd: H,W (%d,%d) must be multiples of %d", H, W, w);
    DSW_REQUIRE(dsd && w2 && w4 && t0 && u && du && dt0 && dw2 && db2 && dw4 && db4, "dsw_sparse_disc_tail_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    const int Hw = H / w, Ww = W / w, L = w * w, n = B * 4 * Hw * Ww;
    DSW_DISPATCH_DTYPE(dsd_dtype, TG, {
        hipLaunchKernelGGL((sd_bwd_b_kernel<TG>), dim3((n + 255) / 256), dim3(256), 0, st, (const TG *)dsd, u, w4, B, Hw, Ww, du);
        hipLaunchKernelGGL((sd_bwd_w4_kernel<TG>), dim3(148), dim3(256), 0, st, (const TG *)dsd, u, B, Hw, Ww, dw4, db4);
    });
    int rc = check_launch("sd_bwd_b/w4 kernels");
    if (rc) return rc;
    hipLaunchKernelGGL(sd_bwd_a_pix_kernel, dim3((B * H * W + 255) / 256), dim3(256), 0, st, du, t0, w2, B, H, W, w, dt0);
    hipLaunchKernelGGL(sd_bwd_w2_kernel, dim3((4 * L + 1 + SW2_ENT - 1) / SW2_ENT), dim3(256), 0, st, du, t0, B, H, W, w, dw2, db2);
    return check_launch("sd_bwd_a/w2 kernels");
}

extern "C" int dsw_sparse_disc_bwd_blocks(int B, int H) { return B > 0 && H > 0 ? B * ((H + SD_ROWS - 1) / SD_ROWS) : 0; }

extern "C" int dsw_sparse_disc_bwd(const void *dsd, int dsd_dtype, const void *x, int x_dtype, const float *w0,
                                   const float *w2, const float *w4, const float *t0, const float *u, int B, int H, int W,
                                   int C, int w, void *dx, int dx_dtype, int accumulate_dx, float *dw0, float *db0,
                                   float *dw2, float *db2, float *dw4, float *db4, float *scratch, void *stream) {
    int rc = check_conv("dsw_sparse_disc_bwd", B, H, W, C);
    if (rc) return rc;
    DSW_REQUIRE(w > 0 && H % w == 0 && W % w == 0, "dsw_sparse_disc_bwd: H,W (%d,%d) must be multiples of %d", H, W, w);
    DSW_REQUIRE(dsd && x && w0 && w2 && w4 && t0 && u && dx && dw0 && db0 && dw2 && db2 && dw4 && db4 && scratch,
                "dsw_sparse_disc_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    const int Hw = H / w, Ww = W / w, n = B * 4 * Hw * Ww;
    // scratch layout: du [n] | dt0 [B*H*W*4] | w0 partials [nblk][4*C*9+4] | packed [4*C*9+4]
    float *du = scratch;
    float *dt0 = du + ((n + 3) & ~3);
    float *part = dt0 + (size_t)B * H * W * 4;
    const int nblk = dsw_sparse_disc_bwd_blocks(B, H);
    const int np = 4 * C * 9 + 4;
    float *packed = part + (size_t)nblk * np;
    rc = dsw_sparse_disc_tail_bwd(dsd, dsd_dtype, w2, w4, t0, u, B, H, W, w, du, dt0, dw2, db2, dw4, db4, stream);
    if (rc) return rc;
    const size_t lds = (size_t)36 * C * sizeof(float);
    DSW_DISPATCH_DTYPE(dx_dtype, TO, {
        auto kern = accumulate_dx ? sd_bwd_dx_kernel<TO, true> : sd_bwd_dx_kernel<TO, false>;
        rc = set_lds_attr(kern, lds, "dsw_sparse_disc_bwd");
        if (rc) return rc;
        const int segs = walk_segs(W, C);
        hipLaunchKernelGGL(kern, dim3((unsigned)B * ((H + DX_ROWS - 1) / DX_ROWS) * segs), dim3(CV_THREADS), lds, st, dt0, w0, H, W, C,
                           segs, (TO *)dx);
    });
    DSW_DISPATCH_DTYPE(x_dtype, TX, {
        auto kern0 = sd_bwd_w0_kernel<TX>;
        const int NI = C / 4 * 3;
        if (NI > WG_THREADS) return fail(DSW_EUNSUPPORTED, "dsw_sparse_disc_bwd: %d channels > %d", C, WG_THREADS / 3 * 4);
        const size_t lds0 = (size_t)NI * 52 * sizeof(float);
        rc = set_lds_attr(kern0, lds0, "dsw_sparse_disc_bwd");
        if (rc) return rc;
        hipLaunchKernelGGL(kern0, dim3(nblk), dim3(WG_THREADS), lds0, st, dt0, (const TX *)x, H, W, C, part);
    });
    rc = check_launch("sd_bwd_dx/w0 kernels");
    if (rc) return rc;
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((np + PR_COLS - 1) / PR_COLS), dim3(256), 0, st, part, nblk, np, packed);
    rc = check_launch("partial_reduce_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(copy2_kernel, dim3((np + 255) / 256), dim3(256), 0, st, packed, 4 * C * 9, 4, dw0, db0);
    return check_launch("copy2_kernel");
}

// floats of scratch dsw_sparse_disc_bwd needs
extern "C" size_t dsw_sparse_disc_bwd_scratch(int B, int H, int W, int C, int w) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || w <= 0) return 0;
    const int n = B * 4 * (H / w) * (W / w);
    const size_t np = (size_t)4 * C * 9 + 4;
    return (size_t)((n + 3) & ~3) + (size_t)B * H * W * 4 + (size_t)dsw_sparse_disc_bwd_blocks(B, H) * np + np;
}

extern "C" int dsw_colsum_blocks(long long rows) { return rows > 0 ? (int)((rows + CS_ROWS - 1) / CS_ROWS) : 0; }

extern "C" int dsw_colsum(const void *x, int dtype, long long rows, int cols, float *out, float *partial, void *stream) {
    if (rows <= 0 || cols <= 0) return fail(DSW_EINVAL, "dsw_colsum: bad sizes rows=%lld cols=%d", rows, cols);
    if (cols % 4 || cols / 4 > CS_THREADS) return fail(DSW_EUNSUPPORTED, "dsw_colsum: %d columns (multiples of 4 up to %d)", cols, 4 * CS_THREADS);
    DSW_REQUIRE(x && out && partial, "dsw_colsum: null pointer");
    hipStream_t st = as_stream(stream);
    const int nblk = dsw_colsum_blocks(rows);
    const size_t lds = (size_t)(CS_THREADS / (cols / 4)) * cols * sizeof(float);
    int rc;
    DSW_DISPATCH_DTYPE(dtype, T, {
        auto kern = colsum_kernel<T>;
        rc = set_lds_attr(kern, lds, "dsw_colsum");
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(CS_THREADS), lds, st, (const T *)x, rows, cols, partial);
    });
    rc = check_launch("colsum_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((cols + PR_COLS - 1) / PR_COLS), dim3(256), 0, st, partial, nblk, cols, out);
    return check_launch("partial_reduce_kernel");
}

// ---------------------------------------------------------------------------------------------
// dsw_colsum for callers that cannot allocate (code under HIP graph capture, tight serving loops):
// the per-block partial sums live in a file-scope device array instead of a caller buffer.
// Every GPU the library runs on holds its own copy of the array (4 MiB) at its own address; host
// code must never pass the array's name to a kernel - it uses the per-device address obtained once
// from hipGetSymbolAddress.  All calls on one GPU share that GPU's array: calls that overlap on one
// GPU (two streams, two host threads) race.  Nothing in it has to start at zero: every block of
// colsum_kernel writes all of its partial sums before partial_reduce_kernel reads them.
// ---------------------------------------------------------------------------------------------
constexpr size_t CS_STATIC_FLOATS = (size_t)1 << 20;
static __device__ __align__(256) float scratch[CS_STATIC_FLOATS];

constexpr int CS_MAX_DEVICES = 64;
static float *g_scratch_ptr[CS_MAX_DEVICES];   // per device, filled on the first call there
static std::mutex g_scratch_mutex;

static int static_scratch_for_current_device(float **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(DSW_ELAUNCH, "dsw_colsum_static: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= CS_MAX_DEVICES) return fail(DSW_EUNSUPPORTED, "dsw_colsum_static: device index %d", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));   // the copy of the array on the current device
        if (e != hipSuccess || p == nullptr)
            return fail(DSW_ELAUNCH, "dsw_colsum_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
        g_scratch_ptr[dev] = static_cast<float *>(p);
    }
    *out = g_scratch_ptr[dev];
    return DSW_OK;
}

extern "C" size_t dsw_colsum_static_max_floats(void) { return CS_STATIC_FLOATS; }

extern "C" int dsw_colsum_static(const void *x, int dtype, long long rows, int cols, float *out, void *stream) {
    if (rows <= 0 || cols <= 0) return fail(DSW_EINVAL, "dsw_colsum_static: bad sizes rows=%lld cols=%d", rows, cols);
    const size_t need = (size_t)dsw_colsum_blocks(rows) * (size_t)cols;
    if (need > CS_STATIC_FLOATS)
        return fail(DSW_EUNSUPPORTED, "dsw_colsum_static: rows=%lld cols=%d needs %zu floats of scratch, the static buffer holds %zu "
                    "(use dsw_colsum with a caller buffer)", rows, cols, need, CS_STATIC_FLOATS);
    float *ptr = nullptr;
    const int rc = static_scratch_for_current_device(&ptr);   // the device the caller made current (the stream's device)
    if (rc) return rc;
    return dsw_colsum(x, dtype, rows, cols, out, ptr, stream);
}

extern "C" int dsw_cast_rows(const void *src, int src_dtype, long long src_ld, void *dst, int dst_dtype, long long dst_ld,
                             long long rows, long long src_rows, int cols, int zero_pad, void *stream) {
    if (rows <= 0 || cols <= 0 || src_ld < cols || dst_ld < cols || src_rows < 0 || src_rows > rows)
        return fail(DSW_EINVAL, "dsw_cast_rows: bad sizes rows=%lld cols=%d src_ld=%lld dst_ld=%lld", rows, cols, src_ld, dst_ld);
    if (cols % 4 || src_ld % 4 || dst_ld % 4)
        return fail(DSW_EUNSUPPORTED, "dsw_cast_rows: cols and row strides must be multiples of 4 (%d, %lld, %lld)", cols, src_ld, dst_ld);
    DSW_REQUIRE(src && dst, "dsw_cast_rows: null pointer");
    const int width = zero_pad ? (int)dst_ld : cols;
    const long long items = rows * (width / 4);
    const unsigned nblk = (unsigned)((items + 255) / 256 < 8192 ? (items + 255) / 256 : 8192);
    DSW_DISPATCH_DTYPE(src_dtype, TS, DSW_DISPATCH_DTYPE(dst_dtype, TD, {
        hipLaunchKernelGGL((cast_rows_kernel<TS, TD>), dim3(nblk), dim3(256), 0, as_stream(stream), (const TS *)src, src_ld,
                           (TD *)dst, dst_ld, rows, src_rows, cols, width);
    }));
    return check_launch("cast_rows_kernel");
}
