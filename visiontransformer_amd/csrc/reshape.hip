// MSFT multi-scale reshapes as single-pass kernels (reference: msft_arch.py:37-78, :156-167):
//   unshuffle  : NCHW [B,c,H,W] --PixelUnshuffle(s)--> NHWC [B,H/s,W/s,c*s*s]  (+ LayerNorm)
//   shuffle    : NHWC [B,H,W,C] --> NCHW --PixelShuffle(s)--> [B,C/(s*s),H*s,W*s]
// A block owns TW consecutive output tokens of one image row and stages them in LDS, so both
// the NCHW side (runs of TW*s pixels of one channel row) and the NHWC side (TW whole token rows)
// are read and written in contiguous runs.  s = 1 gives the plain NCHW<->token-major copies.
#include "common.hpp"

namespace dsw {

constexpr int RS_THREADS = 256;
constexpr int RS_TW = 16;

// channel of the unshuffled tensor for (input channel ci, sub-row dy, sub-col dx)
__device__ __forceinline__ int unshuffled_channel(int ci, int dy, int dx, int s) { return (ci * s + dy) * s + dx; }

// NCHW rows -> LDS tile[p][ch] for tokens (b, ho, wo0 + p)
template <typename T>
__device__ __forceinline__ void load_nchw_tile(const T *__restrict__ x, float *tile, int ld, int b, int c, int H,
                                               int W, int s, int ho, int wo0, int ntok) {
    const int run = ntok * s;
    for (int it = threadIdx.x; it < c * s * run; it += RS_THREADS) {
        const int col = it % run;
        const int dy = (it / run) % s;
        const int ci = it / (run * s);
        const float v = io<T>::ld(x + (((size_t)b * c + ci) * H + ho * s + dy) * W + wo0 * s + col);
        tile[(col / s) * ld + unshuffled_channel(ci, dy, col % s, s)] = v;
    }
}

template <typename T>
__device__ __forceinline__ void store_nchw_tile(T *__restrict__ x, const float *tile, int ld, int b, int c, int H,
                                                int W, int s, int ho, int wo0, int ntok) {
    const int run = ntok * s;
    for (int it = threadIdx.x; it < c * s * run; it += RS_THREADS) {
        const int col = it % run;
        const int dy = (it / run) % s;
        const int ci = it / (run * s);
        io<T>::st(x + (((size_t)b * c + ci) * H + ho * s + dy) * W + wo0 * s + col,
                  tile[(col / s) * ld + unshuffled_channel(ci, dy, col % s, s)]);
    }
}

template <typename TI, typename TO, bool LN>
__global__ __launch_bounds__(RS_THREADS) void unshuffle_fwd_kernel(
    const TI *__restrict__ x, int c, int H, int W, int s, int tiles, const float *__restrict__ gamma,
    const float *__restrict__ beta, float eps, TO *__restrict__ y, float *__restrict__ mean,
    float *__restrict__ rstd) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int Ho = H / s, Wo = W / s, Co = c * s * s, ld = Co + 1;
    const int t = blockIdx.x % tiles, ho = (blockIdx.x / tiles) % Ho, b = blockIdx.x / (tiles * Ho);
    const int wo0 = t * RS_TW, ntok = min(RS_TW, Wo - wo0);
    load_nchw_tile<TI>(x, tile, ld, b, c, H, W, s, ho, wo0, ntok);
    __syncthreads();
    const size_t tok0 = ((size_t)b * Ho + ho) * Wo + wo0;
    if constexpr (LN) {
        float *stat = tile + RS_TW * ld;  // [RS_TW][2]
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int p = wave; p < ntok; p += RS_THREADS / 64) {
            float sum = 0.f;
            for (int ch = lane; ch < Co; ch += 64) sum += tile[p * ld + ch];
            const float mu = wave_sum(sum) / (float)Co;
            float var = 0.f;
            for (int ch = lane; ch < Co; ch += 64) {
                const float dlt = tile[p * ld + ch] - mu;
                var += dlt * dlt;
            }
            const float rs = rsqrtf(wave_sum(var) / (float)Co + eps);
            if (lane == 0) {
                stat[2 * p] = mu;
                stat[2 * p + 1] = rs;
                mean[tok0 + p] = mu;
                rstd[tok0 + p] = rs;
            }
        }
        __syncthreads();
        for (int it = threadIdx.x; it < ntok * Co; it += RS_THREADS) {
            const int p = it / Co, ch = it - p * Co;
            io<TO>::st(y + (tok0 + p) * Co + ch, (tile[p * ld + ch] - stat[2 * p]) * stat[2 * p + 1] * gamma[ch] + beta[ch]);
        }
    } else {
        for (int it = threadIdx.x; it < ntok * Co; it += RS_THREADS) {
            const int p = it / Co, ch = it - p * Co;
            io<TO>::st(y + (tok0 + p) * Co + ch, tile[p * ld + ch]);
        }
    }
}

// shuffle: token-major [B,H,W,C] -> NCHW + PixelShuffle(s).  Here (H, W) is the token grid and the
// output image is (H*s, W*s) with c = C/(s*s) channels: exactly store_nchw_tile's mapping.
template <typename T>
__global__ __launch_bounds__(RS_THREADS) void shuffle_fwd_kernel(const T *__restrict__ tkn, int Hs, int Ws, int C,
                                                                 int s, int tiles, T *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int c = C / (s * s), ld = C + 1;
    const int t = blockIdx.x % tiles, ho = (blockIdx.x / tiles) % Hs, b = blockIdx.x / (tiles * Hs);
    const int wo0 = t * RS_TW, ntok = min(RS_TW, Ws - wo0);
    const size_t tok0 = ((size_t)b * Hs + ho) * Ws + wo0;
    for (int it = threadIdx.x; it < ntok * C; it += RS_THREADS) {
        const int p = it / C, ch = it - p * C;
        tile[p * ld + ch] = io<T>::ld(tkn + (tok0 + p) * C + ch);
    }
    __syncthreads();
    store_nchw_tile<T>(y, tile, ld, b, c, Hs * s, Ws * s, s, ho, wo0, ntok);
}

// LayerNorm backward fused with the inverse reshape.  dy, and the recomputed x-hat, are staged per
// token; dx goes back to NCHW; per-block partial sums of d_gamma / d_beta go to `partial`.
template <typename T>
__global__ __launch_bounds__(RS_THREADS) void unshuffle_ln_bwd_kernel(
    const T *__restrict__ dy, const T *__restrict__ x, int c, int H, int W, int s, int tiles,
    const float *__restrict__ gamma, const float *__restrict__ mean, const float *__restrict__ rstd,
    T *__restrict__ dx, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Ho = H / s, Wo = W / s, Co = c * s * s, ld = Co + 1;
    float *xh = smem;                 // [RS_TW][ld]  x-hat
    float *gg = smem + RS_TW * ld;    // [RS_TW][ld]  dy, then dx
    const int t = blockIdx.x % tiles, ho = (blockIdx.x / tiles) % Ho, b = blockIdx.x / (tiles * Ho);
    const int wo0 = t * RS_TW, ntok = min(RS_TW, Wo - wo0);
    const size_t tok0 = ((size_t)b * Ho + ho) * Wo + wo0;
    load_nchw_tile<T>(x, xh, ld, b, c, H, W, s, ho, wo0, ntok);
    for (int it = threadIdx.x; it < ntok * Co; it += RS_THREADS) {
        const int p = it / Co, ch = it - p * Co;
        gg[p * ld + ch] = io<T>::ld(dy + (tok0 + p) * Co + ch);
    }
    __syncthreads();
    for (int it = threadIdx.x; it < ntok * Co; it += RS_THREADS) {
        const int p = it / Co, ch = it - p * Co;
        xh[p * ld + ch] = (xh[p * ld + ch] - mean[tok0 + p]) * rstd[tok0 + p];
    }
    __syncthreads();
    // parameter gradients: fixed order over the block's tokens
    for (int ch = threadIdx.x; ch < Co; ch += RS_THREADS) {
        float dg = 0.f, db = 0.f;
        for (int p = 0; p < ntok; ++p) {
            dg += gg[p * ld + ch] * xh[p * ld + ch];
            db += gg[p * ld + ch];
        }
        partial[((size_t)blockIdx.x * 2 + 0) * Co + ch] = dg;
        partial[((size_t)blockIdx.x * 2 + 1) * Co + ch] = db;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int p = wave; p < ntok; p += RS_THREADS / 64) {
        float s1 = 0.f, s2 = 0.f;
        for (int ch = lane; ch < Co; ch += 64) {
            const float g = gg[p * ld + ch] * gamma[ch];
            s1 += g;
            s2 += g * xh[p * ld + ch];
        }
        s1 = wave_sum(s1) / (float)Co;
        s2 = wave_sum(s2) / (float)Co;
        const float rs = rstd[tok0 + p];
        for (int ch = lane; ch < Co; ch += 64) {
            const float g = gg[p * ld + ch] * gamma[ch];
            gg[p * ld + ch] = rs * (g - s1 - xh[p * ld + ch] * s2);
        }
    }
    __syncthreads();
    store_nchw_tile<T>(dx, gg, ld, b, c, H, W, s, ho, wo0, ntok);
}

// (thousands of partial rows, a few hundred columns: 32 row slices per block keep enough loads in flight)
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float *__restrict__ partial, int nblocks, int Co,
                                                               float *__restrict__ d_gamma, float *__restrict__ d_beta) {
    __shared__ float red[32][33];
    int i;
    float acc;
    if (!column_sum_32<32>(partial, nblocks, 2 * Co, red, i, acc)) return;  // a partial row is [d_gamma | d_beta]
    const int which = i / Co, ch = i - which * Co;
    (which ? d_beta : d_gamma)[ch] = acc;
}

static int check_reshape(const char *who, int B, int c, int H, int W, int s) {
    if (B <= 0 || c <= 0 || H <= 0 || W <= 0 || s <= 0) return fail(DSW_EINVAL, "%s: bad sizes B=%d c=%d H=%d W=%d s=%d", who, B, c, H, W, s);
    if (H % s || W % s) return fail(DSW_EINVAL, "%s: H,W (%d,%d) must be multiples of the scale %d", who, H, W, s);
    if (c * s * s > 2048) return fail(DSW_EUNSUPPORTED, "%s: %d channels after unshuffle > 2048", who, c * s * s);
    return DSW_OK;
}

template <typename K> static int set_lds(K kern, size_t lds, const char *who) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DSW_ELAUNCH, "%s: LDS attribute: %s", who, hipGetErrorString(e));
    }
    return DSW_OK;
}

}  // namespace dsw

using namespace dsw;

extern "C" int dsw_unshuffle_nhwc_ln_fwd(const void *x, int x_dtype, int B, int c, int H, int W, int s,
                                         const float *gamma, const float *beta, float eps, void *y, int y_dtype,
                                         float *mean, float *rstd, void *stream) {
    int rc = check_reshape("dsw_unshuffle_nhwc_ln_fwd", B, c, H, W, s);
    if (rc) return rc;
    DSW_REQUIRE(x && y && gamma && beta && mean && rstd, "dsw_unshuffle_nhwc_ln_fwd: null pointer");
    const int Co = c * s * s, tiles = (W / s + RS_TW - 1) / RS_TW;
    const size_t lds = (size_t)(RS_TW * (Co + 1) + 2 * RS_TW) * sizeof(float);
    const dim3 grid(B * (H / s) * tiles);
    DSW_DISPATCH_DTYPE(x_dtype, TI, DSW_DISPATCH_DTYPE(y_dtype, TO, {
        auto kern = unshuffle_fwd_kernel<TI, TO, true>;
        rc = set_lds(kern, lds, "dsw_unshuffle_nhwc_ln_fwd");
        if (rc) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(RS_THREADS), lds, as_stream(stream), (const TI *)x, c, H, W, s, tiles, gamma,
                           beta, eps, (TO *)y, mean, rstd);
    }));
    return check_launch("unshuffle_fwd_kernel<LN>");
}

extern "C" int dsw_unshuffle_nhwc_fwd(const void *x, int dtype, int B, int c, int H, int W, int s, void *y,
                                      void *stream) {
    int rc = check_reshape("dsw_unshuffle_nhwc_fwd", B, c, H, W, s);
    if (rc) return rc;
    DSW_REQUIRE(x && y, "dsw_unshuffle_nhwc_fwd: null pointer");
    const int Co = c * s * s, tiles = (W / s + RS_TW - 1) / RS_TW;
    const size_t lds = (size_t)(RS_TW * (Co + 1)) * sizeof(float);
    const dim3 grid(B * (H / s) * tiles);
    DSW_DISPATCH_DTYPE(dtype, T, {
        auto kern = unshuffle_fwd_kernel<T, T, false>;
        rc = set_lds(kern, lds, "dsw_unshuffle_nhwc_fwd");
        if (rc) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(RS_THREADS), lds, as_stream(stream), (const T *)x, c, H, W, s, tiles,
                           (const float *)nullptr, (const float *)nullptr, 0.f, (T *)y, (float *)nullptr, (float *)nullptr);
    });
    return check_launch("unshuffle_fwd_kernel");
}

extern "C" int dsw_shuffle_nchw_fwd(const void *t, int dtype, int B, int H, int W, int C, int s, void *y,
                                    void *stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || s <= 0 || C % (s * s)) return fail(DSW_EINVAL, "dsw_shuffle_nchw_fwd: bad sizes B=%d H=%d W=%d C=%d s=%d", B, H, W, C, s);
    if (C > 2048) return fail(DSW_EUNSUPPORTED, "dsw_shuffle_nchw_fwd: %d channels > 2048", C);
    DSW_REQUIRE(t && y, "dsw_shuffle_nchw_fwd: null pointer");
    const int tiles = (W + RS_TW - 1) / RS_TW;
    const size_t lds = (size_t)(RS_TW * (C + 1)) * sizeof(float);
    const dim3 grid(B * H * tiles);
    int rc;
    DSW_DISPATCH_DTYPE(dtype, T, {
        auto kern = shuffle_fwd_kernel<T>;
        rc = set_lds(kern, lds, "dsw_shuffle_nchw_fwd");
        if (rc) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(RS_THREADS), lds, as_stream(stream), (const T *)t, H, W, C, s, tiles, (T *)y);
    });
    return check_launch("shuffle_fwd_kernel");
}

extern "C" int dsw_unshuffle_nhwc_ln_bwd(const void *dy, const void *x, int dtype, int B, int c, int H, int W,
                                         int s, const float *gamma, const float *mean, const float *rstd,
                                         void *dx, float *d_gamma, float *d_beta, float *partial,
                                         void *stream) {
    int rc = check_reshape("dsw_unshuffle_nhwc_ln_bwd", B, c, H, W, s);
    if (rc) return rc;
    DSW_REQUIRE(dy && x && gamma && mean && rstd && dx && d_gamma && d_beta && partial, "dsw_unshuffle_nhwc_ln_bwd: null pointer");
    const int Co = c * s * s, tiles = (W / s + RS_TW - 1) / RS_TW;
    const size_t lds = (size_t)(2 * RS_TW * (Co + 1)) * sizeof(float);
    if (lds > 160 * 1024) return fail(DSW_EUNSUPPORTED, "dsw_unshuffle_nhwc_ln_bwd: %d channels need %zu B of LDS", Co, lds);
    const int nblocks = B * (H / s) * tiles;
    hipStream_t st = as_stream(stream);
    DSW_DISPATCH_DTYPE(dtype, T, {
        auto kern = unshuffle_ln_bwd_kernel<T>;
        rc = set_lds(kern, lds, "dsw_unshuffle_nhwc_ln_bwd");
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(RS_THREADS), lds, st, (const T *)dy, (const T *)x, c, H, W, s, tiles,
                           gamma, mean, rstd, (T *)dx, partial);
    });
    rc = check_launch("unshuffle_ln_bwd_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * Co + 31) / 32), dim3(1024), 0, st, partial, nblocks, Co, d_gamma, d_beta);
    return check_launch("ln_param_reduce_kernel");
}

// number of blocks of the LayerNorm backward = rows of its `partial` scratch (2*Co floats each)
extern "C" int dsw_unshuffle_blocks(int B, int H, int W, int s) {
    if (B <= 0 || H <= 0 || W <= 0 || s <= 0) return 0;
    return B * (H / s) * ((W / s + RS_TW - 1) / RS_TW);
}
