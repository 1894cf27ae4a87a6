// RMSNorm over the channel rows of the fp32 residual stream (the nn.RMSNorm layers in front of the
// attention modules and the MLP, DSwinT_arch.py:343-345, :355-357):
//   y[m][c] = x[m][c] * rstd[m] * w[c],   rstd[m] = rsqrt(mean_c x[m][c]^2 + eps)
// written straight in the dtype and row stride its consumer wants (bf16 rows padded to 64 elements for
// the projection kernels, padding zeroed; fp32 for DS-MSA), so the normalised tensor is never written
// in fp32 and cast in a second pass.  16 lanes per row, a row's values stay in registers.
#include "common.hpp"

namespace dsw {
namespace {

constexpr int RN_LPR = 16;               // lanes per row
constexpr int RN_RPB = 256 / RN_LPR;     // rows per block pass
constexpr int RN_MAXV = 4;               // float4 per lane: C <= 256

__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename TO>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w, float eps, long long M,
                                                          int C, TO *__restrict__ out, long long ldo, float *__restrict__ rstd) {
    const int sub = threadIdx.x & (RN_LPR - 1), rloc = threadIdx.x / RN_LPR;
    const int nv = C / 4, nvo = (int)(ldo / 4);
    float4 wv[RN_MAXV];
#pragma unroll
    for (int i = 0; i < RN_MAXV; ++i) wv[i] = sub + RN_LPR * i < nv ? *reinterpret_cast<const float4 *>(w + 4 * (sub + RN_LPR * i)) : f4zero();
    for (long long r0 = (long long)blockIdx.x * RN_RPB; r0 < M; r0 += (long long)gridDim.x * RN_RPB) {
        const long long row = r0 + rloc;
        const bool ok = row < M;
        float4 v[RN_MAXV];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < RN_MAXV; ++i) {
            v[i] = ok && sub + RN_LPR * i < nv ? *reinterpret_cast<const float4 *>(x + row * C + 4 * (sub + RN_LPR * i)) : f4zero();
            ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
        ss = group16_sum(ss);
        const float rs = rsqrtf(ss / (float)C + eps);
        if (ok) {
            if (sub == 0 && rstd != nullptr) rstd[row] = rs;
#pragma unroll
            for (int i = 0; i < RN_MAXV; ++i) {
                const int j = sub + RN_LPR * i;
                if (j < nvo) {
                    const float4 y = j < nv ? make_float4(v[i].x * rs * wv[i].x, v[i].y * rs * wv[i].y, v[i].z * rs * wv[i].z, v[i].w * rs * wv[i].w)
                                            : f4zero();
                    io<TO>::st4(out + row * ldo + 4 * j, y);
                }
            }
        }
    }
}

// dx[m][c] = rstd (g_c - xh_c * mean_c(g xh)),  g = dy w,  xh = x rstd;   dw[c] = sum_m dy[m][c] xh[m][c]
// (per-block partial sums of dw, reduced by the caller in a fixed order)
template <typename TG>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const TG *__restrict__ dy, long long ldg, const float *__restrict__ x,
                                                          const float *__restrict__ w, const float *__restrict__ rstd, long long M, int C,
                                                          float *__restrict__ dx, const float *__restrict__ dx_add, float *__restrict__ partial) {
    __shared__ float4 red[RN_RPB][RN_LPR * RN_MAXV];
    const int sub = threadIdx.x & (RN_LPR - 1), rloc = threadIdx.x / RN_LPR;
    const int nv = C / 4;
    float4 wv[RN_MAXV], dwacc[RN_MAXV];
#pragma unroll
    for (int i = 0; i < RN_MAXV; ++i) {
        wv[i] = sub + RN_LPR * i < nv ? *reinterpret_cast<const float4 *>(w + 4 * (sub + RN_LPR * i)) : f4zero();
        dwacc[i] = f4zero();
    }
    for (long long r0 = (long long)blockIdx.x * RN_RPB; r0 < M; r0 += (long long)gridDim.x * RN_RPB) {
        const long long row = r0 + rloc;
        const bool ok = row < M;
        const float rs = ok ? rstd[row] : 0.f;
        float4 xh[RN_MAXV], g[RN_MAXV], av[RN_MAXV];
        float dot = 0.f;
        // the gradient that reached x around the norm is asked for with the row's other operands, ahead of the reduction
#pragma unroll
        for (int i = 0; i < RN_MAXV; ++i)
            av[i] = ok && dx_add != nullptr && sub + RN_LPR * i < nv
                        ? *reinterpret_cast<const float4 *>(dx_add + (size_t)row * C + 4 * (sub + RN_LPR * i)) : f4zero();
#pragma unroll
        for (int i = 0; i < RN_MAXV; ++i) {
            const bool live = ok && sub + RN_LPR * i < nv;
            const float4 xv = live ? *reinterpret_cast<const float4 *>(x + row * C + 4 * (sub + RN_LPR * i)) : f4zero();
            const float4 gy = live ? io<TG>::ld4(dy + row * ldg + 4 * (sub + RN_LPR * i)) : f4zero();
            xh[i] = xv * rs;
            g[i] = make_float4(gy.x * wv[i].x, gy.y * wv[i].y, gy.z * wv[i].z, gy.w * wv[i].w);
            dot += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
            dwacc[i] = dwacc[i] + make_float4(gy.x * xh[i].x, gy.y * xh[i].y, gy.z * xh[i].z, gy.w * xh[i].w);
        }
        dot = group16_sum(dot) / (float)C;
        if (ok) {
#pragma unroll
            for (int i = 0; i < RN_MAXV; ++i)
                if (sub + RN_LPR * i < nv) {
                    const size_t at = (size_t)row * C + 4 * (sub + RN_LPR * i);
                    *reinterpret_cast<float4 *>(dx + at) = av[i] + (g[i] - xh[i] * dot) * rs;
                }
        }
    }
#pragma unroll
    for (int i = 0; i < RN_MAXV; ++i) red[rloc][sub + RN_LPR * i] = dwacc[i];
    __syncthreads();
    if (threadIdx.x < RN_LPR * RN_MAXV) {
        // slot [i][s] of the stacked layout holds float4 index s + 16 i
        const int i = threadIdx.x / RN_LPR, s = threadIdx.x % RN_LPR, j = s + RN_LPR * i;
        if (j < nv) {
            float4 a = f4zero();
#pragma unroll
            for (int r = 0; r < RN_RPB; ++r) a = a + red[r][s + RN_LPR * i];
            *reinterpret_cast<float4 *>(partial + (size_t)blockIdx.x * C + 4 * j) = a;
        }
    }
}

int rn_blocks(long long M) {
    const long long want = (M + RN_RPB - 1) / RN_RPB;
    return (int)(want < 2048 ? want : 2048);
}

}  // namespace
}  // namespace dsw

using namespace dsw;

extern "C" int dsw_rmsnorm_blocks(long long M) { return M > 0 ? rn_blocks(M) : 0; }

extern "C" int dsw_rmsnorm_fwd(const float *x, const float *weight, float eps, long long M, int C, void *out, int out_dtype,
                               long long ldo, float *rstd, void *stream) {
    if (M <= 0 || C <= 0 || ldo < C) return fail(DSW_EINVAL, "dsw_rmsnorm_fwd: bad sizes M=%lld C=%d ldo=%lld", M, C, ldo);
    if (C % 4 || ldo % 4 || C > 4 * RN_LPR * RN_MAXV || ldo > 4 * RN_LPR * RN_MAXV)
        return fail(DSW_EUNSUPPORTED, "dsw_rmsnorm_fwd: C and the output row stride must be multiples of 4, at most %d (C=%d ldo=%lld)",
                    4 * RN_LPR * RN_MAXV, C, ldo);
    DSW_REQUIRE(x && weight && out, "dsw_rmsnorm_fwd: null pointer");
    DSW_DISPATCH_DTYPE(out_dtype, TO, {
        hipLaunchKernelGGL(rmsnorm_fwd_kernel<TO>, dim3(rn_blocks(M)), dim3(256), 0, as_stream(stream), x, weight, eps, M, C, (TO *)out, ldo,
                           rstd);
    });
    return check_launch("rmsnorm_fwd_kernel");
}

extern "C" int dsw_rmsnorm_bwd(const void *dy, int dy_dtype, long long ld_dy, const float *x, const float *weight, const float *rstd,
                               long long M, int C, float *dx, const float *dx_add, float *partial, void *stream) {
    if (M <= 0 || C <= 0 || ld_dy < C) return fail(DSW_EINVAL, "dsw_rmsnorm_bwd: bad sizes M=%lld C=%d ld_dy=%lld", M, C, ld_dy);
    if (C % 4 || ld_dy % 4 || C > 4 * RN_LPR * RN_MAXV)
        return fail(DSW_EUNSUPPORTED, "dsw_rmsnorm_bwd: C and the gradient's row stride must be multiples of 4, C at most %d (C=%d ld_dy=%lld)",
                    4 * RN_LPR * RN_MAXV, C, ld_dy);
    DSW_REQUIRE(dy && x && weight && rstd && dx && partial, "dsw_rmsnorm_bwd: null pointer");
    DSW_DISPATCH_DTYPE(dy_dtype, TG, {
        hipLaunchKernelGGL(rmsnorm_bwd_kernel<TG>, dim3(rn_blocks(M)), dim3(256), 0, as_stream(stream), (const TG *)dy, ld_dy, x, weight, rstd,
                           M, C, dx, dx_add, partial);
    });
    return check_launch("rmsnorm_bwd_kernel");
}

// ---- gated SiLU of the MLP (DSwinT_arch.py:47-69: fc2(act(fc1_gate(x)) * fc1_content(x))) ----------
// h holds gate | content side by side in one row (the two fc1 projections computed as one product):
//   out[m][j] = silu(h[m][j]) * h[m][H + j],  j < H;  columns H..ldo-1 of out are set to zero.
namespace dsw {
namespace {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

template <typename TH, typename TO>
__global__ __launch_bounds__(256) void gated_silu_fwd_kernel(const TH *__restrict__ h, long long ldh, int H, long long M,
                                                             TO *__restrict__ out, long long ldo) {
    const int per_row = (int)(ldo / 4);
    const long long items = M * per_row;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const long long m = it / per_row;
        const int j = (int)(it - m * per_row) * 4;
        float4 y = f4zero();
        if (j < H) {
            const float4 g = io<TH>::ld4(h + m * ldh + j), c = io<TH>::ld4(h + m * ldh + H + j);
            y = make_float4(gated_silu_value(g.x, c.x), gated_silu_value(g.y, c.y), gated_silu_value(g.z, c.z), gated_silu_value(g.w, c.w));
        }
        io<TO>::st4(out + m * ldo + j, y);
    }
}

// dh[m][j] = dout c s (1 + g (1 - s)),  dh[m][H + j] = dout g s,  s = sigmoid(g);  columns 2H..ldh-1 := 0
template <typename TH, typename TO>
__global__ __launch_bounds__(256) void gated_silu_bwd_kernel(const TO *__restrict__ dout, long long ldg, const TH *__restrict__ h,
                                                             long long ldh, int H, long long M, TH *__restrict__ dh) {
    const int per_row = (int)(ldh / 4);
    const long long items = M * per_row;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const long long m = it / per_row;
        const int j = (int)(it - m * per_row) * 4;
        float4 r = f4zero();
        if (j < 2 * H) {
            const int jj = j < H ? j : j - H;
            const float4 g = io<TH>::ld4(h + m * ldh + jj), c = io<TH>::ld4(h + m * ldh + H + jj), d = io<TO>::ld4(dout + m * ldg + jj);
            const float gs[4] = {g.x, g.y, g.z, g.w}, cs[4] = {c.x, c.y, c.z, c.w}, ds[4] = {d.x, d.y, d.z, d.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s = sigmoidf_(gs[e]);
                o[e] = j < H ? ds[e] * cs[e] * s * (1.f + gs[e] * (1.f - s)) : ds[e] * gs[e] * s;
            }
            r = make_float4(o[0], o[1], o[2], o[3]);
        }
        io<TH>::st4(dh + m * ldh + j, r);
    }
}

int gs_check(const char *who, long long M, int H, long long ldh, long long ld2) {
    if (M <= 0 || H <= 0 || ldh < 2 * H || ld2 < H) return fail(DSW_EINVAL, "%s: bad sizes M=%lld H=%d ld=%lld / %lld", who, M, H, ldh, ld2);
    if (H % 4 || ldh % 4 || ld2 % 4) return fail(DSW_EUNSUPPORTED, "%s: H and the row strides must be multiples of 4 (H=%d, %lld, %lld)", who, H, ldh, ld2);
    return DSW_OK;
}
unsigned gs_blocks(long long items) { return (unsigned)((items + 255) / 256 < 16384 ? (items + 255) / 256 : 16384); }

}  // namespace
}  // namespace dsw

extern "C" int dsw_gated_silu_fwd(const void *h, int h_dtype, long long ldh, int H, long long M, void *out, int out_dtype, long long ldo,
                                  void *stream) {
    int rc = gs_check("dsw_gated_silu_fwd", M, H, ldh, ldo);
    if (rc) return rc;
    DSW_REQUIRE(h && out, "dsw_gated_silu_fwd: null pointer");
    DSW_DISPATCH_DTYPE(h_dtype, TH, DSW_DISPATCH_DTYPE(out_dtype, TO, {
        hipLaunchKernelGGL((gated_silu_fwd_kernel<TH, TO>), dim3(gs_blocks(M * (ldo / 4))), dim3(256), 0, as_stream(stream), (const TH *)h, ldh,
                           H, M, (TO *)out, ldo);
    }));
    return check_launch("gated_silu_fwd_kernel");
}

extern "C" int dsw_gated_silu_bwd(const void *dout, int dout_dtype, long long ld_dout, const void *h, int h_dtype, long long ldh, int H,
                                  long long M, void *dh, void *stream) {
    int rc = gs_check("dsw_gated_silu_bwd", M, H, ldh, ld_dout);
    if (rc) return rc;
    DSW_REQUIRE(dout && h && dh, "dsw_gated_silu_bwd: null pointer");
    DSW_DISPATCH_DTYPE(h_dtype, TH, DSW_DISPATCH_DTYPE(dout_dtype, TO, {
        hipLaunchKernelGGL((gated_silu_bwd_kernel<TH, TO>), dim3(gs_blocks(M * (ldh / 4))), dim3(256), 0, as_stream(stream), (const TO *)dout,
                           ld_dout, (const TH *)h, ldh, H, M, (TH *)dh);
    }));
    return check_launch("gated_silu_bwd_kernel");
}

// ---- GELU of msft's MLP (msft_arch.py:25: nn.GELU(), the erf form) on rows with their own stride ------------------
// y[m][j] = x/2 * (1 + erf(x / sqrt 2)), j < C; columns C..ldy-1 := 0.  Backward: dx = dy * (Phi(x) + x * phi(x)).
// The projections hand over and take views of zero-padded rows; an element-wise torch op on such a view runs in its
// strided kernel at a third of the rate.
namespace dsw {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T *__restrict__ x, long long ldx, int C, long long M, T *__restrict__ y,
                                                       long long ldy) {
    const int per_row = (int)(ldy / 4);
    const long long items = M * per_row;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const long long m = it / per_row;
        const int j = (int)(it - m * per_row) * 4;
        float4 r = f4zero();
        if (j < C) {
            const float4 v = io<T>::ld4(x + m * ldx + j);
            r = make_float4(gelu_value<T>(v.x), gelu_value<T>(v.y), gelu_value<T>(v.z), gelu_value<T>(v.w));
        }
        io<T>::st4(y + m * ldy + j, r);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T *__restrict__ dy, long long ld_dy, const T *__restrict__ x, long long ldx,
                                                       int C, long long M, T *__restrict__ dx, long long ld_dx) {
    const int per_row = (int)(ld_dx / 4);
    const long long items = M * per_row;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const long long m = it / per_row;
        const int j = (int)(it - m * per_row) * 4;
        float4 r = f4zero();
        if (j < C) {
            const float4 v = io<T>::ld4(x + m * ldx + j), g = io<T>::ld4(dy + m * ld_dy + j);
            r = make_float4(g.x * gelu_slope<T>(v.x), g.y * gelu_slope<T>(v.y), g.z * gelu_slope<T>(v.z), g.w * gelu_slope<T>(v.w));
        }
        io<T>::st4(dx + m * ld_dx + j, r);
    }
}

int gelu_check(const char *who, long long M, int C, long long l0, long long l1, long long l2) {
    if (M <= 0 || C <= 0 || l0 < C || l1 < C || l2 < C) return fail(DSW_EINVAL, "%s: bad sizes M=%lld C=%d ld=%lld / %lld / %lld", who, M, C, l0, l1, l2);
    if (C % 4 || l0 % 4 || l1 % 4 || l2 % 4) return fail(DSW_EUNSUPPORTED, "%s: C and the row strides must be multiples of 4 (C=%d, %lld, %lld, %lld)", who, C, l0, l1, l2);
    return DSW_OK;
}

}  // namespace
}  // namespace dsw

extern "C" int dsw_gelu_fwd(const void *x, int dtype, long long ldx, int C, long long M, void *y, long long ldy, void *stream) {
    int rc = gelu_check("dsw_gelu_fwd", M, C, ldx, ldy, ldy);
    if (rc) return rc;
    DSW_REQUIRE(x && y, "dsw_gelu_fwd: null pointer");
    DSW_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL(gelu_fwd_kernel<T>, dim3(gs_blocks(M * (ldy / 4))), dim3(256), 0, as_stream(stream), (const T *)x, ldx, C, M, (T *)y, ldy);
    });
    return check_launch("gelu_fwd_kernel");
}

extern "C" int dsw_gelu_bwd(const void *dy, long long ld_dy, const void *x, int dtype, long long ldx, int C, long long M, void *dx,
                            long long ld_dx, void *stream) {
    int rc = gelu_check("dsw_gelu_bwd", M, C, ld_dy, ldx, ld_dx);
    if (rc) return rc;
    DSW_REQUIRE(dy && x && dx, "dsw_gelu_bwd: null pointer");
    DSW_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL(gelu_bwd_kernel<T>, dim3(gs_blocks(M * (ld_dx / 4))), dim3(256), 0, as_stream(stream), (const T *)dy, ld_dy,
                           (const T *)x, ldx, C, M, (T *)dx, ld_dx);
    });
    return check_launch("gelu_bwd_kernel");
}

// ---- up-sampling stage of the reconstruction tail on NHWC (DSwinT_arch.py:426-437: Conv2d -> PixelShuffle(2) -> GELU) ----
// x [B][H][W][4c] holds, per pixel, its four sub-pixels (r1, r2) as runs of c channels - the convolution that
// produced it had its output channels ordered (r1, r2, channel) - and y [B][2H][2W][c] receives
// y[b][2h + r1][2w + r2][:] = gelu(x[b][h][w][(2 r1 + r2) c : (2 r1 + r2 + 1) c]): the shuffle is the addressing of
// the activation's pass, both sides move whole runs of c elements.
namespace dsw {
namespace {
template <typename T>
__global__ __launch_bounds__(256) void shuffle2_gelu_kernel(const T *__restrict__ x, int H, int W, int c, long long items, T *__restrict__ y) {
    const int cq = c >> 2;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const int j = (int)(it % cq);
        long long t = it / cq;
        const int ow = (int)(t % (2 * W));
        t /= 2 * W;
        const int oh = (int)(t % (2 * H));
        const long long b = t / (2 * H);
        const long long src = ((((b * H + (oh >> 1)) * W + (ow >> 1)) * 4 + (oh & 1) * 2 + (ow & 1)) * (long long)c) + 4 * j;
        const float4 v = io<T>::ld4(x + src);
        io<T>::st4(y + it * 4, make_float4(gelu_value<T>(v.x), gelu_value<T>(v.y), gelu_value<T>(v.z), gelu_value<T>(v.w)));
    }
}
}  // namespace
}  // namespace dsw

namespace dsw {
namespace {
// backward of the stage above: dx[b][h][w][(2 r1 + r2) c + j] = dy[b][2h + r1][2w + r2][j] * gelu'(x[...]); the item
// index runs over dy / y, as in the forward
template <typename T>
__global__ __launch_bounds__(256) void shuffle2_gelu_bwd_kernel(const T *__restrict__ dy, const T *__restrict__ x, int H, int W, int c,
                                                                long long items, T *__restrict__ dx) {
    const int cq = c >> 2;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const int j = (int)(it % cq);
        long long t = it / cq;
        const int ow = (int)(t % (2 * W));
        t /= 2 * W;
        const int oh = (int)(t % (2 * H));
        const long long b = t / (2 * H);
        const long long src = ((((b * H + (oh >> 1)) * W + (ow >> 1)) * 4 + (oh & 1) * 2 + (ow & 1)) * (long long)c) + 4 * j;
        const float4 v = io<T>::ld4(x + src), g = io<T>::ld4(dy + it * 4);
        io<T>::st4(dx + src, make_float4(g.x * gelu_slope<T>(v.x), g.y * gelu_slope<T>(v.y), g.z * gelu_slope<T>(v.z), g.w * gelu_slope<T>(v.w)));
    }
}
}  // namespace
}  // namespace dsw

extern "C" int dsw_shuffle2_gelu_nhwc_bwd(const void *dy, const void *x, int dtype, int B, int H, int W, int c, void *dx, void *stream) {
    if (B <= 0 || H <= 0 || W <= 0 || c <= 0) return fail(DSW_EINVAL, "dsw_shuffle2_gelu_nhwc_bwd: bad sizes B=%d H=%d W=%d c=%d", B, H, W, c);
    if (c % 4) return fail(DSW_EUNSUPPORTED, "dsw_shuffle2_gelu_nhwc_bwd: c=%d must be a multiple of 4", c);
    DSW_REQUIRE(dy && x && dx, "dsw_shuffle2_gelu_nhwc_bwd: null pointer");
    const long long items = (long long)B * H * W * c;
    DSW_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL(shuffle2_gelu_bwd_kernel<T>, dim3(gs_blocks(items)), dim3(256), 0, as_stream(stream), (const T *)dy, (const T *)x, H, W, c,
                           items, (T *)dx);
    });
    return check_launch("shuffle2_gelu_bwd_kernel");
}

extern "C" int dsw_shuffle2_gelu_nhwc(const void *x, int dtype, int B, int H, int W, int c, void *y, void *stream) {
    if (B <= 0 || H <= 0 || W <= 0 || c <= 0) return fail(DSW_EINVAL, "dsw_shuffle2_gelu_nhwc: bad sizes B=%d H=%d W=%d c=%d", B, H, W, c);
    if (c % 4) return fail(DSW_EUNSUPPORTED, "dsw_shuffle2_gelu_nhwc: c=%d must be a multiple of 4", c);
    DSW_REQUIRE(x && y, "dsw_shuffle2_gelu_nhwc: null pointer");
    const long long items = (long long)B * H * W * c;  // 4 sub-pixels x c / 4 groups of four
    DSW_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL(shuffle2_gelu_kernel<T>, dim3(gs_blocks(items)), dim3(256), 0, as_stream(stream), (const T *)x, H, W, c, items, (T *)y);
    });
    return check_launch("shuffle2_gelu_kernel");
}

// ---- ReZero residual of the reference's ViTBlock (DSwinT_arch.py:355-357): y = x + r * branch -------
// x, y: the fp32 stream [M][C]; branch: fp32 or bf16 rows with their own stride (the padded views the
// attention modules and the MLP return); r: one fp32 parameter on the device.
namespace dsw {
namespace {

// inverted dropout on 4 consecutive elements: mask bytes (0 = dropped) of a dense [M][C] byte matrix
__device__ __forceinline__ float4 drop4(float4 v, uint32_t m4, float keep) {
    return make_float4((m4 & 0xffu) ? v.x * keep : 0.f, (m4 & 0xff00u) ? v.y * keep : 0.f, (m4 & 0xff0000u) ? v.z * keep : 0.f,
                       (m4 & 0xff000000u) ? v.w * keep : 0.f);
}

template <typename TB>
__global__ __launch_bounds__(256) void rezero_fwd_kernel(const float *__restrict__ x, const TB *__restrict__ b, long long ldb,
                                                         const float *__restrict__ r, const unsigned char *__restrict__ mask, float keep,
                                                         DropArgs drop, long long M, int C, float *__restrict__ y) {
    const float rv = *r;
    const int per_row = C / 4;
    const long long items = M * per_row;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const long long m = it / per_row;
        const int j = (int)(it - m * per_row) * 4;
        const float4 xv = *reinterpret_cast<const float4 *>(x + m * C + j);
        float4 bv = io<TB>::ld4(b + m * ldb + j);
        if (mask != nullptr) bv = drop4(bv, *reinterpret_cast<const uint32_t *>(mask + m * C + j), keep);
        else if (drop.thr) bv = drop4_seeded(bv, m * C + j, drop);
        *reinterpret_cast<float4 *>(y + m * C + j) = xv + bv * rv;
    }
}

// db = r * dy (rows of stride ld_db, padding columns zeroed), partial[block] = sum over its items of dy * b;
// with a dropout mask dy is masked and scaled first (the gradient of the dropped branch)
template <typename TB>
__global__ __launch_bounds__(256) void rezero_bwd_kernel(const float *__restrict__ dy, const TB *__restrict__ b, long long ldb,
                                                         const float *__restrict__ r, const unsigned char *__restrict__ mask, float keep,
                                                         DropArgs drop, long long M, int C, TB *__restrict__ db, long long ld_db,
                                                         float *__restrict__ partial) {
    __shared__ float red[4];
    const float rv = *r;
    const int per_row = (int)(ld_db / 4), nv = C / 4;
    const long long items = M * per_row;
    float acc = 0.f;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const long long m = it / per_row;
        const int jv = (int)(it - m * per_row), j = jv * 4;
        float4 o = f4zero();
        if (jv < nv) {
            float4 g = *reinterpret_cast<const float4 *>(dy + m * C + j);
            const float4 bv = io<TB>::ld4(b + m * ldb + j);
            if (mask != nullptr) g = drop4(g, *reinterpret_cast<const uint32_t *>(mask + m * C + j), keep);
            else if (drop.thr) g = drop4_seeded(g, m * C + j, drop);
            acc += g.x * bv.x + g.y * bv.y + g.z * bv.z + g.w * bv.w;
            o = g * rv;
        }
        io<TB>::st4(db + m * ld_db + j, o);
    }
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

constexpr int RZ_BLOCKS = 2048;
int rz_check(const char *who, long long M, int C, long long ldb) {
    if (M <= 0 || C <= 0 || ldb < C) return fail(DSW_EINVAL, "%s: bad sizes M=%lld C=%d ldb=%lld", who, M, C, ldb);
    if (C % 4 || ldb % 4) return fail(DSW_EUNSUPPORTED, "%s: C and the row strides must be multiples of 4 (C=%d ldb=%lld)", who, C, ldb);
    return DSW_OK;
}
unsigned rz_blocks(long long items) { return (unsigned)((items + 255) / 256 < RZ_BLOCKS ? (items + 255) / 256 : RZ_BLOCKS); }

}  // namespace
}  // namespace dsw

extern "C" int dsw_rezero_blocks(long long M, long long ld) { return M > 0 && ld > 0 ? (int)rz_blocks(M * (ld / 4)) : 0; }

static int rezero_fwd_impl(const char *who, const float *x, const void *branch, int b_dtype, long long ldb, const float *r,
                           const unsigned char *mask, float keep, long long M, int C, float *y, void *stream,
                           const DropArgs &drop = DropArgs{0u, 0u, 0u, 1.f}) {
    int rc = rz_check(who, M, C, ldb);
    if (rc) return rc;
    DSW_REQUIRE(x && branch && r && y, "%s: null pointer", who);
    if ((uintptr_t)mask & 3) return fail(DSW_EUNSUPPORTED, "%s: the mask needs a 4-byte aligned base", who);
    DSW_DISPATCH_DTYPE(b_dtype, TB, {
        hipLaunchKernelGGL(rezero_fwd_kernel<TB>, dim3(rz_blocks(M * (C / 4))), dim3(256), 0, as_stream(stream), x, (const TB *)branch, ldb, r,
                           mask, keep, drop, M, C, y);
    });
    return check_launch("rezero_fwd_kernel");
}

static int rezero_bwd_impl(const char *who, const float *dy, const void *branch, int b_dtype, long long ldb, const float *r,
                           const unsigned char *mask, float keep, long long M, int C, void *d_branch, long long ld_db, float *partial,
                           void *stream, const DropArgs &drop = DropArgs{0u, 0u, 0u, 1.f}) {
    int rc = rz_check(who, M, C, ldb);
    if (rc) return rc;
    if (ld_db < C || ld_db % 4) return fail(DSW_EUNSUPPORTED, "%s: gradient row stride %lld (a multiple of 4, >= C)", who, ld_db);
    DSW_REQUIRE(dy && branch && r && d_branch && partial, "%s: null pointer", who);
    if ((uintptr_t)mask & 3) return fail(DSW_EUNSUPPORTED, "%s: the mask needs a 4-byte aligned base", who);
    DSW_DISPATCH_DTYPE(b_dtype, TB, {
        hipLaunchKernelGGL(rezero_bwd_kernel<TB>, dim3(rz_blocks(M * (ld_db / 4))), dim3(256), 0, as_stream(stream), dy, (const TB *)branch,
                           ldb, r, mask, keep, drop, M, C, (TB *)d_branch, ld_db, partial);
    });
    return check_launch("rezero_bwd_kernel");
}

extern "C" int dsw_rezero_fwd(const float *x, const void *branch, int b_dtype, long long ldb, const float *r, long long M, int C, float *y,
                              void *stream) {
    return rezero_fwd_impl("dsw_rezero_fwd", x, branch, b_dtype, ldb, r, nullptr, 1.f, M, C, y, stream);
}

extern "C" int dsw_rezero_bwd(const float *dy, const void *branch, int b_dtype, long long ldb, const float *r, long long M, int C,
                              void *d_branch, long long ld_db, float *partial, void *stream) {
    return rezero_bwd_impl("dsw_rezero_bwd", dy, branch, b_dtype, ldb, r, nullptr, 1.f, M, C, d_branch, ld_db, partial, stream);
}

extern "C" int dsw_rezero_dropout_fwd(const float *x, const void *branch, int b_dtype, long long ldb, const float *r,
                                      const unsigned char *mask, float keep_scale, long long M, int C, float *y, void *stream) {
    DSW_REQUIRE(mask, "dsw_rezero_dropout_fwd: null mask");
    return rezero_fwd_impl("dsw_rezero_dropout_fwd", x, branch, b_dtype, ldb, r, mask, keep_scale, M, C, y, stream);
}

extern "C" int dsw_rezero_dropout_bwd(const float *dy, const void *branch, int b_dtype, long long ldb, const float *r,
                                      const unsigned char *mask, float keep_scale, long long M, int C, void *d_branch, long long ld_db,
                                      float *partial, void *stream) {
    DSW_REQUIRE(mask, "dsw_rezero_dropout_bwd: null mask");
    return rezero_bwd_impl("dsw_rezero_dropout_bwd", dy, branch, b_dtype, ldb, r, mask, keep_scale, M, C, d_branch, ld_db, partial, stream);
}

extern "C" int dsw_rezero_dropout_seeded_fwd(const float *x, const void *branch, int b_dtype, long long ldb, const float *r, float p,
                                             unsigned long long seed, long long M, int C, float *y, void *stream) {
    DropArgs drop;
    const int rc = make_drop("dsw_rezero_dropout_seeded_fwd", p, seed, drop);
    if (rc) return rc;
    return rezero_fwd_impl("dsw_rezero_dropout_seeded_fwd", x, branch, b_dtype, ldb, r, nullptr, 1.f, M, C, y, stream, drop);
}

extern "C" int dsw_rezero_dropout_seeded_bwd(const float *dy, const void *branch, int b_dtype, long long ldb, const float *r, float p,
                                             unsigned long long seed, long long M, int C, void *d_branch, long long ld_db,
                                             float *partial, void *stream) {
    DropArgs drop;
    const int rc = make_drop("dsw_rezero_dropout_seeded_bwd", p, seed, drop);
    if (rc) return rc;
    return rezero_bwd_impl("dsw_rezero_dropout_seeded_bwd", dy, branch, b_dtype, ldb, r, nullptr, 1.f, M, C, d_branch, ld_db, partial,
                           stream, drop);
}

// ---- LayerNorm over channel rows (the nn.LayerNorm layers of msft's ViTBlock, msft_arch.py:152-153, applied
// at :163-165), same lane layout as the RMSNorm above: out = (x - mean) * rstd * w + b -------------------
namespace dsw {
namespace {

template <typename TO>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ b, float eps, long long M, int C,
                                                            TO *__restrict__ out, long long ldo, float *__restrict__ mean,
                                                            float *__restrict__ rstd) {
    const int sub = threadIdx.x & (RN_LPR - 1), rloc = threadIdx.x / RN_LPR;
    const int nv = C / 4, nvo = (int)(ldo / 4);
    float4 wv[RN_MAXV], bv[RN_MAXV];
#pragma unroll
    for (int i = 0; i < RN_MAXV; ++i) {
        const bool live = sub + RN_LPR * i < nv;
        wv[i] = live ? *reinterpret_cast<const float4 *>(w + 4 * (sub + RN_LPR * i)) : f4zero();
        bv[i] = live && b != nullptr ? *reinterpret_cast<const float4 *>(b + 4 * (sub + RN_LPR * i)) : f4zero();
    }
    for (long long r0 = (long long)blockIdx.x * RN_RPB; r0 < M; r0 += (long long)gridDim.x * RN_RPB) {
        const long long row = r0 + rloc;
        const bool ok = row < M;
        float4 v[RN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < RN_MAXV; ++i) {
            v[i] = ok && sub + RN_LPR * i < nv ? *reinterpret_cast<const float4 *>(x + row * C + 4 * (sub + RN_LPR * i)) : f4zero();
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
        const float mu = group16_sum(s) / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < RN_MAXV; ++i) {
            if (sub + RN_LPR * i < nv) {
                v[i] = make_float4(v[i].x - mu, v[i].y - mu, v[i].z - mu, v[i].w - mu);
                ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
            }
        }
        const float rs = rsqrtf(group16_sum(ss) / (float)C + eps);
        if (ok) {
            if (sub == 0 && rstd != nullptr) {
                mean[row] = mu;
                rstd[row] = rs;
            }
#pragma unroll
            for (int i = 0; i < RN_MAXV; ++i) {
                const int j = sub + RN_LPR * i;
                if (j < nvo) {
                    const float4 y = j < nv ? make_float4(v[i].x * rs * wv[i].x + bv[i].x, v[i].y * rs * wv[i].y + bv[i].y,
                                                          v[i].z * rs * wv[i].z + bv[i].z, v[i].w * rs * wv[i].w + bv[i].w)
                                            : f4zero();
                    io<TO>::st4(out + row * ldo + 4 * j, y);
                }
            }
        }
    }
}

// dx = rstd (g - mean(g) - xh mean(g xh)), g = dy w, xh = (x - mean) rstd;  partial row = [sum dy xh | sum dy]
template <typename TG>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TG *__restrict__ dy, long long ldg, const float *__restrict__ x,
                                                            const float *__restrict__ w, const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, long long M, int C, float *__restrict__ dx,
                                                            const float *__restrict__ dx_add, const float *__restrict__ dx_add2,
                                                            long long ld_add2, int cols2, float *__restrict__ partial) {
    __shared__ float4 red[RN_RPB][RN_LPR * RN_MAXV];
    const int sub = threadIdx.x & (RN_LPR - 1), rloc = threadIdx.x / RN_LPR;
    const int nv = C / 4;
    float4 wv[RN_MAXV], dwacc[RN_MAXV], dbacc[RN_MAXV];
#pragma unroll
    for (int i = 0; i < RN_MAXV; ++i) {
        wv[i] = sub + RN_LPR * i < nv ? *reinterpret_cast<const float4 *>(w + 4 * (sub + RN_LPR * i)) : f4zero();
        dwacc[i] = dbacc[i] = f4zero();
    }
    for (long long r0 = (long long)blockIdx.x * RN_RPB; r0 < M; r0 += (long long)gridDim.x * RN_RPB) {
        const long long row = r0 + rloc;
        const bool ok = row < M;
        const float rs = ok ? rstd[row] : 0.f, mu = ok ? mean[row] : 0.f;
        float4 xh[RN_MAXV], g[RN_MAXV], av[RN_MAXV];
        float s1 = 0.f, s2 = 0.f;
        // gradients that reached x by other ways - a map of x's shape, and one that covers the first cols2 columns - are
        // asked for with the row's other operands, ahead of the reductions
#pragma unroll
        for (int i = 0; i < RN_MAXV; ++i) {
            const int c = 4 * (sub + RN_LPR * i);
            const bool live = ok && sub + RN_LPR * i < nv;
            av[i] = live && dx_add != nullptr ? *reinterpret_cast<const float4 *>(dx_add + row * C + c) : f4zero();
            if (live && dx_add2 != nullptr && c < cols2) av[i] = av[i] + *reinterpret_cast<const float4 *>(dx_add2 + row * ld_add2 + c);
        }
#pragma unroll
        for (int i = 0; i < RN_MAXV; ++i) {
            const bool live = ok && sub + RN_LPR * i < nv;
            const float4 xv = live ? *reinterpret_cast<const float4 *>(x + row * C + 4 * (sub + RN_LPR * i)) : make_float4(mu, mu, mu, mu);
            const float4 gy = live ? io<TG>::ld4(dy + row * ldg + 4 * (sub + RN_LPR * i)) : f4zero();
            xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
            g[i] = make_float4(gy.x * wv[i].x, gy.y * wv[i].y, gy.z * wv[i].z, gy.w * wv[i].w);
            s1 += g[i].x + g[i].y + g[i].z + g[i].w;
            s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
            dwacc[i] = dwacc[i] + make_float4(gy.x * xh[i].x, gy.y * xh[i].y, gy.z * xh[i].z, gy.w * xh[i].w);
            dbacc[i] = dbacc[i] + gy;
        }
        s1 = group16_sum(s1) / (float)C;
        s2 = group16_sum(s2) / (float)C;
        if (ok) {
#pragma unroll
            for (int i = 0; i < RN_MAXV; ++i)
                if (sub + RN_LPR * i < nv) {
                    const int c = 4 * (sub + RN_LPR * i);
                    *reinterpret_cast<float4 *>(dx + row * C + c) =
                        av[i] + make_float4((g[i].x - s1 - xh[i].x * s2) * rs, (g[i].y - s1 - xh[i].y * s2) * rs, (g[i].z - s1 - xh[i].z * s2) * rs,
                                            (g[i].w - s1 - xh[i].w * s2) * rs);
                }
        }
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RN_MAXV; ++i) red[rloc][sub + RN_LPR * i] = which ? dbacc[i] : dwacc[i];
        __syncthreads();
        if (threadIdx.x < RN_LPR * RN_MAXV) {
            const int i = threadIdx.x / RN_LPR, s = threadIdx.x % RN_LPR, j = s + RN_LPR * i;
            if (j < nv) {
                float4 a = f4zero();
#pragma unroll
                for (int r = 0; r < RN_RPB; ++r) a = a + red[r][s + RN_LPR * i];
                *reinterpret_cast<float4 *>(partial + (size_t)blockIdx.x * 2 * C + which * C + 4 * j) = a;
            }
        }
    }
}

}  // namespace
}  // namespace dsw

extern "C" int dsw_layernorm_fwd(const float *x, const float *weight, const float *bias, float eps, long long M, int C, void *out,
                                 int out_dtype, long long ldo, float *mean, float *rstd, void *stream) {
    if (M <= 0 || C <= 0 || ldo < C) return fail(DSW_EINVAL, "dsw_layernorm_fwd: bad sizes M=%lld C=%d ldo=%lld", M, C, ldo);
    if (C % 4 || ldo % 4 || C > 4 * RN_LPR * RN_MAXV || ldo > 4 * RN_LPR * RN_MAXV)
        return fail(DSW_EUNSUPPORTED, "dsw_layernorm_fwd: C and the output row stride must be multiples of 4, at most %d (C=%d ldo=%lld)",
                    4 * RN_LPR * RN_MAXV, C, ldo);
    DSW_REQUIRE(x && weight && out && ((mean == nullptr) == (rstd == nullptr)), "dsw_layernorm_fwd: null pointer");
    DSW_DISPATCH_DTYPE(out_dtype, TO, {
        hipLaunchKernelGGL(layernorm_fwd_kernel<TO>, dim3(rn_blocks(M)), dim3(256), 0, as_stream(stream), x, weight, bias, eps, M, C,
                           (TO *)out, ldo, mean, rstd);
    });
    return check_launch("layernorm_fwd_kernel");
}

extern "C" int dsw_layernorm_bwd(const void *dy, int dy_dtype, long long ld_dy, const float *x, const float *weight, const float *mean,
                                 const float *rstd, long long M, int C, float *dx, const float *dx_add, const float *dx_add2,
                                 long long ld_add2, int cols2, float *partial, void *stream) {
    if (M <= 0 || C <= 0 || ld_dy < C) return fail(DSW_EINVAL, "dsw_layernorm_bwd: bad sizes M=%lld C=%d ld_dy=%lld", M, C, ld_dy);
    if (C % 4 || ld_dy % 4 || C > 4 * RN_LPR * RN_MAXV)
        return fail(DSW_EUNSUPPORTED, "dsw_layernorm_bwd: C and the gradient's row stride must be multiples of 4, C at most %d (C=%d ld_dy=%lld)",
                    4 * RN_LPR * RN_MAXV, C, ld_dy);
    if (dx_add2 != nullptr && (cols2 <= 0 || cols2 > C || cols2 % 4 || ld_add2 < cols2 || ld_add2 % 4 || (reinterpret_cast<uintptr_t>(dx_add2) & 15)))
        return fail(DSW_EINVAL, "dsw_layernorm_bwd: the partial-width addend needs 0 < cols2 <= C, cols2 and its row stride multiples of 4, "
                                "16-byte alignment (cols2=%d ld_add2=%lld C=%d)", cols2, ld_add2, C);
    DSW_REQUIRE(dy && x && weight && mean && rstd && dx && partial, "dsw_layernorm_bwd: null pointer");
    DSW_DISPATCH_DTYPE(dy_dtype, TG, {
        hipLaunchKernelGGL(layernorm_bwd_kernel<TG>, dim3(rn_blocks(M)), dim3(256), 0, as_stream(stream), (const TG *)dy, ld_dy, x, weight,
                           mean, rstd, M, C, dx, dx_add, dx_add2, ld_add2, cols2, partial);
    });
    return check_launch("layernorm_bwd_kernel");
}
