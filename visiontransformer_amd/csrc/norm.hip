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
                                                          float *__restrict__ dx, float *__restrict__ partial) {
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
        float4 xh[RN_MAXV], g[RN_MAXV];
        float dot = 0.f;
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
                if (sub + RN_LPR * i < nv)
                    *reinterpret_cast<float4 *>(dx + row * C + 4 * (sub + RN_LPR * i)) = (g[i] - xh[i] * dot) * rs;
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
                               long long M, int C, float *dx, float *partial, void *stream) {
    if (M <= 0 || C <= 0 || ld_dy < C) return fail(DSW_EINVAL, "dsw_rmsnorm_bwd: bad sizes M=%lld C=%d ld_dy=%lld", M, C, ld_dy);
    if (C % 4 || ld_dy % 4 || C > 4 * RN_LPR * RN_MAXV)
        return fail(DSW_EUNSUPPORTED, "dsw_rmsnorm_bwd: C and the gradient's row stride must be multiples of 4, C at most %d (C=%d ld_dy=%lld)",
                    4 * RN_LPR * RN_MAXV, C, ld_dy);
    DSW_REQUIRE(dy && x && weight && rstd && dx && partial, "dsw_rmsnorm_bwd: null pointer");
    DSW_DISPATCH_DTYPE(dy_dtype, TG, {
        hipLaunchKernelGGL(rmsnorm_bwd_kernel<TG>, dim3(rn_blocks(M)), dim3(256), 0, as_stream(stream), (const TG *)dy, ld_dy, x, weight, rstd,
                           M, C, dx, partial);
    });
    return check_launch("rmsnorm_bwd_kernel");
}
