// Adam step of the reference's trainer (toolkit/init.py:133: torch.optim.Adam(model.parameters(), lr, betas=(0.9, 0.99)),
// stepped at toolkit/model_helper2.py:125) over all parameter tensors in one launch per 64 tensors.
// torch's multi-tensor Adam issues eight elementwise passes over the lists (lerp, mul, addcmul, sqrt, div, add,
// addcdiv, plus the step counters) with the intermediates written to memory; here one pass reads p, g, m, v and
// writes p, m, v: 4 reads + 3 writes of 4 bytes per element, the minimum for the update.
//
// Arithmetic, per element, fp32, in the order and rounding of torch's multi-tensor kernels on this GPU (each of their
// passes was compared bit for bit with the candidate forms):
//   m <- fma(w1, g - m, m)                      w1 = 1 - beta1                  (lerp)
//   v <- v * beta2                                                               (mul)
//   v <- fma(w2, g * g, v)                      w2 = 1 - beta2                  (addcmul)
//   d <- sqrt(v) / bias2_sqrt + eps             bias2_sqrt = sqrt(1 - beta2^t)  (sqrt, div, add)
//   p <- fma(step_size, m / d, p)               step_size = -lr / (1 - beta1^t) (addcdiv)
// Square root and division are correctly rounded; every other line is one rounding per operation shown.
#include "common.hpp"

namespace dsw {
namespace {

constexpr int ADAM_MAXT = 64;                  // tensors per launch (the argument block stays under 4 KB)
constexpr int ADAM_THREADS = 256;
constexpr int ADAM_CHUNK = ADAM_THREADS * 8;   // elements per workgroup

struct AdamBatch {
    float *p[ADAM_MAXT];
    const float *g[ADAM_MAXT];
    float *m[ADAM_MAXT];
    float *v[ADAM_MAXT];
    long long numel[ADAM_MAXT];
    int first_block[ADAM_MAXT + 1];
    int n;
};

struct AdamScalars {
    float w1, beta2, w2, bias2_sqrt, eps, step_size;
};

__device__ __forceinline__ void adam_element(float &p, float g, float &m, float &v, const AdamScalars &s) {
    m = __fmaf_rn(s.w1, __fsub_rn(g, m), m);
    v = __fmul_rn(v, s.beta2);
    v = __fmaf_rn(s.w2, __fmul_rn(g, g), v);
    // sqrtf and `/` are the correctly rounded forms (the __fsqrt_rn intrinsic of this toolchain is the approximate one)
    const float d = __fadd_rn(sqrtf(v) / s.bias2_sqrt, s.eps);
    p = __fmaf_rn(s.step_size, m / d, p);
}

__global__ __launch_bounds__(ADAM_THREADS) void adam_update_kernel(const AdamBatch batch, const AdamScalars s) {
    int t = 0, hi = batch.n;   // first_block[t] <= blockIdx.x < first_block[t + 1], uniform over the workgroup
    while (hi - t > 1) {
        const int mid = (t + hi) >> 1;
        if ((int)blockIdx.x >= batch.first_block[mid]) t = mid;
        else hi = mid;
    }
    float *__restrict__ p = batch.p[t];
    const float *__restrict__ g = batch.g[t];
    float *__restrict__ m = batch.m[t];
    float *__restrict__ v = batch.v[t];
    const long long numel = batch.numel[t];
    const long long e0 = (long long)((int)blockIdx.x - batch.first_block[t]) * ADAM_CHUNK;
    const long long e1 = e0 + ADAM_CHUNK < numel ? e0 + ADAM_CHUNK : numel;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    long long scalar_from = e0;
    if (vec) {
        const long long v1 = e0 + ((e1 - e0) & ~3ll);
        for (long long e = e0 + 4 * threadIdx.x; e < v1; e += 4 * ADAM_THREADS) {
            float4 pp = *reinterpret_cast<const float4 *>(p + e), mm = *reinterpret_cast<const float4 *>(m + e),
                   vv = *reinterpret_cast<const float4 *>(v + e);
            const float4 gg = *reinterpret_cast<const float4 *>(g + e);
            adam_element(pp.x, gg.x, mm.x, vv.x, s);
            adam_element(pp.y, gg.y, mm.y, vv.y, s);
            adam_element(pp.z, gg.z, mm.z, vv.z, s);
            adam_element(pp.w, gg.w, mm.w, vv.w, s);
            *reinterpret_cast<float4 *>(p + e) = pp;
            *reinterpret_cast<float4 *>(m + e) = mm;
            *reinterpret_cast<float4 *>(v + e) = vv;
        }
        scalar_from = v1;
    }
    for (long long e = scalar_from + threadIdx.x; e < e1; e += ADAM_THREADS) {
        float pp = p[e], mm = m[e], vv = v[e];
        adam_element(pp, g[e], mm, vv, s);
        p[e] = pp;
        m[e] = mm;
        v[e] = vv;
    }
}

}  // namespace
}  // namespace dsw

using namespace dsw;

extern "C" int dsw_adam_update(int n, void *const *param, const void *const *grad, void *const *exp_avg, void *const *exp_avg_sq,
                               const long long *numel, float one_minus_beta1, float beta2, float one_minus_beta2,
                               float bias2_sqrt, float eps, float step_size, void *stream) {
    if (n < 0) return fail(DSW_EINVAL, "dsw_adam_update: n=%d", n);
    if (n == 0) return DSW_OK;
    DSW_REQUIRE(param && grad && exp_avg && exp_avg_sq && numel, "dsw_adam_update: null table");
    if (!(bias2_sqrt > 0.f)) return fail(DSW_EINVAL, "dsw_adam_update: bias2_sqrt=%g (must be positive)", (double)bias2_sqrt);
    for (int i = 0; i < n; ++i) {
        if (numel[i] < 0) return fail(DSW_EINVAL, "dsw_adam_update: tensor %d has %lld elements", i, numel[i]);
        if (numel[i] > 0 && (!param[i] || !grad[i] || !exp_avg[i] || !exp_avg_sq[i]))
            return fail(DSW_EINVAL, "dsw_adam_update: tensor %d: null pointer", i);
        if (((uintptr_t)param[i] | (uintptr_t)grad[i] | (uintptr_t)exp_avg[i] | (uintptr_t)exp_avg_sq[i]) & 3)
            return fail(DSW_EINVAL, "dsw_adam_update: tensor %d is not aligned to 4 bytes", i);
    }
    const AdamScalars s{one_minus_beta1, beta2, one_minus_beta2, bias2_sqrt, eps, step_size};
    hipStream_t st = as_stream(stream);
    int i = 0;
    while (i < n) {
        AdamBatch b;
        b.n = 0;
        long long blocks = 0;
        while (i < n && b.n < ADAM_MAXT) {
            const long long nb = (numel[i] + ADAM_CHUNK - 1) / ADAM_CHUNK;
            if (nb == 0) {
                ++i;
                continue;
            }
            if (blocks + nb > 0x7fffffffLL) {
                if (b.n == 0) return fail(DSW_EUNSUPPORTED, "dsw_adam_update: tensor %d has too many elements (%lld)", i, numel[i]);
                break;
            }
            b.p[b.n] = static_cast<float *>(param[i]);
            b.g[b.n] = static_cast<const float *>(grad[i]);
            b.m[b.n] = static_cast<float *>(exp_avg[i]);
            b.v[b.n] = static_cast<float *>(exp_avg_sq[i]);
            b.numel[b.n] = numel[i];
            b.first_block[b.n] = (int)blocks;
            blocks += nb;
            ++b.n;
            ++i;
        }
        if (b.n == 0) break;
        b.first_block[b.n] = (int)blocks;
        hipLaunchKernelGGL(adam_update_kernel, dim3((unsigned)blocks), dim3(ADAM_THREADS), 0, st, b, s);
        const int rc = check_launch("adam_update_kernel");
        if (rc) return rc;
    }
    return DSW_OK;
}
