// Exponential moving average of a model's parameters (toolkit/param_ema.py:9-17 of the reference:
// a Python loop of three elementwise launches per parameter) as one launch per 128 parameters.
//   ema[i] = ema[i] * beta + (1 - beta) * cur[i]
// with the reference's rounding: the two products are rounded to the parameter dtype separately and then
// added (no fused multiply-add), so the result is bit-identical to the three torch ops.
// The tensors of a launch travel as kernel arguments (pointers, sizes and the first block of each), so
// the library needs no device-side table and no allocation.  Pure streaming: 2 reads + 1 write per element.
#include "common.hpp"

namespace dsw {
namespace {

constexpr int EMA_MAXT = 128;               // tensors per launch (the argument block stays under 4 KB)
constexpr int EMA_THREADS = 256;
constexpr int EMA_CHUNK = EMA_THREADS * 8;   // elements per workgroup

struct EmaBatch {
    void *ema[EMA_MAXT];
    const void *cur[EMA_MAXT];
    long long numel[EMA_MAXT];
    int first_block[EMA_MAXT + 1];
    int n;
};

template <typename T> __device__ __forceinline__ float to_dtype(float v);
template <> __device__ __forceinline__ float to_dtype<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_dtype<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

template <typename T> __device__ __forceinline__ float ema_value(float old, float nw, float beta, float omb) {
    return __fadd_rn(to_dtype<T>(__fmul_rn(old, beta)), to_dtype<T>(__fmul_rn(omb, nw)));
}

template <typename T>
__global__ __launch_bounds__(EMA_THREADS) void ema_update_kernel(const EmaBatch batch, float beta, float omb) {
    // which tensor this workgroup belongs to (uniform over the workgroup)
    int t = 0, hi = batch.n;   // first_block[t] <= blockIdx.x < first_block[t + 1]
    while (hi - t > 1) {
        const int mid = (t + hi) >> 1;
        if ((int)blockIdx.x >= batch.first_block[mid]) t = mid;
        else hi = mid;
    }
    T *__restrict__ ema = static_cast<T *>(batch.ema[t]);
    const T *__restrict__ cur = static_cast<const T *>(batch.cur[t]);
    const long long numel = batch.numel[t];
    const long long e0 = (long long)((int)blockIdx.x - batch.first_block[t]) * EMA_CHUNK;
    const long long e1 = e0 + EMA_CHUNK < numel ? e0 + EMA_CHUNK : numel;
    const bool vec = ((reinterpret_cast<uintptr_t>(ema) | reinterpret_cast<uintptr_t>(cur)) & (4 * sizeof(T) - 1)) == 0;
    if (vec) {
        const long long v1 = e0 + ((e1 - e0) & ~3ll);
        for (long long e = e0 + 4 * threadIdx.x; e < v1; e += 4 * EMA_THREADS) {
            const float4 o = io<T>::ld4(ema + e), c = io<T>::ld4(cur + e);
            io<T>::st4(ema + e, make_float4(ema_value<T>(o.x, c.x, beta, omb), ema_value<T>(o.y, c.y, beta, omb),
                                            ema_value<T>(o.z, c.z, beta, omb), ema_value<T>(o.w, c.w, beta, omb)));
        }
        for (long long e = v1 + threadIdx.x; e < e1; e += EMA_THREADS)
            io<T>::st(ema + e, ema_value<T>(io<T>::ld(ema + e), io<T>::ld(cur + e), beta, omb));
    } else {
        for (long long e = e0 + threadIdx.x; e < e1; e += EMA_THREADS)
            io<T>::st(ema + e, ema_value<T>(io<T>::ld(ema + e), io<T>::ld(cur + e), beta, omb));
    }
}

}  // namespace
}  // namespace dsw

using namespace dsw;

extern "C" int dsw_ema_update(int n, void *const *ema, const void *const *cur, const long long *numel, int dtype, float beta,
                              float one_minus_beta, void *stream) {
    if (n < 0) return fail(DSW_EINVAL, "dsw_ema_update: n=%d", n);
    if (n == 0) return DSW_OK;
    DSW_REQUIRE(ema && cur && numel, "dsw_ema_update: null table");
    const size_t es = dtype == DSW_F32 ? 4 : 2;
    for (int i = 0; i < n; ++i) {
        if (numel[i] < 0) return fail(DSW_EINVAL, "dsw_ema_update: tensor %d has %lld elements", i, numel[i]);
        if (numel[i] > 0 && (!ema[i] || !cur[i])) return fail(DSW_EINVAL, "dsw_ema_update: tensor %d: null pointer", i);
        if (((uintptr_t)ema[i] | (uintptr_t)cur[i]) & (es - 1)) return fail(DSW_EINVAL, "dsw_ema_update: tensor %d is not aligned to its element size", i);
    }
    hipStream_t st = as_stream(stream);
    int i = 0;
    while (i < n) {
        EmaBatch b;
        b.n = 0;
        long long blocks = 0;
        while (i < n && b.n < EMA_MAXT) {
            const long long nb = (numel[i] + EMA_CHUNK - 1) / EMA_CHUNK;
            if (nb == 0) {
                ++i;
                continue;
            }
            if (blocks + nb > 0x7fffffffLL) {
                if (b.n == 0) return fail(DSW_EUNSUPPORTED, "dsw_ema_update: tensor %d has too many elements (%lld)", i, numel[i]);
                break;
            }
            b.ema[b.n] = ema[i];
            b.cur[b.n] = cur[i];
            b.numel[b.n] = numel[i];
            b.first_block[b.n] = (int)blocks;
            blocks += nb;
            ++b.n;
            ++i;
        }
        if (b.n == 0) break;
        b.first_block[b.n] = (int)blocks;
        DSW_DISPATCH_DTYPE(dtype, T, {
            hipLaunchKernelGGL(ema_update_kernel<T>, dim3((unsigned)blocks), dim3(EMA_THREADS), 0, st, b, beta, one_minus_beta);
        });
        const int rc = check_launch("ema_update_kernel");
        if (rc) return rc;
    }
    return DSW_OK;
}
