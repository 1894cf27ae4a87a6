// Shared helpers for the gfx950 kernels of libdswin_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/dswin_hip.h"

namespace dsw {

// ---- error plumbing: one message per host thread, nothing shared ---------------------------
char *tls_error_buffer();
inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tls_error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}
inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DSW_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return DSW_OK;
}
#define DSW_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return ::dsw::fail(DSW_EINVAL, __VA_ARGS__); \
    } while (0)

// ---- bf16 <-> f32 -----------------------------------------------------------------------------
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even through the hardware converter (v_cvt_pk_bf16_f32); NaN stays NaN
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    bf16x2_hw v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

template <typename T> struct io;
template <> struct io<float> {
    static constexpr int dtype = DSW_F32;
    __device__ static __forceinline__ float ld(const float *p) { return *p; }
    __device__ static __forceinline__ void st(float *p, float v) { *p = v; }
    // 4 consecutive elements, 16-byte aligned
    __device__ static __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    __device__ static __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
};
template <> struct io<bf16_t> {
    static constexpr int dtype = DSW_BF16;
    __device__ static __forceinline__ float ld(const bf16_t *p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t *p, float v) { *p = f32_to_bf16(v); }
    // 4 consecutive elements, 8-byte aligned
    __device__ static __forceinline__ float4 ld4(const bf16_t *p) {
        uint2 u = *reinterpret_cast<const uint2 *>(p);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                           __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
    __device__ static __forceinline__ void st4(bf16_t *p, float4 v) {
        uint2 u;
        u.x = pack_bf16(v.x, v.y);
        u.y = pack_bf16(v.z, v.w);
        *reinterpret_cast<uint2 *>(p) = u;
    }
};

__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 operator/(float4 a, float s) { return make_float4(a.x / s, a.y / s, a.z / s, a.w / s); }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// silu(g) * c of the gated MLP (DSwinT_arch.py:69), shared by the stand-alone kernel and the projection epilogue so
// that both round alike
// (the reciprocal as the hardware's v_rcp_f32, 1 ulp: the IEEE division costs ten instructions per element, which
// the projection epilogues - vector-bound next to their MFMAs - cannot hide)
__device__ __forceinline__ float gated_silu_value(float g, float c) { return g * __builtin_amdgcn_rcpf(1.f + __expf(-g)) * c; }

// GELU in its erf form (nn.GELU() of msft's MLP, msft_arch.py:25), shared by the stand-alone kernels and the projection
// epilogues so that both round alike.
// fp32 tensors: the library's erff.  bf16 tensors: erf by the rational-times-Gaussian form 7.1.26 of Abramowitz and
// Stegun (|error| <= 1.5e-7, far inside a bf16 rounding) - a dozen vector instructions without a branch, where erff's
// piecewise evaluation made these element-wise kernels instruction-bound; the normal distribution function is taken
// from the complementary side for negative arguments, so nothing cancels.
template <typename T>
__device__ __forceinline__ void gelu_terms(float x, float &cdf, float &gauss) {
    if constexpr (sizeof(T) == 4) {
        cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
        gauss = __expf(-0.5f * x * x);
    } else {
        const float u = fabsf(x) * 0.70710678118654752f;
        const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * u);
        gauss = __expf(-u * u);
        const float p = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
        const float half_erfc = 0.5f * p * gauss;
        cdf = x < 0.f ? half_erfc : 1.f - half_erfc;
    }
}
template <typename T>
__device__ __forceinline__ float gelu_value(float x) {
    float cdf, gauss;
    gelu_terms<T>(x, cdf, gauss);
    return x * cdf;
}
template <typename T>
__device__ __forceinline__ float gelu_slope(float x) {
    float cdf, gauss;
    gelu_terms<T>(x, cdf, gauss);
    return cdf + x * 0.3989422804014327f * gauss;
}

// ---- wave / block reductions (wave = 64 lanes) ----------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// fixed-order block reduction (the same tree on every launch => bitwise reproducible);
// red must hold blockDim.x/64 floats; every thread gets the result.
__device__ __forceinline__ float block_sum(float v, float *red) {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float *red) {
    v = wave_max(v);
    const int wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wid] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// dispatch a functor template on a runtime dtype code
#define DSW_DISPATCH_DTYPE(code, T, ...)                                           \
    do {                                                                           \
        if ((code) == DSW_F32) { using T = float; __VA_ARGS__; }                   \
        else if ((code) == DSW_BF16) { using T = ::dsw::bf16_t; __VA_ARGS__; }     \
        else return ::dsw::fail(DSW_EUNSUPPORTED, "dtype code %d", (int)(code));   \
    } while (0)

// dst[c] = sum_r src[r][c] for a row-major [rows][cols] fp32 matrix of per-block partial sums, in a fixed
// order: a block of 32 * SL threads takes 32 consecutive columns, its SL thread slices every SL-th row (four rows
// of a slice in flight at a time, each with its own running sum: the loop is bound by the latency of its loads),
// and the slices are added in slice order.  The body of the small reduction kernels at the end of the backward
// passes.
template <int SL>
__device__ __forceinline__ bool column_sum_32(const float *__restrict__ src, int rows, int cols, float (*red)[33], int &col, float &total) {
    const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
    col = blockIdx.x * 32 + e;
    float acc = 0.f;
    if (col < cols) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int r = sl;
        for (; r + 3 * SL < rows; r += 4 * SL) {
            a0 += src[(size_t)r * cols + col];
            a1 += src[(size_t)(r + SL) * cols + col];
            a2 += src[(size_t)(r + 2 * SL) * cols + col];
            a3 += src[(size_t)(r + 3 * SL) * cols + col];
        }
        for (; r < rows; r += SL) a0 += src[(size_t)r * cols + col];
        acc = (a0 + a1) + (a2 + a3);
    }
    red[sl][e] = acc;
    __syncthreads();
    if (sl != 0 || col >= cols) return false;
    total = 0.f;
#pragma unroll
    for (int k = 0; k < SL; ++k) total += red[k][e];
    return true;
}

// Blocks that share blockIdx % 8 run on one XCD and share its L2.  Maps the hardware block index to a logical one
// such that every XCD works through one contiguous range of logical indices: neighbours in the logical order (the
// heads of a window, the rows of a window, vertically adjacent pixel tiles of an image) then meet in one L2 instead
// of fetching the lines they share once per XCD.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// Dropout without a mask tensor: a keep decision is a function of a 64-bit seed and a position, which the forward
// and the backward kernels both evaluate (and which a CPU check can evaluate too).  One 32-bit hash serves a PAIR
// of adjacent elements: x = (pair index + seed_lo) ^ seed_hi through the murmur3 finaliser; the first element
// takes bits 0-15 of x, the second bits 16-31, and an element is kept when its 16 bits are >= thr =
// round(p * 65536).  Kept values are scaled by 1 / (1 - p).  How positions map to pair indices is the caller's
// (attention: see attn_common.hpp; element-wise: element index / 2).
struct DropArgs {
    uint32_t thr;      // 0: no dropout
    uint32_t seed_lo, seed_hi;
    float inv_keep;    // 1 / (1 - p)
};
__device__ __forceinline__ uint32_t drop_hash(uint32_t pair_idx, uint32_t seed_lo, uint32_t seed_hi) {
    uint32_t x = (pair_idx + seed_lo) ^ seed_hi;  // then the murmur3 finaliser
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}

// inverted dropout on 4 consecutive elements with the decisions taken from the seed (DropArgs): e0 = index of the first of the 4 elements in the
// dense [M][C] matrix (a multiple of 4); elements 2i, 2i+1 share pair index i (mod 2^32)
__device__ __forceinline__ float4 drop4_seeded(float4 v, long long e0, const DropArgs &d) {
    const uint32_t pair = (uint32_t)(e0 >> 1);
    const uint32_t h0 = drop_hash(pair, d.seed_lo, d.seed_hi), h1 = drop_hash(pair + 1u, d.seed_lo, d.seed_hi);
    return make_float4((h0 & 0xffffu) >= d.thr ? v.x * d.inv_keep : 0.f, (h0 >> 16) >= d.thr ? v.y * d.inv_keep : 0.f,
                       (h1 & 0xffffu) >= d.thr ? v.z * d.inv_keep : 0.f, (h1 >> 16) >= d.thr ? v.w * d.inv_keep : 0.f);
}

int make_drop(const char *who, float p, unsigned long long seed, DropArgs &d);  // C-ABI arguments -> kernel form

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace dsw
