// The gated MLP of DSWT's transformer block with the RMSNorm in front of it and the ReZero residual around it
// (DSwinT_arch.py:47-69 with :345 and :357) as ONE kernel, for inference:
//
//   y = x + r * ( fc2( silu(fc1_gate(n)) * fc1_content(n) ) + b2 ),   n = rms_norm(x) * g
//
// The hidden activations ([tokens, 2 * hidden] pre-activations, [tokens, hidden] gated values) never exist in memory:
// x is read once and y written once, 2 * C * 4 bytes per token.
//
// Work split: a workgroup owns 128 rows of x, 32 per wave.  A wave holds its rows as the B operands of
// H^T = W1 . X^T (x rows on the MFMA lanes: lane (n, g) keeps columns 96 g .. 96 g + 95 of row n; the rows come from
// memory as whole 128-byte lines, 8 threads per row, which also sum the squares for 1 / rms, and cross LDS once as bf16).
// The hidden dimension is walked in chunks of 32 units whose weights go through LDS: 24 MFMAs give the chunk's gate and
// content tiles [32 hidden x 32 rows]; gated, and rounded to bf16, a lane's accumulators ARE the B operand of
// Out^T += W2[:, chunk] . Z^T (the hidden units of W2 are laid out in the order the accumulators hold them), so the
// gated values never leave the registers: 12 more MFMAs into six persistent [32 out x 32 rows] tiles.  The result
// leaves through a per-wave LDS staging area as whole 128-byte row segments, where the fp32 residual (x itself) is
// read back in the same shape.
#include "attn_common.hpp"

namespace dsw {
namespace {

constexpr int ML_BM = 128;       // rows of x per workgroup
constexpr int ML_THREADS = 256;  // 4 waves, 32 rows each
constexpr int ML_K = 192;        // x row as MFMA operands: C <= 192, zero beyond C
constexpr int ML_LS1 = ML_K + 8; // LDS row stride (elements) of the fc1 chunk: 16-byte reads of 8 rows on distinct banks
constexpr int ML_HC = 32;        // hidden units per chunk
constexpr int ML_LS2 = ML_HC + 8;
constexpr int ML_N = 192;        // output columns as MFMA tiles: C <= 192
constexpr int ML_SROW = 144;     // bytes per row of a wave's output staging (32 floats + padding)
constexpr int ML_MAXCHUNKS = 16;

// ML_STAMPS (diagnostic builds only, tools/mlp_stamps.py): thread 0 of every workgroup records the shader clock at
// points of its life into a device array that dsw_debug_read_mlp_stamps copies out
#ifdef ML_STAMPS
__device__ unsigned long long g_mlp_stamps[16 * 8192];
#define ML_STAMP(i)                                                                                              \
    do {                                                                                                         \
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_mlp_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define ML_STAMP(i) ((void)0)
#endif

// a thread's nine 16-byte pieces of a weight chunk pair on their way from L2 to LDS (named members, not arrays: an
// array that lives across the chunk loop ends up in scratch memory)
struct MlPieces {
    uint4 a0, a1, a2, a3, a4, a5, b0, b1, b2;
};
__device__ __forceinline__ void ml_load_w1(MlPieces &r, const bf16_t *__restrict__ w1c, int c, int tid) {
    const uint4 *src = reinterpret_cast<const uint4 *>(w1c + (size_t)c * 64 * ML_K) + tid;
    r.a0 = src[0], r.a1 = src[256], r.a2 = src[512], r.a3 = src[768], r.a4 = src[1024], r.a5 = src[1280];
}
__device__ __forceinline__ void ml_load_w2(MlPieces &r, const bf16_t *__restrict__ w2c, int c, int tid) {
    const uint4 *src = reinterpret_cast<const uint4 *>(w2c + (size_t)c * ML_N * ML_HC) + tid;
    r.b0 = src[0], r.b1 = src[256], r.b2 = src[512];
}
__device__ __forceinline__ void ml_put_w1(bf16_t *W1s, int p, uint4 v) {
    const int row = p / 24, c8 = p - row * 24;
    *reinterpret_cast<uint4 *>(W1s + row * ML_LS1 + c8 * 8) = v;
}
__device__ __forceinline__ void ml_put_w2(bf16_t *W2s, int p, uint4 v) { *reinterpret_cast<uint4 *>(W2s + (p >> 2) * ML_LS2 + (p & 3) * 8) = v; }
__device__ __forceinline__ void ml_write_w(const MlPieces &r, bf16_t *W1s, bf16_t *W2s, int tid) {
    ml_put_w1(W1s, tid, r.a0), ml_put_w1(W1s, tid + 256, r.a1), ml_put_w1(W1s, tid + 512, r.a2);
    ml_put_w1(W1s, tid + 768, r.a3), ml_put_w1(W1s, tid + 1024, r.a4), ml_put_w1(W1s, tid + 1280, r.a5);
    ml_put_w2(W2s, tid, r.b0), ml_put_w2(W2s, tid + 256, r.b1), ml_put_w2(W2s, tid + 512, r.b2);
}

// MFMA weight operands of two k steps: gate and content fragments of fc1 (k steps 2 g, 2 g + 1) / both k steps of the
// output tiles 2 u and 2 u + 1 of fc2
__device__ __forceinline__ void ml_read_w1(const bf16_t *w1g, const bf16_t *w1k, int g, bf16x8 (&d)[4]) {
    d[0] = as_bf16x8(*reinterpret_cast<const uint4 *>(w1g + (2 * g) * 8));
    d[1] = as_bf16x8(*reinterpret_cast<const uint4 *>(w1k + (2 * g) * 8));
    d[2] = as_bf16x8(*reinterpret_cast<const uint4 *>(w1g + (2 * g + 1) * 8));
    d[3] = as_bf16x8(*reinterpret_cast<const uint4 *>(w1k + (2 * g + 1) * 8));
}
__device__ __forceinline__ void ml_read_w2(const bf16_t *w2r, int u, bf16x8 (&d)[4]) {
    d[0] = as_bf16x8(*reinterpret_cast<const uint4 *>(w2r + (2 * u) * 32 * ML_LS2));
    d[1] = as_bf16x8(*reinterpret_cast<const uint4 *>(w2r + (2 * u) * 32 * ML_LS2 + 16));
    d[2] = as_bf16x8(*reinterpret_cast<const uint4 *>(w2r + (2 * u + 1) * 32 * ML_LS2));
    d[3] = as_bf16x8(*reinterpret_cast<const uint4 *>(w2r + (2 * u + 1) * 32 * ML_LS2 + 16));
}

__global__ __launch_bounds__(ML_THREADS, 2) void mlp_gated_residual_kernel(
    const float *__restrict__ x, long long ldx, int M, int C, float norm_eps, const bf16_t *__restrict__ w1c,
    const float *__restrict__ b1c, const bf16_t *__restrict__ w2c, const float *__restrict__ b2, int nchunk,
    const float *__restrict__ rscale, float *__restrict__ y, long long ldy) {
    // w1c [nchunk][64][192] bf16: rows 0..31 of a chunk the gate rows of its 32 hidden units, rows 32..63 their
    //     content rows, the norm's weight multiplied into the columns, zero beyond the matrix
    // b1c [nchunk][64] fp32, the same order; w2c [nchunk][192][32] bf16: W2[out][hidden of the chunk], every 16 hidden
    //     units in the order 0-3, 8-11, 4-7, 12-15; b2 [192] fp32, zero beyond C
    extern __shared__ __align__(16) unsigned char smem[];
    float *b1s = reinterpret_cast<float *>(smem);
    float *b2s = b1s + nchunk * 64;
    float *rstdl = b2s + ML_N;                                  // 1 / rms of the block's rows
    bf16_t *W1s = reinterpret_cast<bf16_t *>(rstdl + ML_BM);
    bf16_t *W2s = W1s + 64 * ML_LS1;
    unsigned char *stg_all = reinterpret_cast<unsigned char *>(W2s + ML_N * ML_LS2);
    bf16_t *As = W1s;   // the block's rows of x as bf16 on their way into the operand registers: [128][ML_LS1], over the
                        // weight chunk and the output staging, which are not in use yet

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * ML_BM;
    ML_STAMP(0);

    // weight chunks: global -> registers -> LDS, 16-byte pieces
    MlPieces wp;
    // The block's rows of x: 8 threads per row, 8 columns each per 64-column group (whole 128-byte lines per row and
    // instruction), summed squares over the 8 lanes of a row, then through LDS as bf16 into the operand registers.
    bf16x8 afr[12];
    float rstd = 1.f;
    {
        const int srow = tid >> 3, scol = (tid & 7) * 8;
        float4 xv[4][3][2];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = m0 + p * 32 + srow;
            const float *xr = x + (size_t)(m < M ? m : 0) * ldx + scol;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    xv[p][j][h] = (m < M && 64 * j + scol + 4 * h + 4 <= C) ? *reinterpret_cast<const float4 *>(xr + 64 * j + 4 * h)
                                                                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        ml_load_w1(wp, w1c, 0, tid);
        ml_load_w2(wp, w2c, 0, tid);
        for (int i = tid; i < nchunk * 64; i += ML_THREADS) b1s[i] = b1c[i];
        if (tid < ML_N) b2s[tid] = b2[tid];
        ML_STAMP(1);   // loads issued
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 v = xv[p][j][h];
                    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                }
            ss += __shfl_xor(ss, 1, 64);
            ss += __shfl_xor(ss, 2, 64);
            ss += __shfl_xor(ss, 4, 64);
            if ((tid & 7) == 0) rstdl[p * 32 + srow] = norm_eps >= 0.f ? rsqrtf(ss / (float)C + norm_eps) : 1.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float4 a = xv[p][j][0], b = xv[p][j][1];
                *reinterpret_cast<uint4 *>(As + (p * 32 + srow) * ML_LS1 + 64 * j + scol) =
                    make_uint4(pack_bf16(a.x, a.y), pack_bf16(a.z, a.w), pack_bf16(b.x, b.y), pack_bf16(b.z, b.w));
            }
        }
        lds_barrier();
        const bf16_t *arow = As + (wave * 32 + l31) * ML_LS1 + 96 * hh;   // lane (n, g) keeps columns 96 g .. 96 g + 95 of row n
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) afr[ks] = as_bf16x8(*reinterpret_cast<const uint4 *>(arow + ks * 8));
        rstd = rstdl[wave * 32 + l31];
        lds_barrier();   // every wave holds its operands: the chunk's weights may take the place of the rows
        ml_write_w(wp, W1s, W2s, tid);
    }
    lds_barrier();
    ML_STAMP(2);       // x and the first chunk arrived and staged

    f32x16 out[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[t][r] = 0.f;

    const bf16_t *w1g = W1s + l31 * ML_LS1 + 96 * hh, *w1k = w1g + 32 * ML_LS1;
    const bf16_t *w2r = W2s + l31 * ML_LS2 + 8 * hh;
    for (int c = 0; c < nchunk; ++c) {
        const bool more = c + 1 < nchunk;
        ml_load_w1(wp, w1c, more ? c + 1 : c, tid);   // (unconditional: a conditional load would send the staging registers through scratch memory)
        f32x16 G, Cn;
#pragma unroll
        for (int r = 0; r < 16; ++r) G[r] = 0.f, Cn[r] = 0.f;
        // weight operands are read from LDS one group of four MFMAs ahead (left to itself the compiler issues each read
        // right in front of its MFMA)
        bf16x8 wf[2][4];
        ml_read_w1(w1g, w1k, 0, wf[0]);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (g + 1 < 6) ml_read_w1(w1g, w1k, g + 1, wf[(g + 1) & 1]);
            else ml_read_w2(w2r, 0, wf[0]);
            G = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[g & 1][0], afr[2 * g], G, 0, 0, 0);
            Cn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[g & 1][1], afr[2 * g], Cn, 0, 0, 0);
            G = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[g & 1][2], afr[2 * g + 1], G, 0, 0, 0);
            Cn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[g & 1][3], afr[2 * g + 1], Cn, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c == 0) ML_STAMP(3);
        // the gated product of the fp32 pre-activations (the separate kernels round gate and content to bf16 first, as
        // they store them), rounded to bf16 as the second projection's operand (accumulator register i of lane half g
        // is hidden unit 8 (i / 4) + 4 g + i % 4 of the chunk)
        uint32_t zp[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bg = *reinterpret_cast<const float4 *>(b1s + c * 64 + 8 * q + 4 * hh);
            const float4 bk = *reinterpret_cast<const float4 *>(b1s + c * 64 + 32 + 8 * q + 4 * hh);
            const float bgv[4] = {bg.x, bg.y, bg.z, bg.w}, bkv[4] = {bk.x, bk.y, bk.z, bk.w};
            float z[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                z[j] = gated_silu_value(G[4 * q + j] * rstd + bgv[j], Cn[4 * q + j] * rstd + bkv[j]);
            }
            zp[2 * q] = pack_bf16(z[0], z[1]);
            zp[2 * q + 1] = pack_bf16(z[2], z[3]);
        }
        if (c == 0) ML_STAMP(4);
        const bf16x8 zb[2] = {as_bf16x8(make_uint4(zp[0], zp[1], zp[2], zp[3])), as_bf16x8(make_uint4(zp[4], zp[5], zp[6], zp[7]))};
        ml_load_w2(wp, w2c, more ? c + 1 : c, tid);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (u + 1 < 3) ml_read_w2(w2r, u + 1, wf[(u + 1) & 1]);
            out[2 * u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u & 1][0], zb[0], out[2 * u], 0, 0, 0);
            out[2 * u + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u & 1][2], zb[0], out[2 * u + 1], 0, 0, 0);
            out[2 * u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u & 1][1], zb[1], out[2 * u], 0, 0, 0);
            out[2 * u + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u & 1][3], zb[1], out[2 * u + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c == 0) ML_STAMP(5);
        lds_barrier();   // every wave has read the chunk
        if (more) ml_write_w(wp, W1s, W2s, tid);
        lds_barrier();
        if (c == 0) ML_STAMP(6);
    }

    // y = x + r * (out + b2): each 32-column tile through the wave's staging rows, out as 128-byte row segments
    ML_STAMP(7);
    const float rs = *rscale;
    unsigned char *stg = stg_all + wave * 32 * ML_SROW;
    // the residual's values in the shape of the stores below, all requested before the first tile is staged (the
    // operand registers are spent by now), so that their way from memory is waited for once
    float4 xres[6][4];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wave * 32 + i * 8 + (lane >> 3), col = t * 32 + (lane & 7) * 4;
            xres[t][i] = (m < M && col + 4 <= C) ? *reinterpret_cast<const float4 *>(x + (size_t)m * ldx + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    ML_STAMP(8);
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        if (t * 32 >= C) break;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(stg + l31 * ML_SROW + (8 * q + 4 * hh) * 4) =
                make_float4(out[t][4 * q], out[t][4 * q + 1], out[t][4 * q + 2], out[t][4 * q + 3]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + (lane >> 3), m = m0 + wave * 32 + row, col = t * 32 + (lane & 7) * 4;
            const float4 v = *reinterpret_cast<const float4 *>(stg + row * ML_SROW + (lane & 7) * 16);
            if (m < M && col + 4 <= C) {
                const float4 xr = xres[t][i];
                const float4 b = *reinterpret_cast<const float4 *>(b2s + col);
                *reinterpret_cast<float4 *>(y + (size_t)m * ldy + col) =
                    make_float4(xr.x + rs * (v.x + b.x), xr.y + rs * (v.y + b.y), xr.z + rs * (v.z + b.z), xr.w + rs * (v.w + b.w));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    ML_STAMP(9);
}

}  // namespace
}  // namespace dsw

using namespace dsw;

extern "C" int dsw_mlp_gated_residual(const float *x, long long ldx, long long M, int C, float norm_eps, const void *w1c, const float *b1c,
                                      const void *w2c, const float *b2, int hidden_chunks, const float *rscale, float *y, long long ldy,
                                      void *stream) {
    if (M < 0 || M > 0x7fffff00LL || C <= 0) return fail(DSW_EINVAL, "dsw_mlp_gated_residual: bad sizes M=%lld C=%d", M, C);
    if (M == 0) return DSW_OK;
    DSW_REQUIRE(x && w1c && b1c && w2c && b2 && rscale && y, "dsw_mlp_gated_residual: null pointer");
    if (C > ML_K || C % 4 || ldx < C || ldy < C || ldx % 4 || ldy % 4 || hidden_chunks < 1 || hidden_chunks > ML_MAXCHUNKS)
        return fail(DSW_EUNSUPPORTED, "dsw_mlp_gated_residual: C=%d (a multiple of 4, <= %d), row strides %lld / %lld (multiples of 4, >= C), "
                    "%d hidden chunks (1..%d)", C, ML_K, ldx, ldy, hidden_chunks, ML_MAXCHUNKS);
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w1c | (uintptr_t)w2c | (uintptr_t)b1c | (uintptr_t)b2) & 15)
        return fail(DSW_EUNSUPPORTED, "dsw_mlp_gated_residual: operands must be 16-byte aligned");
    const size_t lds = (size_t)64 * ML_LS1 * 2 + (size_t)ML_N * ML_LS2 * 2 + ((size_t)hidden_chunks * 64 + ML_N + ML_BM) * 4 + (size_t)4 * 32 * ML_SROW;
    static_assert((size_t)ML_BM * ML_LS1 * 2 <= (size_t)64 * ML_LS1 * 2 + (size_t)ML_N * ML_LS2 * 2 + (size_t)4 * 32 * ML_SROW, "the rows of x fit over the weight chunk and the staging");
    auto kern = mlp_gated_residual_kernel;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DSW_ELAUNCH, "dsw_mlp_gated_residual: LDS attribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((M + ML_BM - 1) / ML_BM)), dim3(ML_THREADS), lds, as_stream(stream), x, ldx, (int)M, C, norm_eps,
                       (const bf16_t *)w1c, b1c, (const bf16_t *)w2c, b2, hidden_chunks, rscale, y, ldy);
    return check_launch("mlp_gated_residual_kernel");
}

namespace dsw {
namespace {
// the operand images of mlp_gated_residual_kernel from the modules' fp32 parameters, one launch
__global__ __launch_bounds__(256) void mlp_gated_prep_kernel(const float *__restrict__ gate_w, const float *__restrict__ content_w,
                                                             const float *__restrict__ gate_b, const float *__restrict__ content_b,
                                                             const float *__restrict__ fc2_w, const float *__restrict__ fc2_b,
                                                             const float *__restrict__ norm_w, int H, int C, int N, int nchunk,
                                                             bf16_t *__restrict__ w1c, float *__restrict__ b1c, bf16_t *__restrict__ w2c,
                                                             float *__restrict__ b2) {
    const int R1 = nchunk * 64 * ML_K, R2 = nchunk * ML_N * ML_HC, R3 = nchunk * 64;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < R1) {
        const int c = i / (64 * ML_K), r = (i / ML_K) % 64, k = i % ML_K;
        const int h = 32 * c + (r & 31);
        const float *w = r < 32 ? gate_w : content_w;
        w1c[i] = f32_to_bf16((h < H && k < C) ? w[(size_t)h * C + k] * (norm_w != nullptr ? norm_w[k] : 1.f) : 0.f);
        return;
    }
    i -= R1;
    if (i < R2) {
        // position p of a chunk's 32 hidden columns: every 16 in the order 0-3, 8-11, 4-7, 12-15
        const int c = i / (ML_N * ML_HC), o = (i / ML_HC) % ML_N, p = i % ML_HC;
        const int q = p & 15, h = 32 * c + (p & 16) + 8 * ((q >> 2) & 1) + 4 * (q >> 3) + (q & 3);
        w2c[i] = f32_to_bf16((o < N && h < H) ? fc2_w[(size_t)o * H + h] : 0.f);
        return;
    }
    i -= R2;
    if (i < R3) {
        const int c = i / 64, r = i % 64, h = 32 * c + (r & 31);
        const float *b = r < 32 ? gate_b : content_b;
        b1c[i] = (b != nullptr && h < H) ? b[h] : 0.f;
        return;
    }
    i -= R3;
    if (i < ML_N) b2[i] = (fc2_b != nullptr && i < N) ? fc2_b[i] : 0.f;
}
}  // namespace
}  // namespace dsw

extern "C" int dsw_mlp_gated_prep(const float *gate_w, const float *content_w, const float *gate_b, const float *content_b, const float *fc2_w,
                                  const float *fc2_b, const float *norm_w, int H, int C, int N, void *w1c, float *b1c, void *w2c, float *b2,
                                  void *stream) {
    if (H <= 0 || C <= 0 || N <= 0 || C > ML_K || N > ML_N || (H + 31) / 32 > ML_MAXCHUNKS)
        return fail(DSW_EUNSUPPORTED, "dsw_mlp_gated_prep: H=%d C=%d N=%d (C <= %d, N <= %d, H <= %d)", H, C, N, ML_K, ML_N, 32 * ML_MAXCHUNKS);
    DSW_REQUIRE(gate_w && content_w && fc2_w && w1c && b1c && w2c && b2, "dsw_mlp_gated_prep: null pointer");
    const int nchunk = (H + 31) / 32;
    const int total = nchunk * 64 * ML_K + nchunk * ML_N * ML_HC + nchunk * 64 + ML_N;
    hipLaunchKernelGGL(mlp_gated_prep_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), gate_w, content_w, gate_b, content_b,
                       fc2_w, fc2_b, norm_w, H, C, N, nchunk, (bf16_t *)w1c, b1c, (bf16_t *)w2c, b2);
    return check_launch("mlp_gated_prep_kernel");
}

#ifdef ML_STAMPS
extern "C" int dsw_debug_read_mlp_stamps(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dsw::g_mlp_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif
