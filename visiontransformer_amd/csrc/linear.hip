// The Linear projections of the attention modules (DSwinT_arch.py:303, :308, :334, :337) as one
// MFMA kernel for their shape: millions of rows, a few hundred columns on either side.
//
//   C[M, ncols] = A[M, K] . Wp[ncols, Kp]^T (+ bias)
//
// A is streamed from HBM exactly once per block column, in whatever dtype / row stride the producer
// left it (fp32 activations are rounded to bf16 on the way into LDS; an optional bf16 copy of A in
// padded rows is written on the side for the weight-gradient GEMM), the weight is a small padded
// bf16 matrix that stays in L2, and the bias add and the output cast happen in the accumulators.
//
// Work split: a workgroup owns 128 rows of A (32 per wave).  The rows' k-slab (up to 192 wide) goes
// through LDS once and is then held as MFMA operands in registers, while the weight is walked in
// chunks of 64 output columns through LDS: one 16-byte LDS read per MFMA.
//   * K <= 192 (forward projections): one slab, any number of column chunks, each stored as soon as
//     it is complete.
//   * K > 192 (input gradients, K = 3C): several slabs; the accumulators of up to three chunks
//     (192 columns) persist across slabs, wider outputs are split over blockIdx.y.
#include "attn_common.hpp"

namespace dsw {
namespace {

constexpr int LN_BM = 128;       // rows of A per workgroup
constexpr int LN_CN = 64;        // output columns per chunk
constexpr int LN_THREADS = 256;  // 4 waves, 32 rows each
constexpr int LN_MAXACC = 3;     // chunks whose accumulators persist across k slabs; 4 with slabs of <= 128 (LN_MAXACC_NARROW)
constexpr int LN_MAXACC_NARROW = 4;  // 256 columns: msft's 216-wide outputs in one pass over A

template <int KS> struct LnCfg {
    static constexpr int LS = KS + 8;  // LDS row stride in elements: 16-byte reads of 32 rows fall on distinct banks
    static constexpr int NKS = KS / 16;
    // A slab wider than 192 (256: msft's 216-wide rows) passes through LDS in two halves - the rows only cross LDS
    // on their way into registers - so that two workgroups still fit a CU; the weight chunk has the slab's full width.
    static constexpr int AH = KS > 192 ? 2 : 1;
    static constexpr int AW = KS / AH, ALS = AW + 8;
    static constexpr int A_PIECES = 4 * (AW / 64);  // 8-element pieces per thread: half of an A slab, weight chunk
    static constexpr int B_PIECES = 2 * (KS / 64);
    static constexpr size_t LDS = ((size_t)LN_BM * ALS + (size_t)LN_CN * LS) * sizeof(bf16_t);
};

// Buffer descriptors: loads past the end of the described range return zero, which takes care of the rows beyond M, of weight rows beyond the matrix and of an
// absent bias without a branch.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(bytes > 0xffffffffull ? 0xffffffffull : bytes), 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// 8 consecutive elements of a row of A as they come from memory (k .. k+7 of a row with K valid
// columns: the part at or beyond K reads as zero)
template <typename T> struct a_piece;
template <> struct a_piece<float> {
    uint4 lo, hi;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, bool) {
        lo = buf_load16(r, voff, soff);
        hi = buf_load16(r, voff + 16, soff);
    }
    __device__ __forceinline__ uint4 pack(int k, int K) const {
        const bool l = k + 4 <= K, h = k + 8 <= K;
        uint4 v;
        v.x = l ? pack_bf16(__uint_as_float(lo.x), __uint_as_float(lo.y)) : 0u;
        v.y = l ? pack_bf16(__uint_as_float(lo.z), __uint_as_float(lo.w)) : 0u;
        v.z = h ? pack_bf16(__uint_as_float(hi.x), __uint_as_float(hi.y)) : 0u;
        v.w = h ? pack_bf16(__uint_as_float(hi.z), __uint_as_float(hi.w)) : 0u;
        return v;
    }
    // inverted dropout of the 8 elements, e0 = index of the first in the dense matrix the decisions are defined on
    __device__ __forceinline__ void drop(long long e0, const DropArgs &d) {
        const float4 a = drop4_seeded(make_float4(__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(lo.z), __uint_as_float(lo.w)), e0, d);
        const float4 b = drop4_seeded(make_float4(__uint_as_float(hi.x), __uint_as_float(hi.y), __uint_as_float(hi.z), __uint_as_float(hi.w)), e0 + 4, d);
        lo = make_uint4(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w));
        hi = make_uint4(__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z), __float_as_uint(b.w));
    }
    // sum of the squares of the elements in front of column K (the fp32 values as they are in memory)
    __device__ __forceinline__ float sumsq(int k, int K) const {
        const bool l = k + 4 <= K, h = k + 8 <= K;
        const float a0 = __uint_as_float(lo.x), a1 = __uint_as_float(lo.y), a2 = __uint_as_float(lo.z), a3 = __uint_as_float(lo.w);
        const float b0 = __uint_as_float(hi.x), b1 = __uint_as_float(hi.y), b2 = __uint_as_float(hi.z), b3 = __uint_as_float(hi.w);
        return (l ? a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3 : 0.f) + (h ? b0 * b0 + b1 * b1 + b2 * b2 + b3 * b3 : 0.f);
    }
};
template <> struct a_piece<bf16_t> {
    uint4 v;
    // half16: rows are only 8-byte aligned (a dense [M][K] matrix with K % 8 == 4): two 8-byte loads
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, bool half16) {
        if (half16) {
            const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
            const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff + 8, (int)soff, 0);
            v = make_uint4(a.x, a.y, b.x, b.y);
        } else {
            v = buf_load16(r, voff, soff);
        }
    }
    __device__ __forceinline__ uint4 pack(int k, int K) const {
        const bool l = k + 4 <= K, h = k + 8 <= K;
        return make_uint4(l ? v.x : 0u, l ? v.y : 0u, h ? v.z : 0u, h ? v.w : 0u);
    }
    __device__ __forceinline__ void drop(long long, const DropArgs &) {}   // the masked operand is always the fp32 gradient
    __device__ __forceinline__ float sumsq(int k, int K) const {
        const bool l = k + 4 <= K, h = k + 8 <= K;
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = __uint_as_float(u[q] << 16), b = __uint_as_float(u[q] & 0xffff0000u);
            s += (q < 2 ? l : h) ? a * a + b * b : 0.f;
        }
        return s;
    }
};

// One chunk of 64 output columns against the wave's 32 rows: 2 MFMAs per k step, their weight
// operands read from LDS one group of two k steps ahead (the scheduling fences keep the compiler
// from hoisting every read of the chunk to the top, which costs 100 registers).
template <int NKS, int LS>
__device__ __forceinline__ void chunk_mfma(const bf16_t *brow, const bf16x8 (&afr)[NKS], f32x16 (&acc)[2]) {
    constexpr int NG = NKS / 2;
    bf16x8 w[2][4];
    auto read_group = [&](int g, bf16x8(&dst)[4]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                dst[2 * j + t] = as_bf16x8(*reinterpret_cast<const uint4 *>(brow + t * 32 * LS + (2 * g + j) * 16));
    };
    read_group(0, w[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) read_group(g + 1, w[(g + 1) & 1]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#ifdef LN_NO_MFMA   // timing experiment (wrong results): one MFMA per group of four
                { if (j == 0 && t == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[g & 1][2 * j + t], afr[2 * g + j], acc[t], 0, 0, 0); }
#else
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[g & 1][2 * j + t], afr[2 * g + j], acc[t], 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
}

// NACC: 0 = one k slab (a chunk is stored as soon as it is complete); > 0 = several slabs, the accumulators of NACC
// chunks persist across them
// LINEAR_STAMPS (diagnostic builds only, tools/linear_stamps.py): thread 0 of every workgroup records the shader clock
// at points of the projection kernel into a device array that dsw_debug_read_linear_stamps copies out
#ifdef LINEAR_STAMPS
__device__ unsigned long long g_linear_stamps[8 * 8192];
#define LN_STAMP(i)                                                                                            \
    do {                                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.y == 0) g_linear_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define LN_STAMP(i) ((void)0)
#endif

// MODE 1 (EXT): the instantiations that know the epilogues beyond bias / residual (gated SiLU, GELU and its slope) and
// the RMS normalisation of A's rows; the plain ones (MODE 0) carry none of their code or registers.
// MODE 2 (CONV): the 3x3 convolution on NHWC pixels (several-slab kernels only), see the staging below.  MODE 3: the
// normalisation alone (single-slab kernels).
template <typename TA, typename TC, int KS, int NACC, int MODE>
__global__ __launch_bounds__(LN_THREADS, 2) void linear_nt_kernel(
    const TA *__restrict__ A, long long lda, int M, int K, const bf16_t *__restrict__ Wp, int ldw, int wrows, int nslab,
    const float *__restrict__ bias, int N, TC *__restrict__ C, long long ldc, int ncols, bf16_t *__restrict__ a_copy,
    long long lda_copy, int chunks_per_block, const float *__restrict__ resid, long long ld_resid, const float *__restrict__ rscale,
    int convH, int convW, int gated_arg, const int *__restrict__ live_units, int unit_rows, bf16_t *__restrict__ aux, long long ld_aux,
    float norm_eps, DropArgs dropa, int drop_C) {
    // MODE 1: gated SiLU (with or without the normalisation); 3: the normalisation alone; 4: the GELU epilogues; 5: dropout
    // dropa.thr != 0 (MODE 4, 5): inverted dropout with seeded decisions on the dense [M][drop_C] matrix (the residual
    // dropout of the callers' blocks, msft_arch.py:163-165 / DSwinT_arch.py:355-357).  fp32 A: A is a gradient that is
    // masked on its way in (the copy for the weight gradient receives the masked values).  fp32 C with a residual
    // (MODE 5): C = resid + rs * dropout(A W^T + bias).
    constexpr bool CONV = MODE == 2, NORM = MODE == 1 || MODE == 3;
    constexpr bool DROP_IN = (MODE == 4 || MODE == 5) && sizeof(TA) == 4, DROP_OUT = MODE == 5 && sizeof(TC) == 4;
    static_assert(!CONV || NACC > 0, "the convolution runs over several slabs");
    const int gated = MODE == 1 ? (gated_arg == 1 ? 1 : 0) : MODE == 4 ? (gated_arg == 3 ? 3 : 2) : 0;
    // norm_eps >= 0 (one slab only): the rows of A are RMS-normalised on the way (nn.RMSNorm in front of the projection,
    // DSwinT_arch.py:355 / :357 with :343-345): C = (A W^T) / sqrt(mean_k A^2 + eps) + bias per row, the norm's weight
    // having been multiplied into W's columns by the caller.  The mean runs over the fp32 values as they are loaded.
    // gated (the epilogue of a bf16 result): 1 = gated SiLU; 2 = GELU, the pre-activation values going to aux when it is
    // given; 3 = the result times GELU's slope at aux's values (the backward of 2 on the way out of the input gradient)
    using cfg = LnCfg<KS>;
    constexpr int LS = cfg::LS, NKS = cfg::NKS, AH = cfg::AH, AW = cfg::AW, ALS = cfg::ALS;
    constexpr bool PERSIST = NACC > 0;
    static_assert(AH == 1 || !PERSIST, "a slab that is staged in halves is the only slab");
    extern __shared__ __align__(16) unsigned char smem[];
    bf16_t *As = reinterpret_cast<bf16_t *>(smem);
    bf16_t *Bs = As + (LN_BM + (CONV ? 2 : 0)) * ALS;
    float *biasl = reinterpret_cast<float *>(Bs + LN_CN * LS);  // bias of this block's columns, zero where there is none
    float *rstdl = biasl + chunks_per_block * LN_CN;            // 1 / rms of the block's rows (norm_eps >= 0)
    const bool prenorm = NORM && !PERSIST && AH == 1 && norm_eps >= 0.f;  // the single-slab kernels of K <= 192 only

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    LN_STAMP(0);
    // fp32 results only: C = resid + rs * (A W^T + bias); resid == C is plain accumulation
    const float rs = rscale != nullptr ? *rscale : 1.f;
    const int m0 = blockIdx.x * LN_BM;
    const int rows = min(M - m0, LN_BM);
    const bool half16 = sizeof(TA) == 2 && (((lda | K) & 7) != 0 || (reinterpret_cast<uintptr_t>(A) & 15) != 0);
    const int total_chunks = (ncols + LN_CN - 1) / LN_CN;
    const int nc0 = blockIdx.y * chunks_per_block;
    const int nch = min(chunks_per_block, total_chunks - nc0);  // >= 1 by the grid's construction
    if (live_units != nullptr) {
        // Rows at and beyond *live_units * unit_rows belong to units (windows) the caller has switched off on the
        // device: a block that holds none of the live rows does no work.  It leaves zeros in its part of the result
        // (and of the bf16 copy), so that whatever reads these rows later finds finite values; with accumulation
        // into the destination the rows are left as they are.
        const long long live_rows = (long long)*live_units * unit_rows;
        if (m0 >= live_rows) {
            if (resid == nullptr) {
                const int c0 = nc0 * (gated == 1 ? LN_CN / 2 : LN_CN), c1 = min((nc0 + nch) * (gated == 1 ? LN_CN / 2 : LN_CN), gated == 1 ? ncols / 2 : ncols);
                const int cq = (c1 - c0) >> 2;
                for (int i = tid; i < rows * cq; i += LN_THREADS) {
                    const int r = i / cq, c = c0 + ((i - r * cq) << 2);
                    io<TC>::st4(C + (size_t)(m0 + r) * ldc + c, f4zero());
                }
            }
            if (a_copy != nullptr && blockIdx.y == 0) {
                const int Kq = (int)(lda_copy >> 2) < ((K + 63) / 64 * 64) >> 2 ? (int)(lda_copy >> 2) : ((K + 63) / 64 * 64) >> 2;
                for (int i = tid; i < rows * Kq; i += LN_THREADS) {
                    const int r = i / Kq, c = (i - r * Kq) << 2;
                    io<bf16_t>::st4(a_copy + (size_t)(m0 + r) * lda_copy + c, f4zero());
                }
            }
            return;
        }
    }

    // staging: 8 threads per row, 8 elements each = 64 columns; a slab is KS / 64 such column groups
    // (immediate offsets) over 4 (A: 128 rows) or 2 (weight chunk: 64 rows) row groups
    const int srow = tid >> 3, scol = (tid & 7) * 8;
    // 3x3 convolution (CONV): the rows are the pixels of NHWC images.  For every tap row dy and k slab the 130 pixels
    // m0 - 1 .. m0 + 128 of the image row(s) shifted by dy go through LDS ONCE (rows of another image row, or outside
    // the image, as zeros: the descriptor covers the whole tensor and such a row gets an offset beyond it), and the
    // three taps dx = -1, 0, 1 read their MFMA operands from LDS rows r + dx: the input leaves L2 three times, not nine.
    // A lane whose pixel sits on the image's left / right edge zeroes its operand for dx = -1 / +1 (the LDS row next
    // to it belongs to the neighbouring image row).
    constexpr bool conv = CONV;
    const __amdgpu_buffer_rsrc_t rs_a = conv ? make_rsrc(A, ((size_t)(M - 1) * lda + K) * sizeof(TA))
                                             : make_rsrc(A + (size_t)m0 * lda, ((size_t)(rows - 1) * lda + K) * sizeof(TA));
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(Wp, (size_t)wrows * ldw * sizeof(bf16_t));
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(bias, bias != nullptr ? (size_t)N * sizeof(float) : 0);
    // bf16 copy of A: plain stores from the first block column
    bf16_t *const copy_row = a_copy != nullptr && blockIdx.y == 0 ? a_copy + (size_t)(m0 + srow) * lda_copy + scol : nullptr;
    for (int i = tid; i < nch * LN_CN; i += LN_THREADS)
        biasl[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_b, (unsigned)(nc0 * LN_CN + i) * 4u, 0, 0));
    unsigned a_off[4], w_off[2];
#pragma unroll
    for (int p = 0; p < 4; ++p) a_off[p] = (unsigned)(((size_t)(p * 32 + srow) * lda + scol) * sizeof(TA));
    int py[5];   // CONV: image row of the pixel in LDS row p * 32 + srow (pixel m0 - 1 + that), -4 = no such pixel
    if constexpr (CONV) {
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            const int px = m0 - 1 + p * 32 + srow;
            py[p] = (px >= 0 && px < M && (p < 4 || srow < 2)) ? (px / convW) % convH : -4;
        }
    }
    const int kslabs = conv ? nslab / 9 : nslab;
#pragma unroll
    for (int q = 0; q < 2; ++q) w_off[q] = (unsigned)(((q * 32 + srow) * ldw + scol) * sizeof(bf16_t));

    f32x16 acc[PERSIST ? NACC : 1][2];
#pragma unroll
    for (int c = 0; c < (PERSIST ? NACC : 1); ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;

    // Results leave through LDS: a lane holds 4 consecutive columns of one row per register group, which
    // stored directly would touch 32 rows with 16 bytes each per instruction.  Each wave lays its tile
    // out in its own 32 rows of As (spent once the operands are in registers; no other wave reads
    // them) and stores whole 128-byte row segments, 8 rows per instruction.
    auto store_tiles = [&](const f32x16(&a)[2], int chunk) {
        constexpr int SW = 128 / sizeof(TC);  // columns per pass
        constexpr int TPP = SW / 32;          // 32-column tiles per pass: 2 (bf16), 1 (fp32)
        constexpr int PC = 16 / sizeof(TC);   // columns per 16-byte piece
        constexpr int SROW = 144;             // staging row stride in bytes
        unsigned char *stg = reinterpret_cast<unsigned char *>(As + wave * 32 * ALS);
        const int nb = (nc0 + chunk) * LN_CN;
        const float rn = prenorm ? rstdl[wave * 32 + l31] : 1.f;  // a lane's accumulators all belong to one row of A
        // the bias comes from LDS (staged once per block)
        float4 bv[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[t][g] = *reinterpret_cast<const float4 *>(biasl + chunk * LN_CN + 4 * hh + t * 32 + 8 * g);
        if constexpr (sizeof(TC) == 2) {
            if (gated == 1) {
                // Gated-SiLU epilogue (DSwinT_arch.py:69): the chunk's first 32 weight rows are gate rows, the other 32
                // the content rows of the same hidden columns, so a lane holds gate and content of one position in
                // the same register of the two tiles.  Both are rounded to bf16 first - what the stand-alone
                // projection would have written - and the chunk leaves as 32 columns (64 bytes per row).
                const int ob = (nc0 + chunk) * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float y[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float bg = j == 0 ? bv[0][g].x : j == 1 ? bv[0][g].y : j == 2 ? bv[0][g].z : bv[0][g].w;
                        const float bc = j == 0 ? bv[1][g].x : j == 1 ? bv[1][g].y : j == 2 ? bv[1][g].z : bv[1][g].w;
                        const float gate = bf16_to_f32(f32_to_bf16(a[0][4 * g + j] * rn + bg));
                        const float cont = bf16_to_f32(f32_to_bf16(a[1][4 * g + j] * rn + bc));
                        y[j] = gated_silu_value(gate, cont);
                    }
                    io<bf16_t>::st4(reinterpret_cast<bf16_t *>(stg + l31 * SROW) + 8 * g + 4 * hh, make_float4(y[0], y[1], y[2], y[3]));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = i * 16 + (lane >> 2), m = m0 + wave * 32 + row, col = ob + (lane & 3) * 8;
                    const uint4 v = *reinterpret_cast<const uint4 *>(stg + row * SROW + (lane & 3) * 16);
                    bf16_t *dst = reinterpret_cast<bf16_t *>(C) + (size_t)m * ldc + col;
                    if (m < M && col + 8 <= (ncols >> 1)) *reinterpret_cast<uint4 *>(dst) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                return;
            }
        }
#pragma unroll
        for (int ps = 0; ps < 2 / TPP; ++ps) {
            // the residual's values are asked for before the tile goes through LDS, so that their latency
            // overlaps the staging
            float4 ro[4];
            uint4 hv[4];
            if constexpr (sizeof(TC) == 2) {
                if (gated == 3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = m0 + wave * 32 + i * 8 + (lane >> 3), col = nb + (lane & 7) * PC;
                        hv[i] = (m < M && col + PC <= ncols) ? *reinterpret_cast<const uint4 *>(aux + (size_t)m * ld_aux + col) : make_uint4(0u, 0u, 0u, 0u);
                    }
                }
            }
            if constexpr (sizeof(TC) == 4) {
                if (resid != nullptr) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = m0 + wave * 32 + i * 8 + (lane >> 3), col = nb + ps * SW + (lane & 7) * PC;
                        ro[i] = (m < M && col + PC <= ncols) ? *reinterpret_cast<const float4 *>(resid + (size_t)m * ld_resid + col)
                                                              : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
#pragma unroll
            for (int tt = 0; tt < TPP; ++tt) {
                const int t = ps * TPP + tt;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v = make_float4(a[t][4 * g] * rn + bv[t][g].x, a[t][4 * g + 1] * rn + bv[t][g].y, a[t][4 * g + 2] * rn + bv[t][g].z,
                                           a[t][4 * g + 3] * rn + bv[t][g].w);
                    if constexpr (DROP_OUT) {
                        if (dropa.thr != 0u) v = drop4_seeded(v, (long long)(m0 + wave * 32 + l31) * drop_C + nb + t * 32 + 8 * g + 4 * hh, dropa);
                    }
                    // bf16: a lane stores 8 bytes, and the rows 8 apart that one store instruction's 16-lane groups hold
                    // are 288 dwords apart - the same banks.  Rows 8-15 and 24-31 therefore keep the two halves of
                    // every 16-byte piece swapped (the reader below swaps them back in registers, for free: which
                    // rows it holds is a compile-time constant per pass).
                    const int swz = sizeof(TC) == 2 ? ((l31 >> 3) & 1) * 4 : 0;
                    io<TC>::st4(reinterpret_cast<TC *>(stg + l31 * SROW) + ((tt * 32 + 8 * g + 4 * hh) ^ swz), v);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + (lane >> 3), m = m0 + wave * 32 + row;
                const int col = nb + ps * SW + (lane & 7) * PC;
                uint4 v = *reinterpret_cast<const uint4 *>(stg + row * SROW + (lane & 7) * 16);
                if (sizeof(TC) == 2 && (i & 1)) v = make_uint4(v.z, v.w, v.x, v.y);
                TC *dst = C + (size_t)m * ldc + col;
                if constexpr (sizeof(TC) == 2) {
                    if (gated == 2) {
                        // v holds what the plain projection would have written: it goes to aux, its GELU to the result
                        if (aux != nullptr && m < M && col + PC <= ncols) *reinterpret_cast<uint4 *>(aux + (size_t)m * ld_aux + col) = v;
                        const unsigned in[4] = {v.x, v.y, v.z, v.w};
                        unsigned o[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            o[q] = pack_bf16(gelu_value<bf16_t>(__uint_as_float(in[q] << 16)), gelu_value<bf16_t>(__uint_as_float(in[q] & 0xffff0000u)));
                        v = make_uint4(o[0], o[1], o[2], o[3]);
                    } else if (gated == 3) {
                        const unsigned in[4] = {v.x, v.y, v.z, v.w}, hx[4] = {hv[i].x, hv[i].y, hv[i].z, hv[i].w};
                        unsigned o[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            o[q] = pack_bf16(__uint_as_float(in[q] << 16) * gelu_slope<bf16_t>(__uint_as_float(hx[q] << 16)),
                                             __uint_as_float(in[q] & 0xffff0000u) * gelu_slope<bf16_t>(__uint_as_float(hx[q] & 0xffff0000u)));
                        v = make_uint4(o[0], o[1], o[2], o[3]);
                    }
                }
                if (m < M) {
                    if constexpr (sizeof(TC) == 4) {
                        if (resid != nullptr && col + PC <= ncols) {
                            const float4 o = ro[i];
                            v = make_uint4(__float_as_uint(o.x + rs * __uint_as_float(v.x)), __float_as_uint(o.y + rs * __uint_as_float(v.y)),
                                           __float_as_uint(o.z + rs * __uint_as_float(v.z)), __float_as_uint(o.w + rs * __uint_as_float(v.w)));
                        }
                    }
                    if (col + PC <= ncols) *reinterpret_cast<uint4 *>(dst) = v;
                    else if (sizeof(TC) == 2 && col + 4 <= ncols) *reinterpret_cast<uint2 *>(dst) = make_uint2(v.x, v.y);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    };

    uint4 breg[cfg::B_PIECES];
    auto load_b = [&](int slab, int chunk) {
#ifdef LN_NO_WLOAD
        if (chunk > 0) return;   // timing experiment (wrong results): the weight chunks after the first are not read
#endif
        const unsigned soff = (unsigned)(((size_t)(nc0 + chunk) * LN_CN * ldw + slab * KS) * sizeof(bf16_t));
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < KS / 64; ++j) breg[q * (KS / 64) + j] = buf_load16(rs_w, w_off[q] + j * 128, soff);
    };
    auto write_b = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < KS / 64; ++j)
                *reinterpret_cast<uint4 *>(Bs + (q * 32 + srow) * LS + j * 64 + scol) = breg[q * (KS / 64) + j];
    };

    const int xw_lane = CONV ? (m0 + wave * 32 + l31) % convW : 0;  // CONV: image column of this lane's pixel
    for (int slab = 0; slab < nslab; ++slab) {
        bf16x8 afr[NKS];  // this wave's 32 rows of the slab as MFMA operands
        // CONV: slab = (tap row, k slab, tap column), the column innermost; wslab = the weight's slab (tap-major along K)
        int wslab = slab, wslab_next = slab + 1, dxi = 0;
        if constexpr (CONV) {
            dxi = slab % 3;
            const int t2 = slab / 3, s1 = slab + 1, t3 = s1 / 3;
            wslab = ((t2 / kslabs) * 3 + dxi) * kslabs + t2 % kslabs;
            wslab_next = ((t3 / kslabs) * 3 + s1 % 3) * kslabs + t3 % kslabs;
        }
        if (!CONV || dxi == 0) {
            constexpr int NP = CONV ? 5 : 4;  // row groups of the staging: the strip's two extra rows make a fifth
            a_piece<TA> ap[AH][NP * (AW / 64)];
            const int tap = CONV ? 0 : slab / kslabs, kslab = CONV ? (slab / 3) % kslabs : slab - tap * kslabs;
            const int dyi = CONV ? slab / 3 / kslabs : 1;
            const unsigned soff = (unsigned)(kslab * KS * sizeof(TA));
            unsigned offs[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if constexpr (CONV) {
                    const int yy = py[p] + dyi - 1;
                    offs[p] = (py[p] != -4 && yy >= 0 && yy < convH)
                                  ? (unsigned)(((size_t)((long long)m0 - 1 + p * 32 + srow + (long long)(dyi - 1) * convW) * lda + scol) * sizeof(TA))
                                  : 0x80000000u;
                } else {
                    offs[p] = a_off[p];
                }
            }
#pragma unroll
            for (int h = 0; h < AH; ++h)
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int j = 0; j < AW / 64; ++j)
                        ap[h][p * (AW / 64) + j].load(rs_a, offs[p] + (h * AW + j * 64) * (unsigned)sizeof(TA), soff, half16);
            if (slab == 0) load_b(0, 0);
            if (slab == 0) LN_STAMP(1);
            if constexpr (DROP_IN) {
                if (dropa.thr != 0u) {
#pragma unroll
                    for (int h = 0; h < AH; ++h)
#pragma unroll
                        for (int p = 0; p < 4; ++p)
#pragma unroll
                            for (int j = 0; j < AW / 64; ++j)
                                ap[h][p * (AW / 64) + j].drop((long long)(m0 + p * 32 + srow) * drop_C + kslab * KS + h * AW + j * 64 + scol, dropa);
                }
            }
            if (prenorm) {
                // a row's columns lie with 8 neighbouring lanes (all halves and column groups of the slab)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float ss = 0.f;
#pragma unroll
                    for (int h = 0; h < AH; ++h)
#pragma unroll
                        for (int j = 0; j < AW / 64; ++j) ss += ap[h][p * (AW / 64) + j].sumsq(h * AW + j * 64 + scol, K);
                    ss += __shfl_xor(ss, 1, 64);
                    ss += __shfl_xor(ss, 2, 64);
                    ss += __shfl_xor(ss, 4, 64);
                    if ((tid & 7) == 0) rstdl[p * 32 + srow] = rsqrtf(ss / (float)K + norm_eps);
                }
            }
#pragma unroll
            for (int h = 0; h < AH; ++h) {
                lds_barrier();  // the previous reads of As / Bs are over
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int j = 0; j < AW / 64; ++j) {
                        const int kc = kslab * KS + h * AW + j * 64;  // first column of this thread's group of 64
                        const uint4 v = ap[h][p * (AW / 64) + j].pack(kc + scol, K);
                        if (p < 4 || srow < 2) *reinterpret_cast<uint4 *>(As + (p * 32 + srow) * ALS + j * 64 + scol) = v;
                        // the bf16 copy of the block's own rows (CONV: the strip of the centre tap row holds them from its
                        // second row on); a half-empty last slab: the columns at and beyond the padded K are not part of it
                        if constexpr (CONV) {
                            const int own = p * 32 + srow - 1;
                            if (copy_row != nullptr && dyi == 1 && own >= 0 && own < rows && (p < 4 || srow < 2) && kc < ((K + 63) & ~63))
                                *reinterpret_cast<uint4 *>(copy_row + ((long long)p * 32 - 1) * lda_copy + kc) = v;
                        } else {
                            if (copy_row != nullptr && p * 32 + srow < rows && kc < ((K + 63) & ~63))
                                *reinterpret_cast<uint4 *>(copy_row + (size_t)p * 32 * lda_copy + kc) = v;
                        }
                    }
                if (h == AH - 1) write_b();
                if (slab == 0 && h == AH - 1) LN_STAMP(2);
                lds_barrier();
                if (slab == 0 && h == AH - 1) LN_STAMP(3);
                if constexpr (!CONV) {
                    const bf16_t *arow = As + (wave * 32 + l31) * ALS + 8 * hh;
#pragma unroll
                    for (int ks = 0; ks < NKS / AH; ++ks) afr[h * (NKS / AH) + ks] = as_bf16x8(*reinterpret_cast<const uint4 *>(arow + ks * 16));
                }
            }
        } else {
            // CONV, the strip's second and third tap column: only the tap's first weight chunk is new
            lds_barrier();
            write_b();
            lds_barrier();
        }
        if constexpr (CONV) {
            // LDS row r holds pixel m0 - 1 + r: this lane's pixel shifted by dx = dxi - 1 is row wave * 32 + l31 + dxi
            const bf16_t *arow = As + (wave * 32 + l31 + dxi) * ALS + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) afr[ks] = as_bf16x8(*reinterpret_cast<const uint4 *>(arow + ks * 16));
            const bool edge = (dxi == 0 && xw_lane == 0) || (dxi == 2 && xw_lane == convW - 1);
            if (__builtin_amdgcn_ballot_w64(edge) != 0ull) {  // uniform: a wave with a pixel on the image's edge (rare)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    uint4 v = *reinterpret_cast<uint4 *>(&afr[ks]);
                    if (edge) v = make_uint4(0u, 0u, 0u, 0u);
                    afr[ks] = as_bf16x8(v);
                }
            }
        }
        for (int chunk = 0; chunk < nch; ++chunk) {
            if (slab == 0 && chunk == 1) LN_STAMP(4);
            if (chunk > 0) {
                lds_barrier();
                write_b();
                lds_barrier();
            }
            if (chunk + 1 < nch) load_b(wslab, chunk + 1);
            else if (slab + 1 < nslab) load_b(wslab_next, 0);
            const bf16_t *brow = Bs + l31 * LS + 8 * hh;
            if constexpr (PERSIST) {
                // chunk is uniform; a switch keeps the accumulators in registers
#pragma unroll
                for (int cc = 0; cc < NACC; ++cc)
                    if (cc == chunk) chunk_mfma<NKS, LS>(brow, afr, acc[cc]);
            } else {
                chunk_mfma<NKS, LS>(brow, afr, acc[0]);
                store_tiles(acc[0], chunk);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][t][r] = 0.f;
            }
        }
    }
    LN_STAMP(5);
    if constexpr (PERSIST) {
#pragma unroll
        for (int cc = 0; cc < NACC; ++cc)
            if (cc < nch) store_tiles(acc[cc], cc);
    }
    LN_STAMP(6);
}

// w [N, K] (fp32 or bf16) -> zero-padded bf16 copies: wp [Np, Kp] and its transpose wpt [Kp, Np]
template <typename TW>
__global__ __launch_bounds__(256) void weight_prep_kernel(const TW *__restrict__ w, int N, int K, int Np, int Kp,
                                                          bf16_t *__restrict__ wp, bf16_t *__restrict__ wpt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Np * Kp) return;
    const int n = i / Kp, k = i - n * Kp;
    const bf16_t v = (n < N && k < K) ? f32_to_bf16(io<TW>::ld(w + (size_t)n * K + k)) : (bf16_t)0;
    if (wp != nullptr) wp[i] = v;
    if (wpt != nullptr) wpt[(size_t)k * Np + n] = v;
}

// the same for many weights in one launch: a block finds its weight by the first-block table (a scalar loop)
constexpr int WP_BATCH = 48;
struct WpBatch {
    const void *w[WP_BATCH];
    bf16_t *wp[WP_BATCH], *wpt[WP_BATCH];
    int N[WP_BATCH], K[WP_BATCH], first_block[WP_BATCH + 1];
    unsigned long long is_bf16;  // bit i: weight i is bf16 (else fp32)
    int n;
};
__global__ __launch_bounds__(256) void weight_prep_batch_kernel(const WpBatch b) {
    int t = 0;
    while (t + 1 < b.n && (int)blockIdx.x >= b.first_block[t + 1]) ++t;
    const int N = b.N[t], K = b.K[t], Np = (N + 63) / 64 * 64, Kp = (K + 63) / 64 * 64;
    const int i = ((int)blockIdx.x - b.first_block[t]) * 256 + threadIdx.x;
    if (i >= Np * Kp) return;
    const int n = i / Kp, k = i - n * Kp;
    bf16_t v = 0;
    if (n < N && k < K)
        v = (b.is_bf16 >> t) & 1ull ? static_cast<const bf16_t *>(b.w[t])[(size_t)n * K + k]
                                     : f32_to_bf16(static_cast<const float *>(b.w[t])[(size_t)n * K + k]);
    if (b.wp[t] != nullptr) b.wp[t][i] = v;
    if (b.wpt[t] != nullptr) b.wpt[t][(size_t)k * Np + n] = v;
}

// ---- weight gradient ---------------------------------------------------------------------------
// dW[n][k] = sum_m dY[m][n] X[m][k]: a tiny output and a reduction over every token.  A workgroup
// (6 waves) owns a 192 x 192 output block in its accumulators and walks its share of the rows in
// stages of 64, both operands staged row-major in LDS and read as MFMA operands with transposing
// reads (the reduction index m is the row of both tiles).  With K < ceil64(K) the first padding
// column of X is replaced by ones on the way into LDS, which makes column K of the result the bias
// gradient sum_m dY[m][n].  Partial blocks go to a workspace and are summed in a fixed order.
// Two shapes of the output block: 192 x 192 on 6 waves (the padded K of the DSWT widths, <= 192, in one block) and
// 128 x 256 on 8 waves for wider K (msft's 216 -> 256, 432 -> 448: one or two blocks of 256 columns of X instead
// of two or three of 192 with the last one mostly padding).
constexpr int WG_BM = 64;  // rows per stage
template <int TN_, int TK_, int WN_, int WK_> struct WgCfg {
    static constexpr int TN = TN_, TK = TK_;      // output rows (columns of dY) / output columns (columns of X) per workgroup
    static constexpr int WN = WN_, WK = WK_;      // waves along n and k
    static constexpr int THREADS = WN * WK * 64;
    static constexpr int AT = TN / WN / 32, BT = TK / WK / 32;  // 32 x 32 tiles of a wave: AT x BT
    // LDS row strides in elements: (stride / 2) dwords = 16 or 48 mod 64 keeps the transposing reads conflict-free
    static constexpr int LSY = TN + 32, LSX = TK + 32;
    static_assert((LSY / 2) % 32 == 16 && (LSX / 2) % 32 == 16, "row strides of the staged operands");
    static constexpr int YP = TN / 8, XP = TK / 8;                       // 16-byte pieces per row
    static constexpr int YROWS = THREADS / YP, XROWS = THREADS / XP;     // rows per pass of the block
    static constexpr int YN = WG_BM / YROWS, XN = WG_BM / XROWS;         // pieces per thread and operand
    static_assert(THREADS % YP == 0 && THREADS % XP == 0 && WG_BM % YROWS == 0 && WG_BM % XROWS == 0, "staging");
    static constexpr size_t LDS = (size_t)WG_BM * (LSY + LSX) * sizeof(bf16_t);
};
using WgNarrow = WgCfg<192, 192, 3, 2>;
using WgWide = WgCfg<128, 256, 2, 4>;
static bool wgrad_wide(int Kp) { return Kp > WgNarrow::TK; }

template <typename Cfg>
__global__ __launch_bounds__(Cfg::THREADS, 2) void linear_wgrad_kernel(const bf16_t *__restrict__ dY, long long ldy,
                                                                       const bf16_t *__restrict__ X, long long ldx, int M, int K, int Kp,
                                                                       int Np, int Kpr, int rows_per_block, int convH, int convW, float *__restrict__ partial,
                                                                       const int *__restrict__ live_units, int unit_rows, int gx, int gz) {
    constexpr int TN = Cfg::TN, TK = Cfg::TK, LSY = Cfg::LSY, LSX = Cfg::LSX, AT = Cfg::AT, BT = Cfg::BT;
    // The blocks of one row range (all n blocks x all k blocks / taps) are neighbours in the logical order, which
    // xcd_remap keeps on one XCD: they read the same rows of dY and X, which then come out of that XCD's L2
    // instead of once per XCD out of HBM.
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = bid % gx, bz = (bid / gx) % gz, by = bid / (gx * gz);
    extern __shared__ __align__(16) unsigned char smem[];
    bf16_t *Ys = reinterpret_cast<bf16_t *>(smem);
    bf16_t *Xs = Ys + WG_BM * LSY;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int wn = wave % Cfg::WN, wk = wave / Cfg::WN;
    const int n0 = bx * TN;
    const int m_begin = by * rows_per_block;
    // rows at and beyond *live_units * unit_rows are switched off by the caller (see linear_nt_kernel): they add
    // nothing, and a block whose range holds none of the live rows writes a zero partial
    const long long m_end = live_units != nullptr ? min((long long)M, (long long)*live_units * unit_rows) : (long long)M;
    const int rows = (int)max(0ll, min(m_end - m_begin, (long long)rows_per_block));
    const int nst = (rows + WG_BM - 1) / WG_BM;
    // 3x3 convolution mode (convH > 0): rows are the pixels of NHWC images, the block index also runs over the 9
    // taps, and a tap's block reads the rows of X shifted by the tap's offset (zero outside the image): its
    // part of the result is dW[:, tap, :], laid side by side in the partial rows.
    const bool conv = convH > 0;
    const int nkb = Kpr / TK;
    const int tap = conv ? bz / nkb : 0, kb = bz - tap * nkb;
    const int shift = conv ? (tap / 3 - 1) * convW + (tap % 3 - 1) : 0;
    const int k0 = kb * TK;  // this block's columns of X
    const bool ones = K < Kp && K >= k0 && K < k0 + TK && (!conv || tap == 4);

    // staging: piece = 8 consecutive columns; YP / XP pieces per row, YROWS / XROWS rows per pass of the block
    const int yrow = tid / Cfg::YP, ypc = tid - yrow * Cfg::YP;
    const int xrow = tid / Cfg::XP, xpc = tid - xrow * Cfg::XP;
    const bool y_ok = n0 + ypc * 8 < Np, x_ok = k0 + xpc * 8 < Kp;
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(dY + (size_t)m_begin * ldy + n0, rows > 0 ? ((size_t)(rows - 1) * ldy + (Np - n0)) * sizeof(bf16_t) : 0);
    const __amdgpu_buffer_rsrc_t rs_x = conv ? make_rsrc(X, ((size_t)(M - 1) * ldx + Kp) * sizeof(bf16_t))
                                             : make_rsrc(X + (size_t)m_begin * ldx, rows > 0 ? ((size_t)(rows - 1) * ldx + Kp) * sizeof(bf16_t) : 0);
    const unsigned y_off = (unsigned)(((size_t)yrow * ldy + ypc * 8) * sizeof(bf16_t));
    const unsigned x_off = (unsigned)(((size_t)xrow * ldx + k0 + xpc * 8) * sizeof(bf16_t));
    const unsigned y_step = (unsigned)(Cfg::YROWS * ldy * sizeof(bf16_t)), x_step = (unsigned)(Cfg::XROWS * ldx * sizeof(bf16_t));

    // two stages of both operands are in flight in registers (a CU needs ~40 KB on the way to keep its share of the
    // HBM stream going, and one stage of one workgroup is 49 KB at best): stage st + 2 is requested as soon as stage
    // st has gone to LDS
    uint4 yrA[Cfg::YN], xrA[Cfg::XN], yrB[Cfg::YN], xrB[Cfg::XN];
    auto load_stage = [&](int st, uint4 (&yr)[Cfg::YN], uint4 (&xr)[Cfg::XN]) {
        const unsigned ys = (unsigned)((size_t)st * WG_BM * ldy * sizeof(bf16_t)), xs = (unsigned)((size_t)st * WG_BM * ldx * sizeof(bf16_t));
#pragma unroll
        for (int i = 0; i < Cfg::YN; ++i) yr[i] = buf_load16(rs_y, y_off + i * y_step, ys);
#pragma unroll
        for (int i = 0; i < Cfg::XN; ++i) {
            if (conv) {
                const int m = m_begin + st * WG_BM + xrow + Cfg::XROWS * i;
                const int xw = m % convW, yh = (m / convW) % convH;
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                const bool in = m < M && yh + dy >= 0 && yh + dy < convH && xw + dx >= 0 && xw + dx < convW;
                xr[i] = buf_load16(rs_x, in ? (unsigned)(((size_t)(m + shift) * ldx + k0 + xpc * 8) * sizeof(bf16_t)) : 0x80000000u, 0u);
            } else {
                xr[i] = buf_load16(rs_x, x_off + i * x_step, xs);
            }
        }
    };
    auto write_stage = [&](int st, const uint4 (&yr)[Cfg::YN], const uint4 (&xr)[Cfg::XN]) {
#pragma unroll
        for (int i = 0; i < Cfg::YN; ++i) {
            uint4 y = yr[i];
            if (!y_ok) y = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4 *>(Ys + (yrow + Cfg::YROWS * i) * LSY + ypc * 8) = y;
        }
#pragma unroll
        for (int i = 0; i < Cfg::XN; ++i) {
            const int r = xrow + Cfg::XROWS * i;
            uint4 x = xr[i];
            if (!x_ok) x = make_uint4(0u, 0u, 0u, 0u);
            if (ones && xpc == ((K - k0) >> 3) && st * WG_BM + r < rows) {
                // element K & 7 of this piece := 1.0 (K % 4 == 0: element 0 or 4)
                if (K & 4) x.z = (x.z & 0xffff0000u) | 0x3f80u;
                else x.x = (x.x & 0xffff0000u) | 0x3f80u;
            }
            *reinterpret_cast<uint4 *>(Xs + r * LSX + xpc * 8) = x;
        }
    };

    f32x16 acc[AT][BT];
#pragma unroll
    for (int a = 0; a < AT; ++a)
#pragma unroll
        for (int b = 0; b < BT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transposing reads: slot j of a lane's 8 reduction indices is row 8 (j >> 2) + 4 hh + (j & 3) of the
    // 16-row step, the same for both operands
    const int trq = (lane >> 2) & 3, trp = lane & 3, trg = (lane >> 4) & 1;
    const bf16_t *ytr = Ys + (4 * hh + trq) * LSY + wn * (32 * AT) + 16 * trg + 4 * trp;
    const bf16_t *xtr = Xs + (4 * hh + trq) * LSX + wk * (32 * BT) + 16 * trg + 4 * trp;

    auto products = [&]() {
#pragma unroll
        for (int ks = 0; ks < WG_BM / 16; ++ks) {
            bf16x8 a[AT], b[BT];
#pragma unroll
            for (int t = 0; t < AT; ++t) {
                const uint2 r0 = lds_read_tr16(ytr + (16 * ks) * LSY + 32 * t), r1 = lds_read_tr16(ytr + (16 * ks + 8) * LSY + 32 * t);
                a[t] = as_bf16x8(make_uint4(r0.x, r0.y, r1.x, r1.y));
            }
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                const uint2 r0 = lds_read_tr16(xtr + (16 * ks) * LSX + 32 * t), r1 = lds_read_tr16(xtr + (16 * ks + 8) * LSX + 32 * t);
                b[t] = as_bf16x8(make_uint4(r0.x, r0.y, r1.x, r1.y));
            }
#pragma unroll
            for (int ta = 0; ta < AT; ++ta)
#pragma unroll
                for (int tb = 0; tb < BT; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
    };
    if (nst > 0) load_stage(0, yrA, xrA);
    if (nst > 1) load_stage(1, yrB, xrB);
    for (int st = 0; st < nst; st += 2) {
        if (st > 0) lds_barrier();  // the previous stage's reads are over
        write_stage(st, yrA, xrA);
        lds_barrier();
        if (st + 2 < nst) load_stage(st + 2, yrA, xrA);
        products();
        if (st + 1 < nst) {
            lds_barrier();
            write_stage(st + 1, yrB, xrB);
            lds_barrier();
            if (st + 3 < nst) load_stage(st + 3, yrB, xrB);
            products();
        }
    }
    // lane = output column k, registers = rows n: 128-byte runs per row
    const int prl = (conv ? 9 : 1) * Kpr;  // partial row length
    float *pb = partial + (size_t)by * Np * prl + tap * Kpr + k0;
#pragma unroll
    for (int ta = 0; ta < AT; ++ta)
#pragma unroll
        for (int tb = 0; tb < BT; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * (32 * AT) + ta * 32 + rho(r) + 4 * hh;
                if (n < Np) pb[(size_t)n * prl + wk * (32 * BT) + tb * 32 + l31] = acc[ta][tb][r];
            }
}

// Weight gradient of a 3x3 convolution with at most 64 input channels (rows of X padded to 64): the layers of a reconstruction tail
// (DSwinT_arch.py:430-437: 64 -> 256 in front of a PixelShuffle, 64 -> 64, 64 -> 3) run over millions of pixels with
// tiny outputs, and a 192 x 192 block per tap uses a ninth of its MFMAs there and reads dY nine times.  Here a workgroup
// owns dW[64 output channels][9 taps x 64 input channels] - the same 36 accumulator tiles on 6 waves - and walks its pixel range in
// stages of 64 pixels of one image row (W % 64 == 0): dY's tile and a three-row strip of X (pixels x0 - 1 .. x0 + 64 of
// the rows y - 1, y, y + 1, zero outside the image) go through LDS once, and a tap's operand is the strip read at the
// tap's row and column offset.  Partials in the layout of linear_wgrad_kernel's convolution mode with Kpr = 64.
constexpr int CS_THREADS = 384, CS_LSY = 96, CS_LSX = 96, CS_SROWS = 3 * 66;
constexpr int CS_YP = 64 * 8, CS_XP = CS_SROWS * 8;                  // 16-byte pieces of dY's tile / of the strip
constexpr int CS_YN = (CS_YP + CS_THREADS - 1) / CS_THREADS, CS_XN = (CS_XP + CS_THREADS - 1) / CS_THREADS;   // per thread: 2, 5
constexpr size_t CS_LDS = (size_t)(64 * CS_LSY + CS_SROWS * CS_LSX) * sizeof(bf16_t);
__global__ __launch_bounds__(CS_THREADS) void conv_wgrad_strip_kernel(const bf16_t *__restrict__ dY, long long ldy, const bf16_t *__restrict__ X,
                                                                      long long ldx, int M, int K, int Np, int rows_per_block, int convH,
                                                                      int convW, float *__restrict__ partial, int gx) {
    const int bid = xcd_remap(blockIdx.x, gridDim.x);   // the output-channel blocks of a pixel range on one XCD: they share the strip
    const int bx = bid % gx, by = bid / gx;
    extern __shared__ __align__(16) unsigned char smem[];
    bf16_t *Ys = reinterpret_cast<bf16_t *>(smem);
    bf16_t *Xs = Ys + 64 * CS_LSY;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int n0 = bx * 64, m_begin = by * rows_per_block;
    const int rows = max(0, min(M - m_begin, rows_per_block));
    const int nst = (rows + 63) / 64;
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(dY + (size_t)m_begin * ldy + n0, rows > 0 ? ((size_t)(rows - 1) * ldy + (Np - n0)) * sizeof(bf16_t) : 0);
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(X, ((size_t)(M - 1) * ldx + 64) * sizeof(bf16_t));

    uint4 yrA[CS_YN], xrA[CS_XN], yrB[CS_YN], xrB[CS_XN];
    auto load_stage = [&](int st, uint4 (&yr)[CS_YN], uint4 (&xr)[CS_XN]) {
        const unsigned ys = (unsigned)((size_t)st * 64 * ldy * sizeof(bf16_t));
#pragma unroll
        for (int i = 0; i < CS_YN; ++i) {
            const int p = tid + CS_THREADS * i;
            yr[i] = p < CS_YP ? buf_load16(rs_y, (unsigned)(((size_t)(p >> 3) * ldy + (p & 7) * 8) * sizeof(bf16_t)), ys) : make_uint4(0u, 0u, 0u, 0u);
        }
        const int ms = m_begin + st * 64;               // the stage's first pixel; all 64 lie in one image row
        const int xw0 = ms % convW, yh = (ms / convW) % convH;
#pragma unroll
        for (int i = 0; i < CS_XN; ++i) {
            const int p = tid + CS_THREADS * i, srow = p >> 3, dyr = srow / 66, pos = srow - dyr * 66;
            const int yy = yh + dyr - 1, xx = xw0 + pos - 1;
            const bool in = p < CS_XP && yy >= 0 && yy < convH && xx >= 0 && xx < convW;
            const long long src = (long long)ms + (long long)(dyr - 1) * convW + (pos - 1);
            xr[i] = buf_load16(rs_x, in ? (unsigned)(((size_t)src * ldx + (p & 7) * 8) * sizeof(bf16_t)) : 0x80000000u, 0u);
        }
    };
    auto write_stage = [&](const uint4 (&yr)[CS_YN], const uint4 (&xr)[CS_XN]) {
#pragma unroll
        for (int i = 0; i < CS_YN; ++i) {
            const int p = tid + CS_THREADS * i;
            if (p < CS_YP) *reinterpret_cast<uint4 *>(Ys + (p >> 3) * CS_LSY + (p & 7) * 8) = yr[i];
        }
#pragma unroll
        for (int i = 0; i < CS_XN; ++i) {
            const int p = tid + CS_THREADS * i;
            uint4 x = xr[i];
            // K < 64: column K of the strip's middle row holds ones, which makes column K of the centre tap's result the
            // bias gradient (K % 4 == 0: element 0 or 4 of its piece); the other taps' column K is not read by the sum
            if (K < 64 && (p >> 3) / 66 == 1 && (p & 7) == (K >> 3)) {
                if (K & 4) x.z = (x.z & 0xffff0000u) | 0x3f80u;
                else x.x = (x.x & 0xffff0000u) | 0x3f80u;
            }
            if (p < CS_XP) *reinterpret_cast<uint4 *>(Xs + (p >> 3) * CS_LSX + (p & 7) * 8) = x;
        }
    };

    f32x16 acc[2][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transposing reads as in linear_wgrad_kernel; wave `wave` owns the column tiles 3 wave .. 3 wave + 2 of the 18
    // (tile j = tap j / 2, input channels 32 (j % 2) ..)
    const int trq = (lane >> 2) & 3, trp = lane & 3, trg = (lane >> 4) & 1;
    const bf16_t *ytr = Ys + (4 * hh + trq) * CS_LSY + 16 * trg + 4 * trp;
    const bf16_t *xtr[3];
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
        const int j = 3 * wave + tb, tap = j >> 1, half = j & 1;
        xtr[tb] = Xs + ((tap / 3) * 66 + 1 + (tap % 3 - 1) + 4 * hh + trq) * CS_LSX + half * 32 + 16 * trg + 4 * trp;
    }
    auto products = [&]() {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[2], b[3];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint2 r0 = lds_read_tr16(ytr + (16 * ks) * CS_LSY + 32 * t), r1 = lds_read_tr16(ytr + (16 * ks + 8) * CS_LSY + 32 * t);
                a[t] = as_bf16x8(make_uint4(r0.x, r0.y, r1.x, r1.y));
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const uint2 r0 = lds_read_tr16(xtr[t] + (16 * ks) * CS_LSX), r1 = lds_read_tr16(xtr[t] + (16 * ks + 8) * CS_LSX);
                b[t] = as_bf16x8(make_uint4(r0.x, r0.y, r1.x, r1.y));
            }
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 3; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
    };
    if (nst > 0) load_stage(0, yrA, xrA);
    if (nst > 1) load_stage(1, yrB, xrB);
    for (int st = 0; st < nst; st += 2) {
        if (st > 0) lds_barrier();
        write_stage(yrA, xrA);
        lds_barrier();
        if (st + 2 < nst) load_stage(st + 2, yrA, xrA);
        products();
        if (st + 1 < nst) {
            lds_barrier();
            write_stage(yrB, xrB);
            lds_barrier();
            if (st + 3 < nst) load_stage(st + 3, yrB, xrB);
            products();
        }
    }
    float *pb = partial + (size_t)by * Np * 576;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
            const int j = 3 * wave + tb;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + ta * 32 + rho(r) + 4 * hh;
                if (n < Np) pb[(size_t)n * 576 + (j >> 1) * 64 + (j & 1) * 32 + l31] = acc[ta][tb][r];
            }
        }
}

// dw[n][t][k] = sum_s partial[s][n][t * Kpr + k] (n < N, k < K, t < ntap); db[n] = the same sum at column K of
// tap ntap / 2 (the only tap, or the centre one).  A workgroup sums 16 groups of 4 columns, its 16 thread
// slices taking every 16th partial block; the slices meet in LDS and are added in slice order (a fixed tree:
// bitwise reproducible).
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float *__restrict__ partial, int S, int Np, int Kpr, int ntap, int N,
                                                                  int K, float *__restrict__ dw, float *__restrict__ db) {
    __shared__ float4 red[16][16];
    const int it = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int prl = ntap * Kpr;
    const int i = blockIdx.x * 16 + it;  // (n, tap, group of 4 columns)
    const int n = i / (prl / 4), c4 = (i - n * (prl / 4)) * 4, tap = c4 / Kpr, k4 = c4 - tap * Kpr;
    const bool live = n < N && (k4 < K || (k4 == K && db != nullptr && tap == ntap / 2));
    float4 a = f4zero();
    if (live) {
        const float4 *p = reinterpret_cast<const float4 *>(partial + (size_t)n * prl + c4);
        const size_t stride = (size_t)Np * prl / 4;
        for (int s0 = sl; s0 < S; s0 += 16) a = a + p[(size_t)s0 * stride];
    }
    red[sl][it] = a;
    __syncthreads();
    if (sl == 0 && live) {
#pragma unroll
        for (int j = 1; j < 16; ++j) a = a + red[j][it];
        if (k4 < K) *reinterpret_cast<float4 *>(dw + ((size_t)n * ntap + tap) * K + k4) = a;
        else db[n] = a.x;
    }
}

int set_lds(const void *kern, size_t lds, const char *who) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DSW_ELAUNCH, "%s: LDS attribute: %s", who, hipGetErrorString(e));
    }
    return DSW_OK;
}

template <typename TA, typename TC, int KS>
int launch_linear(const void *a, long long lda, int M, int K, const void *wp, int ldw, int wrows, int nslab, const float *bias, int N,
                  void *c, long long ldc, int ncols, void *a_copy, long long lda_copy, const float *resid, long long ld_resid,
                  const float *rscale, int convH, int convW, int gated, const int *live_units, int unit_rows, hipStream_t st,
                  void *aux = nullptr, long long ld_aux = 0, float norm_eps = -1.f, DropArgs dropa = DropArgs{0u, 0u, 0u, 1.f},
                  int drop_C = 0) {
    using cfg = LnCfg<KS>;
    const int mt = (M + LN_BM - 1) / LN_BM, total = (ncols + LN_CN - 1) / LN_CN;
    int cpb;
    // slabs of <= 128 leave registers for a fourth persistent chunk
    constexpr int NACC = KS <= 128 ? LN_MAXACC_NARROW : LN_MAXACC;
    if (nslab > 1) {
        cpb = total < NACC ? total : NACC;
    } else {
        // enough workgroups for two per CU; every extra block column re-reads A (from L2 at best)
        int gy = (512 + mt - 1) / mt;
        gy = gy < 1 ? 1 : (gy > total ? total : gy);
        cpb = (total + gy - 1) / gy;
    }
    const dim3 grid(mt, (total + cpb - 1) / cpb);
    const size_t lds = cfg::LDS + ((size_t)cpb * LN_CN + LN_BM) * sizeof(float);
    if (lds > 160 * 1024) return fail(DSW_EUNSUPPORTED, "dsw_linear: %d output columns per block need %zu bytes of LDS", cpb * LN_CN, lds);
    int rc;
    const bool ext = gated != 0 || norm_eps >= 0.f, conv = convH > 0;
    const size_t lds_k = lds + (conv ? 2 * (size_t)cfg::ALS * sizeof(bf16_t) : 0);   // the convolution's strip: two more rows of A
    auto go = [&](auto kern) -> int {
        if ((rc = set_lds((const void *)kern, lds_k, "dsw_linear"))) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(LN_THREADS), lds_k, st, (const TA *)a, lda, M, K, (const bf16_t *)wp, ldw, wrows, nslab,
                           bias, N, (TC *)c, ldc, ncols, (bf16_t *)a_copy, lda_copy, cpb, resid, ld_resid, rscale, convH, convW, gated, live_units,
                           unit_rows, (bf16_t *)aux, ld_aux, norm_eps, dropa, drop_C);
        return DSW_OK;
    };
    const bool dropm = dropa.thr != 0u && gated < 2;   // (the GELU forms carry the mask of their fp32 operand themselves)
    if (ext && sizeof(TC) != 2) return fail(DSW_EUNSUPPORTED, "dsw_linear: the activation / normalisation forms write bf16");
    if (conv && (ext || nslab < 9 || cfg::AH > 1)) return fail(DSW_EINVAL, "dsw_conv3x3_nhwc: nine taps of whole slabs, no activation forms");
    if (nslab > 1) {
        if constexpr (cfg::AH > 1) {
            return fail(DSW_EUNSUPPORTED, "dsw_linear: slabs of %d columns are single", KS);
        } else if constexpr (sizeof(TC) == 2) {
            if ((rc = conv ? go(linear_nt_kernel<TA, TC, KS, NACC, 2>)
                      : gated >= 2 ? go(linear_nt_kernel<TA, TC, KS, NACC, 4>)
                      : dropm ? go(linear_nt_kernel<TA, TC, KS, NACC, 5>) : go(linear_nt_kernel<TA, TC, KS, NACC, 0>)))
                return rc;
        } else {
            if ((rc = conv ? go(linear_nt_kernel<TA, TC, KS, NACC, 2>)
                      : dropm ? go(linear_nt_kernel<TA, TC, KS, NACC, 5>) : go(linear_nt_kernel<TA, TC, KS, NACC, 0>)))
                return rc;
        }
    } else {
        if constexpr (sizeof(TC) == 2) {
            if ((rc = gated == 1 ? go(linear_nt_kernel<TA, TC, KS, 0, 1>)
                      : gated >= 2 ? go(linear_nt_kernel<TA, TC, KS, 0, 4>)
                      : ext ? go(linear_nt_kernel<TA, TC, KS, 0, 3>)
                      : dropm ? go(linear_nt_kernel<TA, TC, KS, 0, 5>) : go(linear_nt_kernel<TA, TC, KS, 0, 0>)))
                return rc;
        } else {
            if ((rc = dropm ? go(linear_nt_kernel<TA, TC, KS, 0, 5>) : go(linear_nt_kernel<TA, TC, KS, 0, 0>))) return rc;
        }
    }
    return check_launch("linear_nt_kernel");
}

}  // namespace
}  // namespace dsw

using namespace dsw;

extern "C" int dsw_weight_prep(const void *w, int w_dtype, int N, int K, void *w_p, void *w_pt, void *stream) {
    if (N <= 0 || K <= 0) return fail(DSW_EINVAL, "dsw_weight_prep: bad sizes N=%d K=%d", N, K);
    DSW_REQUIRE(w && (w_p || w_pt), "dsw_weight_prep: null pointer");
    const int Np = (N + 63) / 64 * 64, Kp = (K + 63) / 64 * 64;
    DSW_DISPATCH_DTYPE(w_dtype, TW, {
        hipLaunchKernelGGL(weight_prep_kernel<TW>, dim3((Np * Kp + 255) / 256), dim3(256), 0, as_stream(stream), (const TW *)w, N, K, Np,
                           Kp, (bf16_t *)w_p, (bf16_t *)w_pt);
    });
    return check_launch("weight_prep_kernel");
}

extern "C" int dsw_weight_prep_batch(int n, const void *const *w, const int *w_dtype, const int *N, const int *K, void *const *w_p,
                                     void *const *w_pt, void *stream) {
    if (n < 0) return fail(DSW_EINVAL, "dsw_weight_prep_batch: n=%d", n);
    DSW_REQUIRE(n == 0 || (w && w_dtype && N && K && w_p && w_pt), "dsw_weight_prep_batch: null pointer");
    for (int i = 0; i < n; ++i) {
        if (N[i] <= 0 || K[i] <= 0 || (long long)((N[i] + 63) / 64 * 64) * ((K[i] + 63) / 64 * 64) > 0x7fffffffLL)
            return fail(DSW_EINVAL, "dsw_weight_prep_batch: bad sizes N=%d K=%d (weight %d)", N[i], K[i], i);
        if (w_dtype[i] != DSW_F32 && w_dtype[i] != DSW_BF16) return fail(DSW_EUNSUPPORTED, "dsw_weight_prep_batch: weight %d: fp32 or bf16", i);
        DSW_REQUIRE(w[i] && (w_p[i] || w_pt[i]), "dsw_weight_prep_batch: null pointer (weight %d)", i);
    }
    for (int i0 = 0; i0 < n; i0 += WP_BATCH) {
        WpBatch b{};
        b.n = n - i0 < WP_BATCH ? n - i0 : WP_BATCH;
        int blocks = 0;
        for (int j = 0; j < b.n; ++j) {
            const int i = i0 + j;
            b.w[j] = w[i];
            b.wp[j] = (bf16_t *)w_p[i];
            b.wpt[j] = (bf16_t *)w_pt[i];
            b.N[j] = N[i];
            b.K[j] = K[i];
            if (w_dtype[i] == DSW_BF16) b.is_bf16 |= 1ull << j;
            b.first_block[j] = blocks;
            blocks += (((N[i] + 63) / 64 * 64) * ((K[i] + 63) / 64 * 64) + 255) / 256;
        }
        b.first_block[b.n] = blocks;
        hipLaunchKernelGGL(weight_prep_batch_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), b);
        int rc = check_launch("weight_prep_batch_kernel");
        if (rc) return rc;
    }
    return DSW_OK;
}

static int linear_impl(const void *a, int a_dtype, long long lda, long long M, int K, const void *w_p, int wrows, int ldw,
                       const float *bias, int N, void *c, int c_dtype, long long ldc, int ncols, int accumulate, void *a_copy,
                       long long lda_copy, const float *resid, long long ld_resid, const float *rscale, void *stream, int gated = 0,
                       const int *live_units = nullptr, int unit_rows = 0, void *aux = nullptr, long long ld_aux = 0, float norm_eps = -1.f,
                       float drop_p = 0.f, unsigned long long drop_seed = 0ull, int drop_C = 0) {
    DropArgs dropa{0u, 0u, 0u, 1.f};
    if (drop_p != 0.f) {
        // the masked side is the fp32 one: the operand (a gradient on its way in) or the result in front of its residual
        const int rcd = make_drop("dsw_linear (dropout)", drop_p, drop_seed, dropa);
        if (rcd) return rcd;
        const bool in = a_dtype == DSW_F32 && resid == nullptr, out = c_dtype == DSW_F32 && resid != nullptr && !accumulate;
        if (in == out || drop_C <= 0 || drop_C % 4 || norm_eps >= 0.f || gated == 1 || live_units != nullptr || (in && drop_C != K))
            return fail(DSW_EUNSUPPORTED, "dsw_linear (dropout): an fp32 operand of dense rows (no residual) or an fp32 result with a "
                                          "residual, a dense row length that is a multiple of 4 (drop_C=%d K=%d)", drop_C, K);
    }
    if (M <= 0 || M > 0x7fffff00LL || K <= 0 || ncols <= 0 || wrows <= 0)
        return fail(DSW_EINVAL, "dsw_linear: bad sizes M=%lld K=%d ncols=%d wrows=%d", M, K, ncols, wrows);
    DSW_REQUIRE(a && w_p && c, "dsw_linear: null pointer");
    const int Kp = (K + 63) / 64 * 64;
    if (K % 4 || lda < K || lda % 4 || ((uintptr_t)a & (a_dtype == DSW_F32 ? 15 : 7)))
        return fail(DSW_EUNSUPPORTED, "dsw_linear: A needs K and its row stride to be multiples of 4 elements, the stride >= K, and a "
                                      "base aligned to 4 elements (K=%d lda=%lld)", K, lda);
    if (ldw < Kp || ldw % 8 || ((uintptr_t)w_p & 15))
        return fail(DSW_EUNSUPPORTED, "dsw_linear: the weight needs rows of >= %d elements (K rounded up to 64), a multiple of 8 (ldw=%d)", Kp, ldw);
    if (ncols % 4 || ldc < (gated == 1 ? ncols / 2 : ncols) || ldc % 4 || ((uintptr_t)c & 15))
        return fail(DSW_EUNSUPPORTED, "dsw_linear: C needs ncols %% 4 == 0, ldc %% 4 == 0, ldc >= ncols, a 16-byte aligned base (ncols=%d ldc=%lld)",
                    ncols, ldc);
    if (bias != nullptr && (N <= 0 || N % 4 || N > ncols || ((uintptr_t)bias & 15)))
        return fail(DSW_EUNSUPPORTED, "dsw_linear: bias needs N %% 4 == 0, N <= ncols, a 16-byte aligned base (N=%d ncols=%d)", N, ncols);
    if (a_copy != nullptr && (lda_copy < Kp || lda_copy % 8 || ((uintptr_t)a_copy & 15)))
        return fail(DSW_EUNSUPPORTED, "dsw_linear: the bf16 copy of A needs rows of >= %d elements, a multiple of 8 (lda_copy=%lld)", Kp, lda_copy);
    if ((accumulate || resid != nullptr) && c_dtype != DSW_F32) return fail(DSW_EUNSUPPORTED, "dsw_linear: accumulation needs an fp32 result");
    if (accumulate) {
        resid = static_cast<const float *>(c);
        ld_resid = ldc;
        rscale = nullptr;
    } else if (resid != nullptr && (ld_resid < ncols || ld_resid % 4 || ((uintptr_t)resid & 15)))
        return fail(DSW_EUNSUPPORTED, "dsw_linear_residual: the residual needs a row stride >= ncols, a multiple of 4, and a 16-byte aligned base "
                                      "(ld_resid=%lld)", ld_resid);
    // k slab: the widest that divides the padded K; a K beyond 192 that only 64 divides (msft: 448, 704) takes slabs
    // of 128 with the last one half empty (A is masked at K, so what the weight rows hold there does not matter:
    // the next row's finite values, or zero past the end of the matrix) rather than twice as many slabs of 64,
    // each with its own round of barriers
    // (a K of 193..256 is one slab of 256 whose rows cross LDS in two halves: no accumulators carried over slabs, every
    // chunk stored as soon as it is complete)
    static const bool no256 = getenv("DSW_LINEAR_NO256") != nullptr;   // experiments: the two-slab path for K of 193..256
    const int KS = Kp % 192 == 0 ? 192 : (Kp == 256 && !no256) ? 256 : (Kp % 128 == 0 || Kp > 192 ? 128 : 64);
    const int nslab = (Kp + KS - 1) / KS;
    if (norm_eps >= 0.f && (nslab != 1 || KS > 192 || live_units != nullptr))
        return fail(DSW_EUNSUPPORTED, "dsw_linear_rmsnorm: the normalised rows must fit one slab of K <= 192, no live-row counter (K=%d)", K);
    if (gated >= 2 && norm_eps >= 0.f) return fail(DSW_EUNSUPPORTED, "dsw_linear: the GELU epilogues do not take the normalisation");
    if (gated >= 2 && (c_dtype != DSW_BF16 || ncols % 8 || ldc % 8 || resid != nullptr || accumulate || live_units != nullptr ||
                       (gated == 3 && aux == nullptr) || (aux != nullptr && (ld_aux < ncols || ld_aux % 8 || ((uintptr_t)aux & 15)))))
        return fail(DSW_EUNSUPPORTED, "dsw_linear_gelu: needs a bf16 result, ncols and the row strides multiples of 8, 16-byte aligned bases "
                                      "(ncols=%d ldc=%lld ld_aux=%lld)", ncols, ldc, ld_aux);
    if (gated == 1 && (nslab != 1 || c_dtype != DSW_BF16 || ncols % 64 || ldc % 8 || a_copy != nullptr || resid != nullptr || accumulate))
        return fail(DSW_EUNSUPPORTED, "dsw_linear_gated_silu: needs K <= 192 (or 128 / 64 exactly padded), a bf16 result with rows of a multiple of "
                                      "8 elements, whole chunks of 64 weight rows (K=%d rows=%d ldo=%lld)", K, ncols, ldc);
    if (live_units != nullptr && (unit_rows <= 0 || (resid != nullptr && !accumulate)))
        return fail(DSW_EINVAL, "dsw_linear: live_units needs unit_rows > 0 and no separate residual (unit_rows=%d)", unit_rows);
    hipStream_t st = as_stream(stream);
    int rc = DSW_OK;
#define DSW_LINEAR_KS(KSV)                                                                                                     \
    DSW_DISPATCH_DTYPE(a_dtype, TA, DSW_DISPATCH_DTYPE(c_dtype, TC, {                                                          \
        rc = launch_linear<TA, TC, KSV>(a, lda, (int)M, K, w_p, ldw, wrows, nslab, bias, N, c, ldc, ncols, a_copy, lda_copy, resid, ld_resid, rscale, 0, 0, gated, live_units, unit_rows, st, aux, ld_aux, norm_eps, dropa, drop_C); \
    }))
    if (KS == 192) DSW_LINEAR_KS(192);
    else if (KS == 256) DSW_LINEAR_KS(256);
    else if (KS == 128) DSW_LINEAR_KS(128);
    else DSW_LINEAR_KS(64);
#undef DSW_LINEAR_KS
    return rc;
}

extern "C" int dsw_linear(const void *a, int a_dtype, long long lda, long long M, int K, const void *w_p, int wrows, int ldw,
                          const float *bias, int N, void *c, int c_dtype, long long ldc, int ncols, int accumulate, void *a_copy,
                          long long lda_copy, const int *live_units, int unit_rows, void *stream) {
    return linear_impl(a, a_dtype, lda, M, K, w_p, wrows, ldw, bias, N, c, c_dtype, ldc, ncols, accumulate, a_copy, lda_copy, nullptr, 0,
                       nullptr, stream, 0, live_units, unit_rows);
}

extern "C" int dsw_linear_residual(const void *a, int a_dtype, long long lda, long long M, int K, const void *w_p, int wrows, int ldw,
                                   const float *bias, int N, const float *resid, long long ld_resid, const float *scale, float *out,
                                   long long ldo, int ncols, void *stream) {
    DSW_REQUIRE(resid && scale, "dsw_linear_residual: null pointer");
    return linear_impl(a, a_dtype, lda, M, K, w_p, wrows, ldw, bias, N, out, DSW_F32, ldo, ncols, 0, nullptr, 0, resid, ld_resid, scale,
                       stream);
}

extern "C" int dsw_linear_gated_silu(const void *a, int a_dtype, long long lda, long long M, int K, const void *w_p, int wrows, int ldw,
                                     const float *bias, void *out, long long ldo, void *stream) {
    return linear_impl(a, a_dtype, lda, M, K, w_p, wrows, ldw, bias, wrows, out, DSW_BF16, ldo, wrows, 0, nullptr, 0, nullptr, 0, nullptr,
                       stream, 1);
}

extern "C" int dsw_linear_rmsnorm(const void *a, int a_dtype, long long lda, long long M, int K, float eps, const void *w_p, int wrows, int ldw,
                                  const float *bias, int N, void *c, int c_dtype, long long ldc, int ncols, void *stream) {
    if (!(eps >= 0.f)) return fail(DSW_EINVAL, "dsw_linear_rmsnorm: eps=%g", (double)eps);
    return linear_impl(a, a_dtype, lda, M, K, w_p, wrows, ldw, bias, N, c, c_dtype, ldc, ncols, 0, nullptr, 0, nullptr, 0, nullptr, stream, 0,
                       nullptr, 0, nullptr, 0, eps);
}

extern "C" int dsw_linear_rmsnorm_gated_silu(const void *a, int a_dtype, long long lda, long long M, int K, float eps, const void *w_p, int wrows,
                                             int ldw, const float *bias, void *out, long long ldo, void *stream) {
    if (!(eps >= 0.f)) return fail(DSW_EINVAL, "dsw_linear_rmsnorm_gated_silu: eps=%g", (double)eps);
    return linear_impl(a, a_dtype, lda, M, K, w_p, wrows, ldw, bias, wrows, out, DSW_BF16, ldo, wrows, 0, nullptr, 0, nullptr, 0, nullptr,
                       stream, 1, nullptr, 0, nullptr, 0, eps);
}

extern "C" int dsw_linear_gelu(const void *a, int a_dtype, long long lda, long long M, int K, const void *w_p, int wrows, int ldw,
                               const float *bias, int N, void *out, long long ldo, int ncols, void *pre, long long ld_pre, void *a_copy,
                               long long lda_copy, void *stream) {
    return linear_impl(a, a_dtype, lda, M, K, w_p, wrows, ldw, bias, N, out, DSW_BF16, ldo, ncols, 0, a_copy, lda_copy, nullptr, 0, nullptr,
                       stream, 2, nullptr, 0, pre, ld_pre);
}

extern "C" int dsw_linear_gelu_bwd(const void *a, int a_dtype, long long lda, long long M, int K, const void *w_p, int wrows, int ldw,
                                   const void *pre, long long ld_pre, void *out, long long ldo, int ncols, void *a_copy, long long lda_copy,
                                   float drop_p, unsigned long long drop_seed, void *stream) {
    DSW_REQUIRE(pre, "dsw_linear_gelu_bwd: null pointer");
    return linear_impl(a, a_dtype, lda, M, K, w_p, wrows, ldw, nullptr, 0, out, DSW_BF16, ldo, ncols, 0, a_copy, lda_copy, nullptr, 0, nullptr,
                       stream, 3, nullptr, 0, const_cast<void *>(pre), ld_pre, -1.f, drop_p, drop_seed, K);
}

extern "C" int dsw_linear_dropout_in(const void *a, long long lda, long long M, int K, const void *w_p, int wrows, int ldw, void *c, int c_dtype,
                                     long long ldc, int ncols, void *a_copy, long long lda_copy, float drop_p, unsigned long long drop_seed,
                                     void *stream) {
    return linear_impl(a, DSW_F32, lda, M, K, w_p, wrows, ldw, nullptr, 0, c, c_dtype, ldc, ncols, 0, a_copy, lda_copy, nullptr, 0, nullptr,
                       stream, 0, nullptr, 0, nullptr, 0, -1.f, drop_p, drop_seed, K);
}

extern "C" int dsw_linear_residual_dropout(const void *a, int a_dtype, long long lda, long long M, int K, const void *w_p, int wrows, int ldw,
                                           const float *bias, int N, const float *resid, long long ld_resid, const float *scale, float *out,
                                           long long ldo, int ncols, float drop_p, unsigned long long drop_seed, void *stream) {
    DSW_REQUIRE(resid && scale, "dsw_linear_residual_dropout: null pointer");
    return linear_impl(a, a_dtype, lda, M, K, w_p, wrows, ldw, bias, N, out, DSW_F32, ldo, ncols, 0, nullptr, 0, resid, ld_resid, scale, stream,
                       0, nullptr, 0, nullptr, 0, -1.f, drop_p, drop_seed, N);
}

static int conv3x3_impl(const void *x, int x_dtype, long long ldx, int B, int H, int W, int C, const void *w_p, int wrows, int ldw,
                        const float *bias, int N, void *out, int out_dtype, long long ldo, int ncols, void *x_copy,
                        long long ld_copy, const float *resid, long long ld_resid, void *stream) {
    const long long M = (long long)B * H * W;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || ncols <= 0 || wrows <= 0)
        return fail(DSW_EINVAL, "dsw_conv3x3_nhwc: bad sizes B=%d H=%d W=%d C=%d ncols=%d", B, H, W, C, ncols);
    DSW_REQUIRE(x && w_p && out, "dsw_conv3x3_nhwc: null pointer");
    const int Kp = (C + 63) / 64 * 64;
    const size_t xs = x_dtype == DSW_F32 ? 4 : 2;
    if (M > 0x7fffff00LL || (size_t)M * ldx * xs >= 0x80000000ull)
        return fail(DSW_EUNSUPPORTED, "dsw_conv3x3_nhwc: the input must stay below 2 GiB (%lld pixels x %lld)", M, ldx);
    if (C % 4 || ldx < C || ldx % 4 || ((uintptr_t)x & (x_dtype == DSW_F32 ? 15 : 7)))
        return fail(DSW_EUNSUPPORTED, "dsw_conv3x3_nhwc: C and the pixel stride must be multiples of 4, the base aligned to 4 elements");
    if (ldw < 9 * Kp || ldw % 8 || ((uintptr_t)w_p & 15))
        return fail(DSW_EUNSUPPORTED, "dsw_conv3x3_nhwc: the weight needs rows of 9 x %d elements (tap-major, channels padded to 64)", Kp);
    if (ncols % 4 || ldo < ncols || ldo % 4 || ((uintptr_t)out & 15))
        return fail(DSW_EUNSUPPORTED, "dsw_conv3x3_nhwc: ncols and the output stride must be multiples of 4, the base 16-byte aligned");
    if (bias != nullptr && (N <= 0 || N % 4 || N > ncols || ((uintptr_t)bias & 15)))
        return fail(DSW_EUNSUPPORTED, "dsw_conv3x3_nhwc: bias needs N %% 4 == 0, N <= ncols, a 16-byte aligned base");
    if (x_copy != nullptr && (ld_copy < Kp || ld_copy % 8 || ((uintptr_t)x_copy & 15)))
        return fail(DSW_EUNSUPPORTED, "dsw_conv3x3_nhwc: the bf16 copy of x needs rows of >= %d elements, a multiple of 8", Kp);
    const int KS = Kp % 192 == 0 ? 192 : (Kp % 128 == 0 ? 128 : 64);
    const int nslab = 9 * (Kp / KS);
    hipStream_t st = as_stream(stream);
    int rc = DSW_OK;
#define DSW_CONV_KS(KSV)                                                                                                          \
    DSW_DISPATCH_DTYPE(x_dtype, TA, DSW_DISPATCH_DTYPE(out_dtype, TC, {                                                           \
        rc = launch_linear<TA, TC, KSV>(x, ldx, (int)M, C, w_p, ldw, wrows, nslab, bias, N, out, ldo, ncols, x_copy, ld_copy, resid, ld_resid, nullptr, H, W, 0, nullptr, 0, st); \
    }))
    if (KS == 192) DSW_CONV_KS(192);
    else if (KS == 128) DSW_CONV_KS(128);
    else DSW_CONV_KS(64);
#undef DSW_CONV_KS
    return rc;
}

extern "C" int dsw_conv3x3_nhwc(const void *x, int x_dtype, long long ldx, int B, int H, int W, int C, const void *w_p, int wrows, int ldw,
                                const float *bias, int N, void *out, int out_dtype, long long ldo, int ncols, void *x_copy,
                                long long ld_copy, void *stream) {
    return conv3x3_impl(x, x_dtype, ldx, B, H, W, C, w_p, wrows, ldw, bias, N, out, out_dtype, ldo, ncols, x_copy, ld_copy, nullptr, 0, stream);
}

extern "C" int dsw_conv3x3_nhwc_residual(const void *x, int x_dtype, long long ldx, int B, int H, int W, int C, const void *w_p, int wrows,
                                         int ldw, const float *bias, int N, const float *resid, long long ld_resid, float *out,
                                         long long ldo, int ncols, void *stream) {
    DSW_REQUIRE(resid, "dsw_conv3x3_nhwc_residual: null pointer");
    if (ld_resid < ncols || ld_resid % 4 || ((uintptr_t)resid & 15))
        return fail(DSW_EUNSUPPORTED, "dsw_conv3x3_nhwc_residual: the residual needs rows of >= %d floats, a multiple of 4, 16-byte aligned", ncols);
    return conv3x3_impl(x, x_dtype, ldx, B, H, W, C, w_p, wrows, ldw, bias, N, out, DSW_F32, ldo, ncols, nullptr, 0, resid, ld_resid, stream);
}

extern "C" int dsw_conv3x3_nhwc_residual_copy(const void *x, int x_dtype, long long ldx, int B, int H, int W, int C, const void *w_p,
                                              int wrows, int ldw, const float *bias, int N, const float *resid, long long ld_resid,
                                              float *out, long long ldo, int ncols, void *x_copy, long long ld_copy, void *stream) {
    DSW_REQUIRE(resid, "dsw_conv3x3_nhwc_residual_copy: null pointer");
    if (ld_resid < ncols || ld_resid % 4 || ((uintptr_t)resid & 15))
        return fail(DSW_EUNSUPPORTED, "dsw_conv3x3_nhwc_residual_copy: the residual needs rows of >= %d floats, a multiple of 4, 16-byte aligned", ncols);
    return conv3x3_impl(x, x_dtype, ldx, B, H, W, C, w_p, wrows, ldw, bias, N, out, DSW_F32, ldo, ncols, x_copy, ld_copy, resid, ld_resid, stream);
}

static void wgrad_split(long long M, int Np, int Kp, int ntap, int *rows_per_block, int *S) {
    const int TN = wgrad_wide(Kp) ? WgWide::TN : WgNarrow::TN, TK = wgrad_wide(Kp) ? WgWide::TK : WgNarrow::TK;
    const int nb = ((Np + TN - 1) / TN) * ((Kp + TK - 1) / TK) * ntap;
    static const int target = [] {   // DSW_WGRAD_WGS in the environment: the workgroup count aimed at (experiments)
        const char *e = getenv("DSW_WGRAD_WGS");
        return e != nullptr && atoi(e) > 0 ? atoi(e) : 256;
    }();
    long long want = target / nb;  // about one workgroup per CU: every extra one adds a partial block to write and re-read
    if (want < 8 && ntap > 1) want = 8;  // 9 taps: keep some row ranges so that a block's stream stays short
    if (want < 1) want = 1;
    long long rpb = (M + want - 1) / want;
    rpb = (rpb + WG_BM - 1) / WG_BM * WG_BM;
    *rows_per_block = (int)rpb;
    *S = (int)((M + rpb - 1) / rpb);
}

// the strip kernel takes a 3x3 convolution with at most 64 input channels on images whose width is a multiple of 64
static bool conv_strip(int K, int convW) { return K <= 64 && K % 4 == 0 && convW > 0 && convW % 64 == 0 && getenv("DSW_CONV_WGRAD_NO_STRIP") == nullptr; }
static void conv_strip_split(long long M, int Np, int *rows_per_block, int *S) {
    long long want = 512 / (Np / 64);   // about two workgroups per CU
    if (want < 1) want = 1;
    long long rpb = (M + want - 1) / want;
    rpb = (rpb + 63) / 64 * 64;
    *rows_per_block = (int)rpb;
    *S = (int)((M + rpb - 1) / rpb);
}

static size_t wgrad_ws(long long M, int N, int K, int ntap, int convW = 0) {
    if (M <= 0 || N <= 0 || K <= 0 || M > 0x7fffff00LL) return 0;
    const int Np = (N + 63) / 64 * 64, Kp = (K + 63) / 64 * 64;
    if (ntap == 9 && conv_strip(K, convW)) {
        int rpb, S;
        conv_strip_split(M, Np, &rpb, &S);
        return (size_t)S * Np * 576;
    }
    const int TK = wgrad_wide(Kp) ? WgWide::TK : WgNarrow::TK, Kpr = (Kp + TK - 1) / TK * TK;
    int rpb, S;
    wgrad_split(M, Np, Kp, ntap, &rpb, &S);
    return (size_t)S * Np * Kpr * ntap;
}

extern "C" size_t dsw_linear_wgrad_workspace(long long M, int N, int K) { return wgrad_ws(M, N, K, 1); }
extern "C" size_t dsw_conv3x3_wgrad_workspace(int B, int H, int W, int Co, int Ci) { return wgrad_ws((long long)B * H * W, Co, Ci, 9, W); }

static int wgrad_launch(const char *who, const void *dy, long long ldy, const void *x, long long ldx, long long M, int N, int K, int convH,
                        int convW, float *dw, float *db, float *workspace, void *stream, const int *live_units = nullptr, int unit_rows = 0) {
    if (live_units != nullptr && (unit_rows <= 0 || convH > 0)) return fail(DSW_EINVAL, "%s: live_units needs unit_rows > 0 (unit_rows=%d)", who, unit_rows);
    if (M <= 0 || M > 0x7fffff00LL || N <= 0 || K <= 0) return fail(DSW_EINVAL, "%s: bad sizes M=%lld N=%d K=%d", who, M, N, K);
    DSW_REQUIRE(dy && x && dw && workspace, "%s: null pointer", who);
    const int ntap = convH > 0 ? 9 : 1;
    const int Np = (N + 63) / 64 * 64, Kp = (K + 63) / 64 * 64;
    const bool wide = wgrad_wide(Kp);
    const int TN = wide ? WgWide::TN : WgNarrow::TN, TK = wide ? WgWide::TK : WgNarrow::TK, Kpr = (Kp + TK - 1) / TK * TK;
    if (K % 4) return fail(DSW_EUNSUPPORTED, "%s: K=%d must be a multiple of 4", who, K);
    if (db != nullptr && K == Kp) return fail(DSW_EUNSUPPORTED, "%s: the bias gradient rides in a padding column of x, K=%d leaves none", who, K);
    if (ldy < Np || ldx < Kp || ldy % 8 || ldx % 8 || ((uintptr_t)dy & 15) || ((uintptr_t)x & 15))
        return fail(DSW_EUNSUPPORTED, "%s: dy / x need rows of >= %d / %d bf16 elements (zero padded), multiples of 8, 16-byte aligned "
                                      "(ldy=%lld ldx=%lld)", who, Np, Kp, ldy, ldx);
    if (((uintptr_t)dw & 15) || ((uintptr_t)workspace & 15)) return fail(DSW_EUNSUPPORTED, "%s: dw / workspace must be 16-byte aligned", who);
    int rpb, S;
    if (ntap == 9 && conv_strip(K, convW)) {
        if (db != nullptr && K == 64) return fail(DSW_EUNSUPPORTED, "%s: the bias gradient rides in a padding column of x, K=%d leaves none", who, K);
        conv_strip_split(M, Np, &rpb, &S);
        if ((long long)rpb * ldy * 2 >= 0x7fffffffLL || (size_t)M * ldx * 2 >= 0x80000000ull)
            return fail(DSW_EUNSUPPORTED, "%s: operands beyond 2 GiB per block (%d rows) or per tensor", who, rpb);
        hipStream_t st = as_stream(stream);
        int rc = set_lds((const void *)conv_wgrad_strip_kernel, CS_LDS, who);
        if (rc) return rc;
        const int gx = Np / 64;
        hipLaunchKernelGGL(conv_wgrad_strip_kernel, dim3(gx * S), dim3(CS_THREADS), CS_LDS, st, (const bf16_t *)dy, ldy, (const bf16_t *)x, ldx,
                           (int)M, K, Np, rpb, convH, convW, workspace, gx);
        rc = check_launch("conv_wgrad_strip_kernel");
        if (rc) return rc;
        const int items = N * (9 * 64 / 4);
        hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((items + 15) / 16), dim3(256), 0, st, workspace, S, Np, 64, 9, N, K, dw, db);
        return check_launch("linear_wgrad_reduce_kernel");
    }
    wgrad_split(M, Np, Kp, ntap, &rpb, &S);
    if ((long long)rpb * (ldy > ldx ? ldy : ldx) * 2 >= 0x7fffffffLL || (ntap > 1 && (size_t)M * ldx * 2 >= 0x80000000ull))
        return fail(DSW_EUNSUPPORTED, "%s: operands beyond 2 GiB per block (%d rows) or per tensor", who, rpb);
    hipStream_t st = as_stream(stream);
    const int gx = (Np + TN - 1) / TN, gz = ntap * (Kpr / TK);
    const dim3 grid(gx * S * gz);
    int rc;
    if (wide) {
        rc = set_lds((const void *)linear_wgrad_kernel<WgWide>, WgWide::LDS, who);
        if (rc) return rc;
        hipLaunchKernelGGL(linear_wgrad_kernel<WgWide>, grid, dim3(WgWide::THREADS), WgWide::LDS, st, (const bf16_t *)dy, ldy, (const bf16_t *)x,
                           ldx, (int)M, K, Kp, Np, Kpr, rpb, convH, convW, workspace, live_units, unit_rows, gx, gz);
    } else {
        rc = set_lds((const void *)linear_wgrad_kernel<WgNarrow>, WgNarrow::LDS, who);
        if (rc) return rc;
        hipLaunchKernelGGL(linear_wgrad_kernel<WgNarrow>, grid, dim3(WgNarrow::THREADS), WgNarrow::LDS, st, (const bf16_t *)dy, ldy,
                           (const bf16_t *)x, ldx, (int)M, K, Kp, Np, Kpr, rpb, convH, convW, workspace, live_units, unit_rows, gx, gz);
    }
    rc = check_launch("linear_wgrad_kernel");
    if (rc) return rc;
    const int items = N * (ntap * Kpr / 4);
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((items + 15) / 16), dim3(256), 0, st, workspace, S, Np, Kpr, ntap, N, K, dw, db);
    return check_launch("linear_wgrad_reduce_kernel");
}

extern "C" int dsw_linear_wgrad(const void *dy, long long ldy, const void *x, long long ldx, long long M, int N, int K, float *dw,
                                float *db, float *workspace, const int *live_units, int unit_rows, void *stream) {
    return wgrad_launch("dsw_linear_wgrad", dy, ldy, x, ldx, M, N, K, 0, 0, dw, db, workspace, stream, live_units, unit_rows);
}

extern "C" int dsw_conv3x3_wgrad(const void *dy, long long ldy, const void *x, long long ldx, int B, int H, int W, int Co, int Ci,
                                 float *dw, float *db, float *workspace, void *stream) {
    if (B <= 0 || H <= 0 || W <= 0) return fail(DSW_EINVAL, "dsw_conv3x3_wgrad: bad sizes B=%d H=%d W=%d", B, H, W);
    return wgrad_launch("dsw_conv3x3_wgrad", dy, ldy, x, ldx, (long long)B * H * W, Co, Ci, H, W, dw, db, workspace, stream);
}

#ifdef LINEAR_STAMPS
extern "C" int dsw_debug_read_linear_stamps(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dsw::g_linear_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif
