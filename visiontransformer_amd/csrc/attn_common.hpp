// Shared pieces of the fused window-attention kernels (forward and backward).
#pragma once
#include "common.hpp"

namespace dsw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct AttnGeom {
    int n_win;       // total windows
    int H, W;        // image size in tokens (0,0 = packed windows)
    int nWh, nWw;    // windows per image column / row (grid mode)
    int C, heads, d; // channels, heads, head dim
    int layout;      // DSW_QKV_*
    int ldq, ldo, ldg;  // row strides in elements: qkv / dqkv (>= 3C), out (>= C), dout (>= C)
    const int *n_live;  // device counter or null: packed windows with index >= *n_live are switched off
};

// Attention dropout (DSwinT_arch.py:172, msft_arch.py:136) without a mask tensor: the keep decision of score
// element (window, head, query q, key k) is a function of a 64-bit seed and the element's position that the
// forward and the backward kernels both evaluate (and that a CPU check can evaluate too).  One 32-bit hash
// serves the two adjacent keys 2j, 2j+1 of a query row: pair index = ((window * heads + head) * L + q) * L/2 + j
// (mod 2^32); the even key takes the hash's low 16 bits, the odd key the high 16 bits, and an element is kept
// when its 16 bits are >= thr = round(p * 65536).  Kept probabilities are scaled by 1 / (1 - p).
// (DropArgs and drop_hash are in common.hpp: the residual kernels use the same rule on their elements)

// Row of the C layout of a 32x32 MFMA result held in register r by lane half hh.
__device__ __forceinline__ constexpr int rho(int r) { return (r & 3) + 8 * (r >> 2); }

// token (window, l) -> row of the [rows, 3C] / [rows, C] matrices
template <int WS>
__device__ __forceinline__ size_t token_row(const AttnGeom &g, int win, int l) {
    if (g.H == 0) return (size_t)win * (WS * WS) + l;
    const int per = g.nWh * g.nWw;
    const int b = win / per, r = win - b * per;
    const int wr = r / g.nWw, wc = r - wr * g.nWw;
    const int i = l / WS, j = l - i * WS;
    return ((size_t)b * g.H + wr * WS + i) * g.W + wc * WS + j;
}

__device__ __forceinline__ void qkv_offsets(const AttnGeom &g, int head, int &qoff, int &koff, int &voff) {
    if (g.layout == DSW_QKV_CHUNKED) {
        qoff = head * g.d;
        koff = g.C + head * g.d;
        voff = 2 * g.C + head * g.d;
    } else {
        qoff = head * 3 * g.d;
        koff = qoff + g.d;
        voff = qoff + 2 * g.d;
    }
}

// Switched-off windows (DS-MSA slots behind the windows that passed the gate, counted on the device): no arithmetic;
// the window's rows of the result are set to zero so that later readers find finite values.
__device__ __forceinline__ bool window_is_off(const AttnGeom &g, int win) { return g.n_live != nullptr && win >= *g.n_live; }

// zero the head's d columns starting at column col0 of the window's L rows of a [rows][ld] matrix, and (last head
// only) the padding columns first_pad..ld-1
template <typename T>
__device__ __forceinline__ void zero_window_slice(T *__restrict__ m, int ld, size_t origin, int pitch, int ws, int col0, int d,
                                                  bool pad, int first_pad, int tid, int nthreads) {
    const int L = ws * ws;
    for (int it = tid; it < L * d; it += nthreads) {
        const int l = it / d, e = it - l * d;
        m[(origin + (size_t)(l / ws) * pitch + (l % ws)) * ld + col0 + e] = (T)0;
    }
    if (pad && ld > first_pad) {
        const int np = ld - first_pad;
        for (int it = tid; it < L * np; it += nthreads) {
            const int l = it / np, e = it - l * np;
            m[(origin + (size_t)(l / ws) * pitch + (l % ws)) * ld + first_pad + e] = (T)0;
        }
    }
}

// LDS element type and strides (in elements) chosen conflict-free for the reads below:
//   row-major [tokens][DP + RPAD]  read 16 B per lane, rows on lanes
//   transposed [DP][L + 4]         read 8 B (bf16) / 16 B (f32) per lane, rows on lanes
template <typename T> struct lds_traits;
template <> struct lds_traits<bf16_t> { static constexpr int RPAD = 8; };
template <> struct lds_traits<float> { static constexpr int RPAD = 4; };

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory
// counter, which would stall on global loads that are deliberately left in flight across it.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0); vmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
}

// ds_read_b64_tr_b16: within each group of 16 lanes, lane 4q+p passes the address of 4 consecutive
// 16-bit elements (columns 4p..4p+3) of row q of a 4 x 16 block; lane i receives column i of the
// block, rows 0..3 in elements 0..3.  Addresses must be 8-byte aligned and EXEC all ones.
typedef short i16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_read_tr16(const void *p) {
    const i16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4 *)p);
    return __builtin_bit_cast(uint2, v);
}

// relative-position table in LDS: [(2w-1)][TS] with TS = 3w, which makes the per-lane lookups of
// one wave-instruction fall on distinct banks (w = 16: two runs of 16 that are 48 apart; w = 8:
// four runs of 8 that are 24 apart).
template <int WS> struct bias_lds {
    static constexpr int TS = 3 * WS;
    static constexpr int ROWS = 2 * WS - 1;
    static constexpr int FLOATS = ROWS * TS;
};


constexpr int BGW_STRIDE = 32, BGW_ROWS = 17;  // per-wave bias-gradient table (w = 16): [qi - kip + 1][dj + 15]

// 8 consecutive bf16 (or DP/2 floats handled by the caller) of a global row, zero past d
__device__ __forceinline__ uint4 load_bf16x8_masked(const bf16_t *row, int e0, int d) {
    uint32_t v[4];
    if ((d & 1) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (e0 + 2 * i < d) ? *reinterpret_cast<const uint32_t *>(row + e0 + 2 * i) : 0u;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t lo = (e0 + 2 * i < d) ? row[e0 + 2 * i] : 0u;
            const uint32_t hi = (e0 + 2 * i + 1 < d) ? row[e0 + 2 * i + 1] : 0u;
            v[i] = lo | (hi << 16);
        }
    }
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// lanes 32-63: rotate the 16-lane rows left by 4 more; lanes 0-31 unchanged
__device__ __forceinline__ float upper_rows_rot4(float x) {
    const int xi = __float_as_int(x);
    return __int_as_float(__builtin_amdgcn_update_dpp(xi, xi, 0x120 + 12, 0xC, 0xF, false));
}
// lanes 0-15 receive the value of lane + 16, lanes 32-47 that of lane + 16 (v_permlane16_swap)
__device__ __forceinline__ float rows_odd_to_even(float x) {
    const unsigned xi = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
    return __uint_as_float(r[1]);
}
// lanes 0-31 receive the value of lane + 32 (v_permlane32_swap); lanes 32-63 are unspecified
__device__ __forceinline__ float upper_half_to_lower(float x) {
    const unsigned xi = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
    return __uint_as_float(r[1]);
}

// ---- pieces shared by the kernels for 16 x 16 windows in bf16 (attn_fwd_w16.hip, attn_bwd_w16.hip) -------------
// XOR pattern of the 16-byte slots of row `row` of a [rows][DP] bf16 image (64- or 128-byte rows).
// ds_read_b128 of one slot of 32 consecutive rows, and ds_read_b64_tr_b16 of 4 consecutive rows x 32 columns,
// then touch every bank exactly once per lane group.
template <int DP> __device__ __forceinline__ int slot_flip(int row) {
    if constexpr (DP == 32) return (row >> 2) & 3;
    else return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
}
// element offset of (row, col) in such an image; col % 8 stays inside the slot
template <int DP> __device__ __forceinline__ int img(int row, int col) {
    return row * DP + (((col >> 3) ^ slot_flip<DP>(row)) << 3) + (col & 7);
}

// A slot = 8 consecutive bf16 (16 bytes) of a head's slice of a token row.  The slice starts on a 4-byte boundary
// only, and its last slot may be partial: `cnt` (0..4) is the number of dwords of the slot that belong to the
// slice; nothing outside them is touched (the neighbouring bytes belong to another head, or lie past the buffer).
struct __attribute__((packed, aligned(4))) u32x4_a4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(4))) u32x3_a4 { uint32_t x, y, z; };
struct __attribute__((packed, aligned(4))) u32x2_a4 { uint32_t x, y; };
__device__ __forceinline__ uint4 load_slot(const bf16_t *p, int cnt) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (cnt >= 4) {
        const u32x4_a4 t = *reinterpret_cast<const u32x4_a4 *>(p);
        v = make_uint4(t.x, t.y, t.z, t.w);
    } else if (cnt == 3) {
        const u32x3_a4 t = *reinterpret_cast<const u32x3_a4 *>(p);
        v = make_uint4(t.x, t.y, t.z, 0u);
    } else if (cnt == 2) {
        const u32x2_a4 t = *reinterpret_cast<const u32x2_a4 *>(p);
        v = make_uint4(t.x, t.y, 0u, 0u);
    } else if (cnt == 1) {
        v.x = *reinterpret_cast<const uint32_t *>(p);
    }
    return v;
}
__device__ __forceinline__ void store_slot(bf16_t *p, uint4 v, int cnt) {
    if (cnt >= 4) {
        u32x4_a4 t;
        t.x = v.x, t.y = v.y, t.z = v.z, t.w = v.w;
        *reinterpret_cast<u32x4_a4 *>(p) = t;
    } else if (cnt == 3) {
        u32x3_a4 t;
        t.x = v.x, t.y = v.y, t.z = v.z;
        *reinterpret_cast<u32x3_a4 *>(p) = t;
    } else if (cnt == 2) {
        u32x2_a4 t;
        t.x = v.x, t.y = v.y;
        *reinterpret_cast<u32x2_a4 *>(p) = t;
    } else if (cnt == 1) {
        *reinterpret_cast<uint32_t *>(p) = v.x;
    }
}
// 8 consecutive bf16 at a 4-byte-aligned address
__device__ __forceinline__ uint4 load16_a4(const bf16_t *p) {
    const u32x4_a4 t = *reinterpret_cast<const u32x4_a4 *>(p);
    return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void store16_a4(bf16_t *p, uint4 v) {
    u32x4_a4 t;
    t.x = v.x, t.y = v.y, t.z = v.z, t.w = v.w;
    *reinterpret_cast<u32x4_a4 *>(p) = t;
}
// dwords of slot c (8 elements each) that lie inside a slice of d elements (d even)
__device__ __forceinline__ int slot_dwords(int d, int c) {
    const int n = (d >> 1) - 4 * c;
    return n < 0 ? 0 : (n > 4 ? 4 : n);
}
// both lane halves receive max(x[l], x[l + 32]), l = lane % 32
__device__ __forceinline__ float max_across_halves(float x) {
    const unsigned xi = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// x as two bf16 (low half: rounded value, high half: rounded remainder) whose sum carries 16 mantissa bits
__device__ __forceinline__ uint32_t bf16_hi_lo(float x) {
    const bf16_t h = f32_to_bf16(x);
    const bf16_t l = f32_to_bf16(x - bf16_to_f32(h));
    return (uint32_t)h | ((uint32_t)l << 16);
}

// A wave's 32 x 32 result (accumulator rows = head-dim columns in registers 4g..4g+3 = columns 8g + 4hh ..+3, lane =
// token 32 wave + l31 of the window) to the head's 30-column slice of the tokens' rows of `dst`, times `mul`, as bf16.
// The tile goes through 2 KB of LDS that only this wave touches (`stage`, a [32][32] image in the slot_flip layout)
// so that one store instruction carries the complete 60-byte pieces of 16 rows, four lanes a row: pieces that arrive
// at a cold cache line in two instructions were seen to cost a read-modify-write of the line each (twice the bytes
// written, and the lines fetched on top).  The last slot leaves as the 16 bytes that end with the slice.
// `dst` is the wave-uniform address of the window's first token's slice; a token's row is rel_row * ld further on.
__device__ __forceinline__ void store_tile_rows_d30(bf16_t *stage, const f32x16 &acc, float mul, bf16_t *dst, int ld, int pitch,
                                                    int wave, int lane) {
    const int l31 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int p = 0; p < 2; ++p) {  // slots 2p + hh: columns 16p + 8hh .. +7 = register groups 2p (hh = 0: first half) ...
        // registers 4g..4g+3 hold columns 8g + 4hh..: a slot of 8 columns is split over the two lane halves, so the
        // halves exchange: afterwards the lower half holds slot 2p whole (g = 2p from both), the upper slot 2p + 1
        const uint32_t a0 = pack_bf16(acc[8 * p] * mul, acc[8 * p + 1] * mul), a1 = pack_bf16(acc[8 * p + 2] * mul, acc[8 * p + 3] * mul);
        const uint32_t b0 = pack_bf16(acc[8 * p + 4] * mul, acc[8 * p + 5] * mul), b1 = pack_bf16(acc[8 * p + 6] * mul, acc[8 * p + 7] * mul);
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        *reinterpret_cast<uint4 *>(stage + img<32>(l31, 8 * (2 * p + hh))) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
    // (a wave's LDS accesses complete in order, and only this wave touches these rows)
    const int c = lane & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (lane >> 2) + 16 * i, tok = 32 * wave + row;
        uint4 v = *reinterpret_cast<const uint4 *>(stage + img<32>(row, 8 * c));
        const uint32_t before = *reinterpret_cast<const uint32_t *>(stage + img<32>(row, 22));  // columns 22, 23
        if (c == 3) v = make_uint4(before, v.x, v.y, v.z);
        const unsigned e = (unsigned)((tok >> 4) * pitch + (tok & 15)) * ld + 8 * c - (c == 3 ? 2 : 0);
        store16_a4(dst + e, v);
    }
}

// forward for 16 x 16 windows in bf16 with an even head dim <= 30, no LePE, no dropout (attn_fwd_w16.hip); returns
// DSW_OK after launching, or a negative value when the shape is not one it takes (the caller then uses the general
// kernel)
int launch_fwd_w16(const AttnGeom &g, const void *qkv, const float *table, float scale, void *out, float *lse, hipStream_t st);

// backward for 16 x 16 windows in bf16 without any exchange of dS between waves (attn_bwd_w16.hip); returns
// DSW_OK after launching, or a negative value when the shape is not one it takes (the caller then uses
// the general kernel)
// (it has no dropout: the caller sends calls with dropout to the general kernel)
int launch_bwd_w16_two_phase(const AttnGeom &g, const void *qkv, const void *out, const void *dout, const float *table,
                             float scale, const float *lse, void *dqkv, float *ws_table, hipStream_t st);

}  // namespace dsw
