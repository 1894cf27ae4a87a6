// Fused window attention, backward: dq, dk, dv, d(bias table) (+ LePE gradients) for one
// (window, head) per workgroup, on the gfx950 matrix cores.
// Reference ops differentiated: DSwinT_arch.py:122-125, :171-172; msft_arch.py:113-136.
//
// Structure.  Each wave owns 32-key tiles of the window and keeps dK^T and dV^T of those keys
// in accumulators for the whole sweep, with its K and V fragments in registers.  The workgroup
// walks the window's 32-query blocks; a block's Q and dO rows are staged in LDS (row-major, and
// transposed for the bf16 operand reads).  With the KEY on the MFMA lane, S = Q K^T and
// dP = dO V^T come out as (query = row = register, key = column = lane), which is already the
// B operand of dV^T += dO^T P and dK^T += Q^T dS (they sum over the accumulator's row index).
// Only dS crosses LDS, once per block, for dQ = dS K (16x16 output tiles, one per wave, summed
// over all keys - no cross-wave reduction).  P is recomputed from the forward's log-sum-exp;
// delta = rowsum(dO * O) comes from the saved output.  The bias-table gradient is accumulated
// with LDS float adds into one private table per wave (program order inside a wave => bitwise
// reproducible), summed in wave order and written as a per-(window, head) partial that a second
// kernel reduces over windows in a fixed order.
#include <utility>

#include "attn_common.hpp"

#ifndef DSW_ABL
#define DSW_ABL 0
#endif
#pragma clang fp contract(fast)

namespace dsw {

template <typename T, int WS, int DP> struct BwdCfg {
    static constexpr bool BF = sizeof(T) == 2;
    static constexpr int L = WS * WS;
    static constexpr int NT = L / 32;                 // key tiles = query blocks
    static constexpr int NW = NT < 4 ? NT : 4;        // waves
    static constexpr int KT = NT / NW;                // key tiles per wave
    static constexpr int THREADS = NW * 64;
    static constexpr int QS = DP + lds_traits<T>::RPAD;  // row-major stride of Qb / dOb / Ks
    static constexpr int TQS = 32 + 4;                    // stride of the transposed block tiles (bf16)
    static constexpr int KTS = L + 8;                     // stride of Kt (bf16) [DP][L]
    static constexpr int SS = BF ? L + 8 : L + 4;         // stride of the dS block [32][L]
    static constexpr int QTILES = (32 / 16) * (DP / 16);  // 16x16 dQ tiles per block
    static constexpr int QT_PER_WAVE = QTILES / NW;
    // element counts (in T) of the LDS regions
    static constexpr int N_K = BF ? DP * KTS : L * QS;
    static constexpr int N_DS = 32 * SS;
    static constexpr int N_QB = 32 * QS;
    static constexpr int N_QBT = BF ? DP * TQS : 0;
    static constexpr size_t lds_bytes(bool bias) {
        return (size_t)(N_K + N_DS + 2 * N_QB + 2 * N_QBT) * sizeof(T) + (size_t)2 * L * sizeof(float) +
               (bias ? (size_t)(bias_lds<WS>::FLOATS + (WS == 16 ? NW * 31 * 32 : 0)) * sizeof(float) : 0);
    }
    static constexpr int MIN_WAVES = (BF && !(WS == 16 && DP == 64)) ? 2 : 1;
};

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

// ---- d(bias table) for 16 x 16 windows: diagonal sums in registers ---------------------------------
// In a dS tile (query = row = register, key = column = lane) the table entry of an element is
// (di, dj) = (qi - ki, qj - kj).  Rotating each register's 16-lane rows by its qj puts every
// element with the same dj in the same lane, so the 8 registers of one query row-group add up
// element-wise; lanes whose rotation wrapped belong to dj + 16 and are kept apart by a 0/1 mask.
template <int QJ_LO>  // rotate lane rows left by qj: lanes 0-31 hold qj = QJ_LO, lanes 32-63 qj = QJ_LO + 4
__device__ __forceinline__ float rot_rows_by_qj(float x) {
    const int xi = __float_as_int(x);
    int y = xi;
    if constexpr (((16 - QJ_LO) & 15) != 0) y = __builtin_amdgcn_update_dpp(y, xi, 0x120 + ((16 - QJ_LO) & 15), 0x3, 0xF, false);
    if constexpr (((12 - QJ_LO) & 15) != 0) y = __builtin_amdgcn_update_dpp(y, xi, 0x120 + ((12 - QJ_LO) & 15), 0xC, 0xF, false);
    return __int_as_float(y);
}
// lanes 0-31 receive the value of lane + 32 (v_permlane32_swap); lanes 32-63 are unspecified
__device__ __forceinline__ float upper_half_to_lower(float x) {
    const unsigned xi = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
    return __uint_as_float(r[1]);
}
template <int R>
__device__ __forceinline__ void diag_step(const f32x16 &ds, const float (&wrapf)[8], float (&all)[2], float (&wrapped)[2]) {
    constexpr int QJ_LO = (R & 3) + 8 * ((R >> 2) & 1);
    const float y = rot_rows_by_qj<QJ_LO>(ds[R]);
    all[R >> 3] += y;
    wrapped[R >> 3] = fmaf(y, wrapf[R & 7], wrapped[R >> 3]);
}
template <int... R>
__device__ __forceinline__ void diag_sums(std::integer_sequence<int, R...>, const f32x16 &ds, const float (&wrapf)[8],
                                          float (&all)[2], float (&wrapped)[2]) {
    (diag_step<R>(ds, wrapf, all, wrapped), ...);
}
constexpr int BGW_STRIDE = 32, BGW_FLOATS = 31 * 32;  // per-wave gradient table [di+15][dj+15], w = 16

template <typename T, int WS, int DP, bool BIAS, bool LEPE>
__global__ __launch_bounds__((BwdCfg<T, WS, DP>::THREADS), (BwdCfg<T, WS, DP>::MIN_WAVES)) void win_attn_bwd_kernel(
    AttnGeom g, const T *__restrict__ qkv, const T *__restrict__ out, const T *__restrict__ dout,
    const float *__restrict__ table, float scale, const float *__restrict__ lepe_w,
    const float *__restrict__ lepe_b, const float *__restrict__ lse, T *__restrict__ dqkv,
    float *__restrict__ ws_table, const float *__restrict__ delta_in) {
    using Cfg = BwdCfg<T, WS, DP>;
    constexpr bool BF = Cfg::BF;
    constexpr int L = Cfg::L, NT = Cfg::NT, NW = Cfg::NW, KT = Cfg::KT, THREADS = Cfg::THREADS;
    constexpr int QS = Cfg::QS, TQS = Cfg::TQS, KTS = Cfg::KTS, SS = Cfg::SS;
    constexpr int TS = bias_lds<WS>::TS, TBF = bias_lds<WS>::FLOATS, TR = bias_lds<WS>::ROWS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *Kl = reinterpret_cast<T *>(smem);   // bf16: K^T [DP][KTS];  f32: K [L][QS]
    T *dSs = Kl + Cfg::N_K;                // [32][SS]
    T *Qb = dSs + Cfg::N_DS;               // [32][QS]
    T *dOb = Qb + Cfg::N_QB;               // [32][QS]
    T *Qbt = dOb + Cfg::N_QB;              // bf16 only: [DP][TQS]
    T *dObt = Qbt + Cfg::N_QBT;            // bf16 only: [DP][TQS]
    float *delta_s = reinterpret_cast<float *>(dObt + Cfg::N_QBT);  // [L]
    float *lse_s = delta_s + L;                                     // [L], already times log2(e)
    float *tb = lse_s + L;                                          // BIAS: [TBF]
    float *bgw = tb + TBF;                                          // BIAS, w = 16: [NW][BGW_FLOATS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int id = xcd_remap(blockIdx.x, g.n_win * g.heads);
    const int win = id / g.heads, head = id - win * g.heads;
    const int d = g.d, C = g.C, C3 = 3 * g.C;
    int qoff, koff, voff;
    qkv_offsets(g, head, qoff, koff, voff);
    size_t origin;
    int pitch;
    if (g.H == 0) {
        origin = (size_t)win * L;
        pitch = WS;
    } else {
        const int per = g.nWh * g.nWw;
        const int b = win / per, r = win - b * per;
        const int wr = r / g.nWw, wc = r - wr * g.nWw;
        origin = ((size_t)b * g.H + wr * WS) * g.W + wc * WS;
        pitch = g.W;
    }
    auto trow = [&](int l) -> size_t { return origin + (size_t)(l / WS) * pitch + (l % WS); };

    // Loads are issued in batches before any of them is consumed (a load-use-per-trip loop pays a
    // full memory round trip per trip); the "pair" path moves two bf16 per lane and needs an even
    // head dim, the "element" path is the general one (f32, odd head dims).
    constexpr int DW = DP / 2;
    const bool pair_path = BF && ((d & 1) == 0);  // uniform
    const float sc2 = scale * LOG2E;
    auto tof = [](T v) -> float {
        if constexpr (BF) return bf16_to_f32(v);
        else return v;
    };
    auto bits = [](T v) -> uint32_t {
        if constexpr (BF) return (uint32_t)v;
        else return __float_as_uint(v);
    };

    // ---- prologue: K into LDS, delta = rowsum(dO * O), lse, bias tables -------------------------
    if (pair_path) {
        if constexpr (BF) {
            constexpr int ITER = L * DW / THREADS, BATCH = ITER < 8 ? ITER : 8;
            for (int i0 = 0; i0 < ITER; i0 += BATCH) {
                uint32_t kv[BATCH], ov[BATCH], gv[BATCH];
#pragma unroll
                for (int i = 0; i < BATCH; ++i) {
                    const int it = tid + (i0 + i) * THREADS;
                    const int row = it / DW, ec = min((it - row * DW) * 2, d - 2);
                    const size_t tr = trow(row);
                    kv[i] = *reinterpret_cast<const uint32_t *>(qkv + tr * C3 + koff + ec);
                    ov[i] = *reinterpret_cast<const uint32_t *>(out + tr * C + head * d + ec);
                    gv[i] = *reinterpret_cast<const uint32_t *>(dout + tr * C + head * d + ec);
                }
#pragma unroll
                for (int i = 0; i < BATCH; ++i) {
                    const int it = tid + (i0 + i) * THREADS;
                    const int row = it / DW, e = (it - row * DW) * 2;
                    const bool ok = e < d;
                    const uint32_t k2 = ok ? kv[i] : 0u;
                    Kl[e * KTS + row] = (T)(k2 & 0xffffu);
                    Kl[(e + 1) * KTS + row] = (T)(k2 >> 16);
                    float part = ok ? __uint_as_float(ov[i] << 16) * __uint_as_float(gv[i] << 16) +
                                          __uint_as_float(ov[i] & 0xffff0000u) * __uint_as_float(gv[i] & 0xffff0000u)
                                    : 0.f;
#pragma unroll
                    for (int o = 1; o < DW; o <<= 1) part += __shfl_xor(part, o, 64);
                    if (e == 0) delta_s[row] = part;
                }
            }
        }
    } else {
        constexpr int ITER = L * DP / THREADS, BATCH = 8;
        for (int i0 = 0; i0 < ITER; i0 += BATCH) {
            T kv[BATCH], ov[BATCH], gv[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int it = tid + (i0 + i) * THREADS;
                const int row = it / DP, ec = min(it - row * DP, d - 1);
                const size_t tr = trow(row);
                kv[i] = qkv[tr * C3 + koff + ec];
                ov[i] = out[tr * C + head * d + ec];
                gv[i] = dout[tr * C + head * d + ec];
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int it = tid + (i0 + i) * THREADS;
                const int row = it / DP, e = it - row * DP;
                const bool ok = e < d;
                if constexpr (BF) Kl[e * KTS + row] = ok ? kv[i] : (T)0;
                else Kl[row * QS + e] = ok ? kv[i] : (T)0;
                float part = ok ? tof(ov[i]) * tof(gv[i]) : 0.f;
#pragma unroll
                for (int o = 1; o < DP; o <<= 1) part += __shfl_xor(part, o, 64);
                if (e == 0) delta_s[row] = part;
            }
        }
    }
    for (int row = tid; row < L; row += THREADS) lse_s[row] = lse[((size_t)win * g.heads + head) * L + row] * LOG2E;
    if (delta_in != nullptr) {
        // LePE: `out` holds SDPA + LePE, so rowsum(dO * O_sdpa) was computed by the LePE pre-pass
        __syncthreads();
        for (int row = tid; row < L; row += THREADS) delta_s[row] = delta_in[((size_t)win * g.heads + head) * L + row];
    }
    if constexpr (BIAS) {
        for (int it = tid; it < TR * TR; it += THREADS) {
            const int di = it / TR, dj = it - di * TR;
            tb[di * TS + dj] = table[(size_t)it * g.heads + head] * LOG2E;
        }
        if constexpr (WS == 16) {
            for (int it = tid; it < NW * BGW_FLOATS; it += THREADS) bgw[it] = 0.f;
        }
    }
    // 1.0 where rotating a register's lane row by its qj wrapped around (those elements belong
    // to dj + 16); index = register & 7, whose qj is (j & 3) + 8 (j >> 2) + 4 hh
    float wrapf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wrapf[j] = ((lane & 15) + (j & 3) + 8 * (j >> 2) + 4 * hh >= 16) ? 1.f : 0.f;

    // this wave's K and V fragments (B operands: k-dim = head dim, column = key on the lane)
    uint4 kfb[BF ? KT : 1][DP / 16], vfb[BF ? KT : 1][DP / 16];
    float kff[BF ? 1 : KT][BF ? 1 : DP / 2], vff[BF ? 1 : KT][BF ? 1 : DP / 2];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int key = 32 * (wave * KT + kt) + l31;
        const T *krow = qkv + trow(key) * C3 + koff;
        const T *vrow = qkv + trow(key) * C3 + voff;
        if constexpr (BF) {
#pragma unroll
            for (int s = 0; s < DP / 16; ++s) {
                kfb[kt][s] = load_bf16x8_masked(krow, 16 * s + 8 * hh, d);
                vfb[kt][s] = load_bf16x8_masked(vrow, 16 * s + 8 * hh, d);
            }
        } else {
#pragma unroll
            for (int s = 0; s < DP / 2; ++s) {
                const int e = hh * (DP / 2) + s;
                kff[kt][s] = (e < d) ? krow[e] : 0.f;
                vff[kt][s] = (e < d) ? vrow[e] : 0.f;
            }
        }
    }

    f32x16 dKacc[KT][DP / 32], dVacc[KT][DP / 32];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int u = 0; u < DP / 32; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dKacc[kt][u][r] = 0.f;
                dVacc[kt][u][r] = 0.f;
            }

    // ---- the next block's Q and dO rows travel in registers while the current block computes ----
    constexpr int NI_P = 32 * DW / THREADS, NI_E = 32 * DP / THREADS;
    uint32_t nq[NI_E], ng[NI_E];
    auto load_block = [&](int qb) {
        if (pair_path) {
#pragma unroll
            for (int i = 0; i < NI_P; ++i) {
                const int it = tid + i * THREADS;
                const int row = it / DW, ec = min((it - row * DW) * 2, d - 2);
                const size_t tr = trow(qb * 32 + row);
#if DSW_ABL == 15
                nq[i] = (uint32_t)tr * 77u + ec; ng[i] = (uint32_t)tr * 13u + ec;
#else
                nq[i] = *reinterpret_cast<const uint32_t *>(qkv + tr * C3 + qoff + ec);
                ng[i] = *reinterpret_cast<const uint32_t *>(dout + tr * C + head * d + ec);
#endif
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI_E; ++i) {
                const int it = tid + i * THREADS;
                const int row = it / DP, ec = min(it - row * DP, d - 1);
                const size_t tr = trow(qb * 32 + row);
                nq[i] = bits(qkv[tr * C3 + qoff + ec]);
                ng[i] = bits(dout[tr * C + head * d + ec]);
            }
        }
    };
    // Q is stored times scale*log2(e): S comes out of the MFMA in the base-2 domain, and dK picks
    // up the inverse factor once, in the epilogue.
    auto store_block = [&]() {
        if (pair_path) {
            if constexpr (BF) {
#pragma unroll
                for (int i = 0; i < NI_P; ++i) {
                    const int it = tid + i * THREADS;
                    const int row = it / DW, e = (it - row * DW) * 2;
                    const bool ok = e < d;
                    const uint32_t q2 = ok ? pack_bf16(__uint_as_float(nq[i] << 16) * sc2, __uint_as_float(nq[i] & 0xffff0000u) * sc2) : 0u;
                    const uint32_t g2 = ok ? ng[i] : 0u;
                    *reinterpret_cast<uint32_t *>(Qb + row * QS + e) = q2;
                    *reinterpret_cast<uint32_t *>(dOb + row * QS + e) = g2;
                    Qbt[e * TQS + row] = (T)(q2 & 0xffffu);
                    Qbt[(e + 1) * TQS + row] = (T)(q2 >> 16);
                    dObt[e * TQS + row] = (T)(g2 & 0xffffu);
                    dObt[(e + 1) * TQS + row] = (T)(g2 >> 16);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI_E; ++i) {
                const int it = tid + i * THREADS;
                const int row = it / DP, e = it - row * DP;
                const bool ok = e < d;
                if constexpr (BF) {
                    const T qv = ok ? f32_to_bf16(bf16_to_f32((T)nq[i]) * sc2) : (T)0;
                    const T gv = ok ? (T)ng[i] : (T)0;
                    Qb[row * QS + e] = qv;
                    dOb[row * QS + e] = gv;
                    Qbt[e * TQS + row] = qv;
                    dObt[e * TQS + row] = gv;
                } else {
                    Qb[row * QS + e] = ok ? __uint_as_float(nq[i]) * sc2 : 0.f;
                    dOb[row * QS + e] = ok ? __uint_as_float(ng[i]) : 0.f;
                }
            }
        }
    };

    constexpr int BG_N = (2 * WS - 1 + 7) / 8;  // bias-gradient entries (di values) per thread
    float bgacc[BG_N];
#pragma unroll
    for (int n = 0; n < BG_N; ++n) bgacc[n] = 0.f;

    load_block(0);
    for (int qb = 0; qb < NT; ++qb) {
        lds_barrier();  // previous block's LDS reads are done (first pass: prologue writes land)
        store_block();
        lds_barrier();
        if (qb + 1 < NT) load_block(qb + 1);

#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int tile = wave * KT + kt;
            const int key = 32 * tile + l31;
            // table entry of (q, key): (qi-ki+w-1)*TS + (qj-kj+w-1); q = qb*32 + rho(r) + 4hh
            const int ki = key / WS, kj = key - ki * WS;
            const int lane_base = (WS - 1) * (TS + 1) - ki * TS - kj + 4 * hh + (qb * 32 / WS) * TS;
            // accumulators start from the row constants: S' = S*log2e + bias - lse, dP' = dP - delta
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q_c = rho(r);
                // per-row constants of accumulator row r: query qb*32 + rho(r) + 4hh
                float v0 = -lse_s[qb * 32 + q_c + 4 * hh];
                if constexpr (BIAS) v0 += tb[lane_base + (q_c / WS) * TS + (q_c % WS)];
                sacc[r] = v0;
                pacc[r] = -delta_s[qb * 32 + q_c + 4 * hh];
            }
            // ---- S = Q K^T and dP = dO V^T (rows = queries, lane = key) ---------------------------
            if constexpr (BF) {
#pragma unroll
                for (int s = 0; s < DP / 16; ++s) {
                    const uint4 qa = *reinterpret_cast<const uint4 *>(Qb + l31 * QS + 16 * s + 8 * hh);
                    const uint4 ga = *reinterpret_cast<const uint4 *>(dOb + l31 * QS + 16 * s + 8 * hh);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(qa), as_bf16x8(kfb[kt][s]), sacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ga), as_bf16x8(vfb[kt][s]), pacc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int s4 = 0; s4 < DP / 8; ++s4) {
                    const float4 qa = *reinterpret_cast<const float4 *>(Qb + l31 * QS + hh * (DP / 2) + 4 * s4);
                    const float4 ga = *reinterpret_cast<const float4 *>(dOb + l31 * QS + hh * (DP / 2) + 4 * s4);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, kff[kt][4 * s4 + 0], sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, kff[kt][4 * s4 + 1], sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, kff[kt][4 * s4 + 2], sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, kff[kt][4 * s4 + 3], sacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.x, vff[kt][4 * s4 + 0], pacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.y, vff[kt][4 * s4 + 1], pacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.z, vff[kt][4 * s4 + 2], pacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.w, vff[kt][4 * s4 + 3], pacc, 0, 0, 0);
                }
            }
            // ---- P = exp2(S'), dS = P * dP' ------------------------------------------------------------
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#if DSW_ABL == 17
                const float p = sacc[r] * 0.001f;
#else
                const float p = __builtin_amdgcn_exp2f(sacc[r]);
#endif
                const float ds = p * pacc[r];
                sacc[r] = p;
                pacc[r] = ds;
            }
            if constexpr (BIAS && WS == 16 && DSW_ABL != 13) {
                float all[2] = {0.f, 0.f}, wrapped[2] = {0.f, 0.f};
                diag_sums(std::make_integer_sequence<int, 16>{}, pacc, wrapf, all, wrapped);
#pragma unroll
                for (int G = 0; G < 2; ++G) {  // lanes 0-31 += lanes 32-63 (one VALU swap, no LDS round trip)
                    all[G] += upper_half_to_lower(all[G]);
                    wrapped[G] += upper_half_to_lower(wrapped[G]);
                }
                if (hh == 0) {
                    // lane (kip, m): entries (di, -m) and, for wrapped elements, (di, 16 - m).  Plain
                    // read-add-write on the wave's own table (no two lanes of one instruction share an
                    // entry; LDS float atomics cost hundreds of cycles each on this chip)
                    const int m = lane & 15, kip = (lane >> 4) & 1;
                    float *tw = bgw + wave * BGW_FLOATS;
#pragma unroll
                    for (int G = 0; G < 2; ++G) {
                        const int row = (2 * (qb - tile) + G - kip + 15) * BGW_STRIDE;
                        tw[row + 15 - m] += all[G] - wrapped[G];
                        if (m > 0) tw[row + 31 - m] += wrapped[G];
                    }
                }
            }
            // ---- dS block to LDS for dQ ---------------------------------------------------------------
#pragma unroll
            for (int r = 0; r < (DSW_ABL == 14 ? 0 : 16); ++r) {
                if constexpr (BF) dSs[(rho(r) + 4 * hh) * SS + key] = f32_to_bf16(pacc[r]);
                else dSs[(rho(r) + 4 * hh) * SS + key] = pacc[r];
            }
            // ---- dV^T += dO^T P,  dK^T += Q^T dS  (sum over the accumulator rows = queries) -----------
            if constexpr (BF) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const uint4 pb = make_uint4(pack_bf16(sacc[8 * s + 0], sacc[8 * s + 1]), pack_bf16(sacc[8 * s + 2], sacc[8 * s + 3]),
                                                pack_bf16(sacc[8 * s + 4], sacc[8 * s + 5]), pack_bf16(sacc[8 * s + 6], sacc[8 * s + 7]));
                    const uint4 db = make_uint4(pack_bf16(pacc[8 * s + 0], pacc[8 * s + 1]), pack_bf16(pacc[8 * s + 2], pacc[8 * s + 3]),
                                                pack_bf16(pacc[8 * s + 4], pacc[8 * s + 5]), pack_bf16(pacc[8 * s + 6], pacc[8 * s + 7]));
#pragma unroll
                    for (int u = 0; u < DP / 32; ++u) {
                        const T *gr = dObt + (32 * u + l31) * TQS + 16 * s + 4 * hh;
                        const T *qr = Qbt + (32 * u + l31) * TQS + 16 * s + 4 * hh;
                        const uint2 g0 = *reinterpret_cast<const uint2 *>(gr), g1 = *reinterpret_cast<const uint2 *>(gr + 8);
                        const uint2 q0 = *reinterpret_cast<const uint2 *>(qr), q1 = *reinterpret_cast<const uint2 *>(qr + 8);
                        dVacc[kt][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(g0.x, g0.y, g1.x, g1.y)), as_bf16x8(pb), dVacc[kt][u], 0, 0, 0);
                        dKacc[kt][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(q0.x, q0.y, q1.x, q1.y)), as_bf16x8(db), dKacc[kt][u], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#pragma unroll
                    for (int u = 0; u < DP / 32; ++u) {
                        const float ga = dOb[(rho(r) + 4 * hh) * QS + 32 * u + l31];
                        const float qa = Qb[(rho(r) + 4 * hh) * QS + 32 * u + l31];
                        dVacc[kt][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, sacc[r], dVacc[kt][u], 0, 0, 0);
                        dKacc[kt][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa, pacc[r], dKacc[kt][u], 0, 0, 0);
                    }
                }
            }
        }
        lds_barrier();  // the whole dS block is in LDS

        // ---- d(bias table): every thread owns a few table entries (di, dj) and sums the dS elements
        // of this block that map to them, straight from the dS tile - a gather in a fixed order,
        // so no atomics and bitwise reproducible.  Lanes that differ in dj read neighbouring keys.
        if constexpr (BIAS && WS != 16) {
            const int dj = (tid % (2 * WS)) - (WS - 1);   // lane WS*2-1 of each group is idle
            const int dig = tid / (2 * WS);                // 8 groups of di values
            if (dj < WS) {
#pragma unroll
                for (int n = 0; n < BG_N; ++n) {
                    const int di = dig + 8 * n - (WS - 1);
                    if (di < WS) {
                        float acc = 0.f;
#pragma unroll
                        for (int qr = 0; qr < 32 / WS; ++qr) {
                            const int ki = qb * (32 / WS) + qr - di;
                            if (ki >= 0 && ki < WS) {
#pragma unroll
                                for (int qj = 0; qj < WS; ++qj) {
                                    const int kj = qj - dj;
                                    if (kj >= 0 && kj < WS) acc += tof(dSs[(qr * WS + qj) * SS + ki * WS + kj]);
                                }
                            }
                        }
                        bgacc[n] += acc;
                    }
                }
            }
        }

        // ---- dQ block = dS K * scale: 16x16 output tiles, one (or more) per wave, K-dim = all keys ---
#if DSW_ABL == 11
        if (scale != 12345.f) continue;
#endif
#pragma unroll
        for (int tq = 0; tq < Cfg::QT_PER_WAVE; ++tq) {
            const int tile = wave * Cfg::QT_PER_WAVE + tq;
            const int mq = tile / (DP / 16), nd = tile - mq * (DP / 16);
            const int l15 = lane & 15, l4 = lane >> 4;
            f32x4 qacc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) {
#pragma unroll 4
                for (int s = 0; s < L / 32; ++s) {
                    const uint4 a = *reinterpret_cast<const uint4 *>(dSs + (16 * mq + l15) * SS + 32 * s + 8 * l4);
                    const uint4 b = *reinterpret_cast<const uint4 *>(Kl + (16 * nd + l15) * KTS + 32 * s + 8 * l4);
                    qacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a), as_bf16x8(b), qacc, 0, 0, 0);
                }
            } else {
#pragma unroll 4
                for (int s = 0; s < L / 4; ++s) {
                    const float a = dSs[(16 * mq + l15) * SS + 4 * s + l4];
                    const float b = Kl[(4 * s + l4) * QS + 16 * nd + l15];
                    qacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, qacc, 0, 0, 0);
                }
            }
            // C layout: column (head dim) = lane & 15, row (query) = 4*(lane>>4) + reg
            const int e = 16 * nd + l15;
#if DSW_ABL == 12
            if (qacc[0] == 12345.f)
#endif
            if (e < d) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qb * 32 + 16 * mq + 4 * l4 + r;
                    io<T>::st(dqkv + trow(q) * C3 + qoff + e, qacc[r] * scale);
                }
            }
        }
    }

    // ---- epilogue: dK^T, dV^T (rows = head dim in registers, lane = key) ------------------------
    if (pair_path) {
        if constexpr (BF) {
            // each wave turns its tiles into [key][head dim] rows in a private slice of the (now free)
            // dS region, so that the global stores are runs of whole rows instead of one dword in each
            // of 32 rows; a wave's LDS accesses complete in order and nobody else touches its slice
            __syncthreads();
#if DSW_ABL == 16
            if (scale != 12345.f) return;
#endif
            T *slice = dSs + wave * (32 * QS);
            constexpr int DWR = DP / 2, RPI = 64 / DWR;
            const int dw = lane % DWR, rsub = lane / DWR;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int which = 0; which < 2; ++which) {  // 0: dK, 1: dV
#pragma unroll
                    for (int u = 0; u < DP / 32; ++u) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            float v[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = which == 0 ? dKacc[kt][u][4 * g4 + j] * LN2 : dVacc[kt][u][4 * g4 + j];
                            *reinterpret_cast<uint2 *>(slice + l31 * QS + 32 * u + 8 * g4 + 4 * hh) =
                                make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 32 / RPI; ++i) {
                        const int row = i * RPI + rsub;
                        uint32_t v = *reinterpret_cast<const uint32_t *>(slice + row * QS + 2 * dw);
                        if (2 * dw < d) {
                            T *dst = dqkv + trow(32 * (wave * KT + kt) + row) * C3 + (which == 0 ? koff : voff) + 2 * dw;
                            if (which == 1 && delta_in != nullptr) {  // add the LePE share of dV left there by the pre-pass
                                const uint32_t old = *reinterpret_cast<const uint32_t *>(dst);
                                v = pack_bf16(__uint_as_float(v << 16) + __uint_as_float(old << 16),
                                              __uint_as_float(v & 0xffff0000u) + __uint_as_float(old & 0xffff0000u));
                            }
                            *reinterpret_cast<uint32_t *>(dst) = v;
                        }
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int key = 32 * (wave * KT + kt) + l31;
            T *dkrow = dqkv + trow(key) * C3 + koff;
            T *dvrow = dqkv + trow(key) * C3 + voff;
#pragma unroll
            for (int u = 0; u < DP / 32; ++u) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int e = 32 * u + rho(r) + 4 * hh;
                    if (e < d) {
                        float v0 = dVacc[kt][u][r];
                        if (delta_in != nullptr) v0 += io<T>::ld(dvrow + e);
                        io<T>::st(dkrow + e, dKacc[kt][u][r] * LN2);
                        io<T>::st(dvrow + e, v0);
                    }
                }
            }
        }
    }

    if constexpr (BIAS) {
        // the table entered the logits times log2(e) inside exp2, so d/dtable is dS itself
        float *dst = ws_table + ((size_t)win * g.heads + head) * (TR * TR);
        if constexpr (WS == 16) {
            __syncthreads();
            for (int it = tid; it < TR * TR; it += THREADS) {
                const int di = it / TR, dj = it - di * TR;
                float acc = 0.f;
#pragma unroll
                for (int wv = 0; wv < NW; ++wv) acc += bgw[wv * BGW_FLOATS + di * BGW_STRIDE + dj];
                dst[it] = acc;
            }
        } else {
            const int dj = (tid % (2 * WS)) - (WS - 1);
            const int dig = tid / (2 * WS);
            if (dj < WS) {
#pragma unroll
                for (int n = 0; n < BG_N; ++n) {
                    const int di = dig + 8 * n - (WS - 1);
                    if (di < WS) dst[(di + WS - 1) * TR + dj + WS - 1] = bgacc[n];
                }
            }
        }
    }
}

// d_table[t][h] = sum over windows of ws[(win*heads + h)*T + t], in a fixed order: 16 window
// groups (window index mod 16) summed by 16 threads each, then the 16 partials in group order.
__global__ __launch_bounds__(256) void bias_grad_reduce_kernel(const float *__restrict__ ws, int n_win, int heads, int T,
                                                               float *__restrict__ d_table) {
    __shared__ float part[16][17];
    const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + e;  // entry index t*heads + h
    float acc = 0.f;
    if (i < T * heads) {
        const int t = i / heads, h = i - t * heads;
        for (int wdx = grp; wdx < n_win; wdx += 16) acc += ws[((size_t)wdx * heads + h) * T + t];
    }
    part[grp][e] = acc;
    __syncthreads();
    if (grp == 0 && i < T * heads) {
        float tot = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) tot += part[k2][e];
        d_table[i] = tot;
    }
}

// ---- LePE backward pre-pass (msft_arch.py:113-125, :136): one block per (window, head) -------------
// With out = SDPA(q,k,v) + LePE(v):  delta = rowsum(dO * (out - LePE(v)))   (for the main kernel),
// dV_lepe = conv^T(dO) written into the v slots of dqkv (the main kernel adds its dV on top),
// and per-window partial sums of the depthwise weight / bias gradients.
template <typename T, int WS>
__global__ __launch_bounds__(256) void lepe_bwd_pre_kernel(AttnGeom g, const T *__restrict__ qkv, const T *__restrict__ out,
                                                           const T *__restrict__ dout, const float *__restrict__ lepe_w,
                                                           const float *__restrict__ lepe_b, T *__restrict__ dqkv,
                                                           float *__restrict__ delta_out, float *__restrict__ ws_lepe) {
    constexpr int L = WS * WS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int d = g.d, C = g.C, C3 = 3 * g.C;
    T *Vs = reinterpret_cast<T *>(smem);  // [L][d]
    T *Gs = Vs + L * d;                   // dO
    T *Os = Gs + L * d;                   // out
    float *wl = reinterpret_cast<float *>(Os + L * d + ((L * d) & 1));  // [d][10]: 9 taps + bias
    const int tid = threadIdx.x;
    const int win = blockIdx.x / g.heads, head = blockIdx.x - win * g.heads;
    int qoff, koff, voff;
    qkv_offsets(g, head, qoff, koff, voff);
    size_t origin;
    int pitch;
    if (g.H == 0) {
        origin = (size_t)win * L;
        pitch = WS;
    } else {
        const int per = g.nWh * g.nWw;
        const int b = win / per, r = win - b * per;
        const int wr = r / g.nWw, wc = r - wr * g.nWw;
        origin = ((size_t)b * g.H + wr * WS) * g.W + wc * WS;
        pitch = g.W;
    }
    auto trow = [&](int l) -> size_t { return origin + (size_t)(l / WS) * pitch + (l % WS); };
    for (int it = tid; it < L * d; it += 256) {
        const int row = it / d, e = it - row * d;
        const size_t tr = trow(row);
        Vs[it] = qkv[tr * C3 + voff + e];
        Gs[it] = dout[tr * C + head * d + e];
        Os[it] = out[tr * C + head * d + e];
    }
    for (int it = tid; it < d * 10; it += 256) {
        const int e = it / 10, k = it - e * 10;
        wl[it] = k < 9 ? lepe_w[(size_t)(head * d + e) * 9 + k] : lepe_b[head * d + e];
    }
    __syncthreads();
    // delta
    for (int q = tid; q < L; q += 256) {
        const int qi = q / WS, qj = q - qi * WS;
        float acc = 0.f;
        for (int e = 0; e < d; ++e) {
            float lp = wl[e * 10 + 9];
#pragma unroll
            for (int a = -1; a <= 1; ++a)
#pragma unroll
                for (int bb = -1; bb <= 1; ++bb) {
                    const int ni = qi + a, nj = qj + bb;
                    if (ni >= 0 && ni < WS && nj >= 0 && nj < WS) lp += wl[e * 10 + (a + 1) * 3 + bb + 1] * io<T>::ld(Vs + (ni * WS + nj) * d + e);
                }
            acc += io<T>::ld(Gs + q * d + e) * (io<T>::ld(Os + q * d + e) - lp);
        }
        delta_out[((size_t)win * g.heads + head) * L + q] = acc;
    }
    // dV_lepe[k][e] = sum_taps W[e][tap] * dO[k - tap][e]
    for (int it = tid; it < L * d; it += 256) {
        const int k = it / d, e = it - k * d;
        const int ki = k / WS, kj = k - ki * WS;
        float acc = 0.f;
#pragma unroll
        for (int a = -1; a <= 1; ++a)
#pragma unroll
            for (int bb = -1; bb <= 1; ++bb) {
                const int ni = ki - a, nj = kj - bb;
                if (ni >= 0 && ni < WS && nj >= 0 && nj < WS) acc += wl[e * 10 + (a + 1) * 3 + bb + 1] * io<T>::ld(Gs + (ni * WS + nj) * d + e);
            }
        io<T>::st(dqkv + trow(k) * C3 + voff + e, acc);
    }
    // weight / bias gradient partials of this window
    for (int it = tid; it < d * 10; it += 256) {
        const int e = it / 10, k = it - e * 10;
        float acc = 0.f;
        if (k == 9) {
            for (int q = 0; q < L; ++q) acc += io<T>::ld(Gs + q * d + e);
        } else {
            const int a = k / 3 - 1, bb = k - (k / 3) * 3 - 1;
            for (int q = 0; q < L; ++q) {
                const int ni = q / WS + a, nj = q % WS + bb;
                if (ni >= 0 && ni < WS && nj >= 0 && nj < WS) acc += io<T>::ld(Gs + q * d + e) * io<T>::ld(Vs + (ni * WS + nj) * d + e);
            }
        }
        ws_lepe[((size_t)win * C + head * d + e) * 10 + k] = acc;
    }
}

// d_lepe_w[c][tap], d_lepe_b[c] = sum over windows (ascending) of the partials
__global__ void lepe_grad_reduce_kernel(const float *__restrict__ ws, int n_win, int C, float *__restrict__ d_w,
                                        float *__restrict__ d_b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * 10) return;
    float acc = 0.f;
    for (int wdx = 0; wdx < n_win; ++wdx) acc += ws[(size_t)wdx * C * 10 + i];
    const int c = i / 10, k = i - c * 10;
    if (k < 9) d_w[c * 9 + k] = acc;
    else d_b[c] = acc;
}

template <typename T, int WS, int DP>
static int launch_bwd(const AttnGeom &g, const void *qkv, const void *out, const void *dout, const float *table,
                      float scale, const float *lepe_w, const float *lepe_b, const float *lse, void *dqkv,
                      float *d_table, float *d_lepe_w, float *d_lepe_b, float *ws_table, float *ws_lepe,
                      float *ws_delta, hipStream_t st) {
    using Cfg = BwdCfg<T, WS, DP>;
    const bool bias = table != nullptr, lepe = lepe_w != nullptr;
    const size_t lds = Cfg::lds_bytes(bias);
    if (lds > 160 * 1024) return fail(DSW_EUNSUPPORTED, "win_attn_bwd: w=%d head dim %d needs %zu B of LDS", WS, g.d, lds);
    const dim3 grid(g.n_win * g.heads), block(Cfg::THREADS);
    int rc;
    if (lepe) {
        const size_t l2 = (size_t)(3 * Cfg::L * g.d + 1) * sizeof(T) + (size_t)g.d * 10 * sizeof(float) + 16;
        if (l2 > 160 * 1024) return fail(DSW_EUNSUPPORTED, "win_attn_bwd: LePE with w=%d head dim %d needs %zu B of LDS", WS, g.d, l2);
        auto pre = lepe_bwd_pre_kernel<T, WS>;
        if (l2 > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)pre, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
            if (e != hipSuccess) return fail(DSW_ELAUNCH, "win_attn_bwd: LDS attribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(pre, grid, dim3(256), l2, st, g, (const T *)qkv, (const T *)out, (const T *)dout, lepe_w, lepe_b,
                           (T *)dqkv, ws_delta, ws_lepe);
        rc = check_launch("lepe_bwd_pre_kernel");
        if (rc) return rc;
    }
#define DSW_BWD_LAUNCH(B_)                                                                                       \
    do {                                                                                                         \
        auto kern = win_attn_bwd_kernel<T, WS, DP, B_, false>;                                                   \
        if (lds > 48 * 1024) {                                                                                   \
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return fail(DSW_ELAUNCH, "win_attn_bwd: LDS attribute: %s", hipGetErrorString(e)); \
        }                                                                                                        \
        hipLaunchKernelGGL(kern, grid, block, lds, st, g, (const T *)qkv, (const T *)out, (const T *)dout, table, \
                           scale, lepe_w, lepe_b, lse, (T *)dqkv, ws_table, lepe ? ws_delta : (const float *)nullptr); \
    } while (0)
    if (bias) DSW_BWD_LAUNCH(true);
    else DSW_BWD_LAUNCH(false);
#undef DSW_BWD_LAUNCH
    rc = check_launch("win_attn_bwd_kernel");
    if (rc) return rc;
    if (bias) {
        const int T2 = (2 * WS - 1) * (2 * WS - 1);
        hipLaunchKernelGGL(bias_grad_reduce_kernel, dim3((T2 * g.heads + 15) / 16), dim3(256), 0, st, ws_table,
                           g.n_win, g.heads, T2, d_table);
        rc = check_launch("bias_grad_reduce_kernel");
        if (rc) return rc;
    }
    if (lepe) {
        hipLaunchKernelGGL(lepe_grad_reduce_kernel, dim3((g.C * 10 + 255) / 256), dim3(256), 0, st, ws_lepe, g.n_win, g.C,
                           d_lepe_w, d_lepe_b);
        rc = check_launch("lepe_grad_reduce_kernel");
    }
    return rc;
}

int make_geom(const char *who, int n_img, int H, int W, int w, int C, int heads, int layout, AttnGeom &g);

}  // namespace dsw

using namespace dsw;

static size_t table_ws_floats(const AttnGeom &g, int w) { return (size_t)g.n_win * g.heads * (2 * w - 1) * (2 * w - 1); }
static size_t lepe_ws_floats(const AttnGeom &g) { return (size_t)g.n_win * g.C * 10; }
static size_t delta_ws_floats(const AttnGeom &g, int w) { return (size_t)g.n_win * g.heads * w * w; }

extern "C" size_t dsw_win_attn_bwd_workspace(int n_img, int H, int W, int w, int C, int heads, int has_bias,
                                             int has_lepe) {
    AttnGeom g;
    if (make_geom("dsw_win_attn_bwd_workspace", n_img, H, W, w, C, heads, DSW_QKV_CHUNKED, g)) return 0;
    size_t fl = 0;
    if (has_bias) fl += table_ws_floats(g, w);
    if (has_lepe) fl += lepe_ws_floats(g) + delta_ws_floats(g, w);
    return fl * sizeof(float);
}

extern "C" int dsw_win_attn_bwd(const void *qkv, const void *out, const void *dout, int dtype, int layout,
                                int n_img, int H, int W, int w, int C, int heads, const float *bias_table,
                                float scale, const float *lepe_w, const float *lepe_b, const float *lse,
                                void *dqkv, float *d_bias_table, float *d_lepe_w, float *d_lepe_b,
                                void *workspace, size_t workspace_bytes, void *stream) {
    AttnGeom g;
    int rc = make_geom("dsw_win_attn_bwd", n_img, H, W, w, C, heads, layout, g);
    if (rc) return rc;
    DSW_REQUIRE(qkv && out && dout && lse && dqkv, "dsw_win_attn_bwd: null pointer");
    DSW_REQUIRE((bias_table == nullptr) == (d_bias_table == nullptr), "dsw_win_attn_bwd: bias_table and d_bias_table go together");
    DSW_REQUIRE((lepe_w == nullptr) == (d_lepe_w == nullptr) && (lepe_w == nullptr) == (lepe_b == nullptr) &&
                    (lepe_w == nullptr) == (d_lepe_b == nullptr),
                "dsw_win_attn_bwd: the four LePE pointers go together");
    const size_t need = dsw_win_attn_bwd_workspace(n_img, H, W, w, C, heads, bias_table != nullptr, lepe_w != nullptr);
    DSW_REQUIRE(workspace_bytes >= need && (need == 0 || workspace), "dsw_win_attn_bwd: workspace %zu B < %zu B", workspace_bytes, need);
    float *ws_table = reinterpret_cast<float *>(workspace);
    float *ws_lepe = ws_table + (bias_table ? table_ws_floats(g, w) : 0);
    float *ws_delta = ws_lepe + (lepe_w ? lepe_ws_floats(g) : 0);
    hipStream_t st = as_stream(stream);
    const bool big = g.d > 32;
    DSW_DISPATCH_DTYPE(dtype, T, {
        if (w == 16) return big ? launch_bwd<T, 16, 64>(g, qkv, out, dout, bias_table, scale, lepe_w, lepe_b, lse, dqkv, d_bias_table, d_lepe_w, d_lepe_b, ws_table, ws_lepe, ws_delta, st)
                                : launch_bwd<T, 16, 32>(g, qkv, out, dout, bias_table, scale, lepe_w, lepe_b, lse, dqkv, d_bias_table, d_lepe_w, d_lepe_b, ws_table, ws_lepe, ws_delta, st);
        return big ? launch_bwd<T, 8, 64>(g, qkv, out, dout, bias_table, scale, lepe_w, lepe_b, lse, dqkv, d_bias_table, d_lepe_w, d_lepe_b, ws_table, ws_lepe, ws_delta, st)
                   : launch_bwd<T, 8, 32>(g, qkv, out, dout, bias_table, scale, lepe_w, lepe_b, lse, dqkv, d_bias_table, d_lepe_w, d_lepe_b, ws_table, ws_lepe, ws_delta, st);
    });
    return DSW_OK;
}
