// Fused window attention, backward: dq, dk, dv, d(bias table) (+ LePE gradients) for one
// (window, head) per workgroup, on the gfx950 matrix cores.
// Reference ops differentiated: DSwinT_arch.py:122-125, :171-172; msft_arch.py:113-136.
//
// Structure.  Each wave owns one 32-key tile of the window and keeps dK^T and dV^T of those keys
// in accumulators for the whole sweep, with its K and V fragments in registers.  The workgroup
// walks the window's 32-query blocks; a block's Q and dO rows are staged in LDS (row-major; the
// transposed operands of the bf16 path come from transposing LDS reads).  With the KEY on the MFMA
// lane, S = Q K^T and dP = dO V^T come out as (query = row = register, key = column = lane), which is
// already the B operand of dV^T += dO^T P and dK^T += Q^T dS (they sum over the accumulator's row
// index).  Only dS crosses LDS, once per block, for dQ = dS K (16x16 output tiles spread over the
// waves; when there are more waves than tiles the key range of a tile is split over two waves and
// the halves meet in LDS one barrier later).  P is recomputed from the forward's log-sum-exp;
// delta = rowsum(dO * O) comes from the saved output.  The bias-table gradient is accumulated in
// registers along the table's diagonals and added into one private LDS table per wave (program
// order inside a wave => bitwise reproducible), summed in wave order and written as a
// per-(window, head) partial that a second kernel reduces over windows in a fixed order.
#include <stdlib.h>
#include <utility>

#include "attn_common.hpp"

#pragma clang fp contract(fast)

namespace dsw {


template <typename T, int WS, int DP> struct BwdCfg {
    static constexpr bool BF = sizeof(T) == 2;
    static constexpr int L = WS * WS;
    static constexpr int NT = L / 32;                 // key tiles = query blocks
    static constexpr int NW = NT;                     // waves: one key tile each
    static constexpr int THREADS = NW * 64;
    static constexpr int QS = DP + lds_traits<T>::RPAD;  // row-major stride of Qb / dOb (and K, f32)
    static constexpr int KTS = L + 8;                     // stride of K^T (bf16) [DP][L]
    static constexpr int DSS = 48;                        // stride of the bf16 dS^T image [L keys][32 queries]
    static constexpr int SS = L + 4;                      // stride of the f32 dS block [32][L]
    static constexpr int QTILES = (32 / 16) * (DP / 16);  // 16x16 dQ tiles per block
    static constexpr int KSPLIT = NW > QTILES ? NW / QTILES : 1;  // waves sharing one dQ tile (key halves)
    static constexpr int QT_PER_WAVE = NW >= QTILES ? 1 : QTILES / NW;
    static_assert(KSPLIT <= 2, "a dQ tile is split over at most two waves");
    // element counts (in T) of the LDS regions
    static constexpr int N_K = BF ? DP * KTS : L * QS;
    static constexpr int N_DS0 = BF ? L * DSS : 32 * SS;
    static constexpr int N_EPI = BF ? NW * 32 * QS : 0;  // the bf16 epilogue's per-wave row slices reuse the dS region
    static constexpr int N_DS = N_DS0 > N_EPI ? N_DS0 : N_EPI;
    static constexpr int N_QB = 32 * QS;
    static constexpr bool DBUF = BF;  // two staging buffers for the Q / dO block: one barrier less per block
    static constexpr size_t lds_bytes(bool bias) {
        return (size_t)(N_K + N_DS + (DBUF ? 4 : 2) * N_QB) * sizeof(T) + (size_t)2 * L * sizeof(float) +
               (KSPLIT > 1 ? (size_t)QTILES * 64 * 4 * sizeof(float) : 0) +
               (bias ? (size_t)(bias_lds<WS>::FLOATS + (WS == 16 ? NW * BGW_ROWS * BGW_STRIDE : 0)) * sizeof(float) : 0);
    }
    static constexpr int MIN_WAVES = (WS == 16 && BF && DP == 32) ? 4 : (WS == 16 ? 2 : 2);
};

// ---- d(bias table) for 16 x 16 windows: diagonal sums in registers ---------------------------------
// In a dS tile (query = row = register, key = column = lane) the table entry of an element is
// (di, dj) = (qi - ki, qj - kj).  Rotating each register's 16-lane rows left by its qj puts every
// element with the same dj in the same lane, so the 8 registers of one query row-group add up
// element-wise; lanes whose rotation wrapped belong to dj + 16 and are kept apart by a lane mask.
// Register R holds qj = QJ_LO(R) in lanes 0-31 and QJ_LO(R) + 4 in lanes 32-63: every register is
// rotated by QJ_LO only (one DPP operand, the same for all four lane rows) and the upper half's
// sums get their missing rotation by 4 once, at the end (rotations commute with the sums).
template <int R>
__device__ __forceinline__ void diag_step(const f32x16 &ds, const bool (&wrap)[8], float (&all)[2], float (&wrapped)[2]) {
    constexpr int QJ_LO = (R & 3) + 8 * ((R >> 2) & 1), ROR = (16 - QJ_LO) & 15;
    float y = ds[R];
    if constexpr (ROR != 0) y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(ds[R]), 0x120 + ROR, 0xF, 0xF, false));
    all[R >> 3] += y;
    wrapped[R >> 3] += wrap[R & 7] ? y : 0.f;  // the lane masks live in scalar registers
}
template <int... R>
__device__ __forceinline__ void diag_sums(std::integer_sequence<int, R...>, const f32x16 &ds, const bool (&wrap)[8],
                                          float (&all)[2], float (&wrapped)[2]) {
    (diag_step<R>(ds, wrap, all, wrapped), ...);
}
template <typename T, int WS, int DP, bool BIAS, bool PAIR, bool DROP>
__global__ __launch_bounds__((BwdCfg<T, WS, DP>::THREADS), (BwdCfg<T, WS, DP>::MIN_WAVES)) void win_attn_bwd_kernel(
    AttnGeom g, const T *__restrict__ qkv, const T *__restrict__ out, const T *__restrict__ dout,
    const float *__restrict__ table, float scale, const float *__restrict__ lse, T *__restrict__ dqkv,
    float *__restrict__ ws_table, const float *__restrict__ delta_in, DropArgs drop) {
    using Cfg = BwdCfg<T, WS, DP>;
    constexpr bool BF = Cfg::BF;
    static_assert(BF || !PAIR, "the pair path moves two bf16 per lane");
    constexpr int L = Cfg::L, NT = Cfg::NT, NW = Cfg::NW, THREADS = Cfg::THREADS;
    constexpr int QS = Cfg::QS, KTS = Cfg::KTS, DSS = Cfg::DSS, SS = Cfg::SS;
    constexpr int QTILES = Cfg::QTILES, KSPLIT = Cfg::KSPLIT, QTW = Cfg::QT_PER_WAVE;
    constexpr int TS = bias_lds<WS>::TS, TBF = bias_lds<WS>::FLOATS, TR = bias_lds<WS>::ROWS;
    constexpr int BGW_FLOATS = BGW_ROWS * BGW_STRIDE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *Kl = reinterpret_cast<T *>(smem);   // bf16: K^T [DP][KTS];  f32: K [L][QS]
    T *dSs = Kl + Cfg::N_K;                // bf16: dS^T [L][DSS], 8-byte chunks swizzled;  f32: dS [32][SS]
    T *Qb = dSs + Cfg::N_DS;               // [32][QS]  (DBUF: second pair of buffers 2*N_QB further on)
    T *dOb = Qb + Cfg::N_QB;               // [32][QS]
    constexpr bool DBUF = Cfg::DBUF;
    constexpr int BUF = 2 * Cfg::N_QB;     // elements from one (Qb, dOb) pair to the other
    float *ndelta_s = reinterpret_cast<float *>(Qb + (DBUF ? 4 : 2) * Cfg::N_QB);  // [L]  -delta
    float *nlse_s = ndelta_s + L;                                   // [L]  -lse * log2(e)
    float *qpart = nlse_s + L;                                      // KSPLIT = 2: [QTILES][64] float4
    float *tb = qpart + (KSPLIT > 1 ? QTILES * 64 * 4 : 0);         // BIAS: [TBF]
    float *bgw = tb + TBF;                                          // BIAS, w = 16: [NW][BGW_FLOATS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int id = xcd_remap(blockIdx.x, g.n_win * g.heads);
    const int win = __builtin_amdgcn_readfirstlane(id / g.heads), head = id - win * g.heads;  // (wave-uniform quotients of the vector unit's division, handed back to scalar registers)
    const int d = g.d, C = g.ldo, CG = g.ldg, C3 = g.ldq;  // row strides of out, dout and qkv / dqkv
    int qoff, koff, voff;
    qkv_offsets(g, head, qoff, koff, voff);
    size_t origin;
    int pitch;
    if (g.H == 0) {
        origin = (size_t)win * L;
        pitch = WS;
    } else {
        const int per = g.nWh * g.nWw;
        const int b = __builtin_amdgcn_readfirstlane(win / per), r = win - b * per;
        const int wr = __builtin_amdgcn_readfirstlane(r / g.nWw), wc = r - wr * g.nWw;
        origin = ((size_t)b * g.H + wr * WS) * g.W + wc * WS;
        pitch = g.W;
    }
    if (window_is_off(g, win)) {  // uniform over the workgroup: zero gradients, nothing read
        zero_window_slice<T>(dqkv, C3, origin, pitch, WS, qoff, d, false, 0, tid, THREADS);
        zero_window_slice<T>(dqkv, C3, origin, pitch, WS, koff, d, false, 0, tid, THREADS);
        zero_window_slice<T>(dqkv, C3, origin, pitch, WS, voff, d, head == g.heads - 1, 3 * g.C, tid, THREADS);
        if constexpr (BIAS) {
            constexpr int TRR = 2 * WS - 1;
            for (int it = tid; it < TRR * TRR; it += THREADS) ws_table[((size_t)win * g.heads + head) * (TRR * TRR) + it] = 0.f;
        }
        return;
    }
    auto trow = [&](int l) -> size_t { return origin + (size_t)(l / WS) * pitch + (l % WS); };
    const float sc2 = scale * LOG2E;
    auto bits = [](T v) -> uint32_t {
        if constexpr (BF) return (uint32_t)v;
        else return __float_as_uint(v);
    };
    // staging geometry: a thread keeps one element (pair) column and a token row of each sweep of
    // RS consecutive tokens; loads are issued in batches before any of them is consumed
    constexpr int EPT = PAIR ? 2 : 1;             // elements per thread item
    constexpr int DW = DP / EPT, RS = THREADS / DW;  // items per row, rows per sweep
    const int r0 = tid / DW, e = (tid - r0 * DW) * EPT;
    const int ec = min(e, d - EPT);               // clamped: padding lanes re-read valid data
    const bool eok = e < d;
    const size_t hd = (size_t)head * d;

    // ---- prologue: K into LDS, -delta = -rowsum(dO * O), -lse, bias tables ------------------------
    {
        constexpr int ITER = L / RS, BATCH = ITER < 8 ? ITER : 8;
        for (int i0 = 0; i0 < ITER; i0 += BATCH) {
            uint32_t kv[BATCH], ov[BATCH], gv[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const size_t tr = trow((i0 + i) * RS + r0);
                if constexpr (PAIR) {
                    kv[i] = *reinterpret_cast<const uint32_t *>(qkv + tr * C3 + koff + ec);
                    ov[i] = *reinterpret_cast<const uint32_t *>(out + tr * C + hd + ec);
                    gv[i] = *reinterpret_cast<const uint32_t *>(dout + tr * CG + hd + ec);
                } else {
                    kv[i] = bits(qkv[tr * C3 + koff + ec]);
                    ov[i] = bits(out[tr * C + hd + ec]);
                    gv[i] = bits(dout[tr * CG + hd + ec]);
                }
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int row = (i0 + i) * RS + r0;
                float part;
                if constexpr (PAIR) {
                    const uint32_t k2 = eok ? kv[i] : 0u;
                    Kl[e * KTS + row] = (T)(k2 & 0xffffu);
                    Kl[(e + 1) * KTS + row] = (T)(k2 >> 16);
                    part = eok ? __uint_as_float(ov[i] << 16) * __uint_as_float(gv[i] << 16) +
                                     __uint_as_float(ov[i] & 0xffff0000u) * __uint_as_float(gv[i] & 0xffff0000u)
                               : 0.f;
                } else if constexpr (BF) {
                    Kl[e * KTS + row] = eok ? (T)kv[i] : (T)0;
                    part = eok ? bf16_to_f32((T)ov[i]) * bf16_to_f32((T)gv[i]) : 0.f;
                } else {
                    Kl[row * QS + e] = eok ? __uint_as_float(kv[i]) : 0.f;
                    part = eok ? __uint_as_float(ov[i]) * __uint_as_float(gv[i]) : 0.f;
                }
#pragma unroll
                for (int o = 1; o < DW; o <<= 1) part += __shfl_xor(part, o, 64);
                if (e == 0) ndelta_s[row] = -part;
            }
        }
    }
    for (int row = tid; row < L; row += THREADS) nlse_s[row] = -lse[((size_t)win * g.heads + head) * L + row] * LOG2E;
    if (delta_in != nullptr) {
        // LePE: `out` holds SDPA + LePE, so rowsum(dO * O_sdpa) was computed by the LePE pre-pass
        __syncthreads();
        for (int row = tid; row < L; row += THREADS) ndelta_s[row] = -delta_in[((size_t)win * g.heads + head) * L + row];
    }
    if constexpr (BIAS) {
        for (int it = tid; it < TR * TR; it += THREADS) {
            const int di = it / TR, dj = it - di * TR;
            tb[di * TS + dj] = table[(size_t)it * g.heads + head] * LOG2E;
        }
    }
    // true where the full rotation of a register's lane row (by qj = QJ_LO + 4 hh) wraps around (those
    // elements belong to dj + 16), seen after the partial rotation by QJ_LO = (j & 3) + 8 (j >> 2):
    // lane m then holds kj = (m + QJ_LO) mod 16, and it wrapped if kj < qj.  Index = register & 7.
    bool wrapf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int qlo = (j & 3) + 8 * (j >> 2);
        wrapf[j] = (((lane & 15) + qlo) & 15) < qlo + 4 * hh;
    }

    // this wave's K and V fragments (B operands: k-dim = head dim, column = key on the lane)
    const int tile = wave;                    // key tile of this wave
    const int key = 32 * tile + l31;
    uint4 kfb[BF ? DP / 16 : 1], vfb[BF ? DP / 16 : 1];
    float kff[BF ? 1 : DP / 2], vff[BF ? 1 : DP / 2];
    {
        const T *krow = qkv + trow(key) * C3 + koff;
        const T *vrow = qkv + trow(key) * C3 + voff;
        if constexpr (BF) {
#pragma unroll
            for (int s = 0; s < DP / 16; ++s) {
                kfb[s] = load_bf16x8_masked(krow, 16 * s + 8 * hh, d);
                vfb[s] = load_bf16x8_masked(vrow, 16 * s + 8 * hh, d);
            }
        } else {
#pragma unroll
            for (int s = 0; s < DP / 2; ++s) {
                const int ee = hh * (DP / 2) + s;
                kff[s] = (ee < d) ? krow[ee] : 0.f;
                vff[s] = (ee < d) ? vrow[ee] : 0.f;
            }
        }
    }

    f32x16 dKacc[DP / 32], dVacc[DP / 32];
#pragma unroll
    for (int u = 0; u < DP / 32; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dKacc[u][r] = 0.f;
            dVacc[u][r] = 0.f;
        }

    // ---- the next block's Q and dO rows travel in registers while the current block computes ----
    constexpr int NI = 32 / RS > 0 ? 32 / RS : 1;  // sweeps per 32-row block
    static_assert(32 % RS == 0 || RS > 32, "block sweeps");
    const bool stager = RS <= 32 || r0 < 32;       // with more than 32 rows per sweep only the first 32 take part
    uint32_t nq[NI], ng[NI];
    auto load_block = [&](int qb) {
        if (!stager) return;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const size_t tr = trow(qb * 32 + i * RS + r0);
            if constexpr (PAIR) {
                nq[i] = *reinterpret_cast<const uint32_t *>(qkv + tr * C3 + qoff + ec);
                ng[i] = *reinterpret_cast<const uint32_t *>(dout + tr * CG + hd + ec);
            } else {
                nq[i] = bits(qkv[tr * C3 + qoff + ec]);
                ng[i] = bits(dout[tr * CG + hd + ec]);
            }
        }
    };
    // Q is stored times scale*log2(e): S comes out of the MFMA in the base-2 domain, and dK picks
    // up the inverse factor once, in the epilogue.
    auto store_block = [&](int buf) {
        if (!stager) return;
        T *Qb = dSs + Cfg::N_DS + buf * BUF, *dOb = Qb + Cfg::N_QB;  // shadow: the pair being filled
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = i * RS + r0;
            if constexpr (PAIR) {
                const uint32_t q2 = eok ? pack_bf16(__uint_as_float(nq[i] << 16) * sc2, __uint_as_float(nq[i] & 0xffff0000u) * sc2) : 0u;
                *reinterpret_cast<uint32_t *>(Qb + row * QS + e) = q2;
                *reinterpret_cast<uint32_t *>(dOb + row * QS + e) = eok ? ng[i] : 0u;
            } else if constexpr (BF) {
                Qb[row * QS + e] = eok ? f32_to_bf16(bf16_to_f32((T)nq[i]) * sc2) : (T)0;
                dOb[row * QS + e] = eok ? (T)ng[i] : (T)0;
            } else {
                Qb[row * QS + e] = eok ? __uint_as_float(nq[i]) * sc2 : 0.f;
                dOb[row * QS + e] = eok ? __uint_as_float(ng[i]) : 0.f;
            }
        }
    };

    // lane geometry of the transposing reads (bf16): lane 4q+p of a 16-lane group passes row q, columns 4p..
    const int trq = (lane >> 2) & 3, trp = lane & 3, trg = (lane >> 4) & 1;
    const T *qtr = Qb + (4 * hh + trq) * QS + 16 * trg + 4 * trp;   // Q^T / dO^T operands: rows = queries
    const T *gtr = dOb + (4 * hh + trq) * QS + 16 * trg + 4 * trp;
    // dQ work unit of this wave: 16x16 tile(s) and key range
    const int l15 = lane & 15, l4 = lane >> 4;
    const int qtile0 = KSPLIT > 1 ? wave % QTILES : wave * QTW, khalf = KSPLIT > 1 ? wave / QTILES : 0;
    constexpr int KEYS_PER_UNIT = L / KSPLIT;
    f32x4 qkeep[QTW];  // KSPLIT = 2: the lower key half's partial dQ waits here for the upper half
    auto finish_dq = [&](int qb, const f32x4(&qa)[QTW], bool add_partner) {
#pragma unroll
        for (int tq = 0; tq < QTW; ++tq) {
            const int t16 = qtile0 + tq;
            const int mq = t16 / (DP / 16), nd = t16 - mq * (DP / 16);
            f32x4 v = qa[tq];
            if constexpr (KSPLIT > 1) {
                if (add_partner) {
                    const float4 pp = *reinterpret_cast<const float4 *>(qpart + (t16 * 64 + lane) * 4);
                    v[0] += pp.x, v[1] += pp.y, v[2] += pp.z, v[3] += pp.w;
                }
            }
            // C layout: column (head dim) = lane & 15, row (query) = 4*(lane>>4) + reg
            const int ee = 16 * nd + l15;
            if (ee < d) {
#pragma unroll
                for (int r = 0; r < 4; ++r) io<T>::st(dqkv + trow(qb * 32 + 16 * mq + 4 * l4 + r) * C3 + qoff + ee, v[r] * scale);
            }
        }
    };

    constexpr int BG_N = (2 * WS - 1 + 7) / 8;  // w = 8: bias-gradient entries (di values) per thread
    float bgacc[BG_N];
#pragma unroll
    for (int n = 0; n < BG_N; ++n) bgacc[n] = 0.f;
    const int ki = key / WS, kj = key - ki * WS;

    float bg_carry_a = 0.f, bg_carry_w = 0.f;  // w = 16 bias gradient: (G = 1, kip = 0) sums of the previous block
    load_block(0);
    if constexpr (DBUF) store_block(0);
    for (int qb = 0; qb < NT; ++qb) {
        // previous block's LDS reads are done (first pass: prologue writes land); with two staging
        // buffers this block's Q / dO rows were stored during the previous block and are now visible
        lds_barrier();
        if constexpr (KSPLIT > 1) {
            if (qb > 0 && khalf == 0) finish_dq(qb - 1, qkeep, true);  // the upper key half's partial has landed
        }
        if constexpr (!DBUF) {
            store_block(0);
            lds_barrier();
        }
        if (qb + 1 < NT) load_block(qb + 1);
        const int cur = DBUF ? (qb & 1) * BUF : 0;  // staging pair of this block

        // table entry of (q, key): (qi-ki+w-1)*TS + (qj-kj+w-1); q = qb*32 + rho(r) + 4hh
        const int lane_base = (WS - 1) * (TS + 1) - ki * TS - kj + 4 * hh + (qb * 32 / WS) * TS;
        // accumulators start from the row constants: S' = S*log2e + bias - lse, dP' = dP - delta
        f32x16 sacc, pacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q_c = rho(r);
            float v0 = nlse_s[qb * 32 + q_c + 4 * hh];
            if constexpr (BIAS) v0 += tb[lane_base + (q_c / WS) * TS + (q_c % WS)];
            sacc[r] = v0;
            pacc[r] = ndelta_s[qb * 32 + q_c + 4 * hh];
        }
        // ---- S = Q K^T and dP = dO V^T (rows = queries, lane = key) ---------------------------------
        if constexpr (BF) {
#pragma unroll
            for (int s = 0; s < DP / 16; ++s) {
                const uint4 qa = *reinterpret_cast<const uint4 *>(Qb + cur + l31 * QS + 16 * s + 8 * hh);
                const uint4 ga = *reinterpret_cast<const uint4 *>(dOb + cur + l31 * QS + 16 * s + 8 * hh);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(qa), as_bf16x8(kfb[s]), sacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ga), as_bf16x8(vfb[s]), pacc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s4 = 0; s4 < DP / 8; ++s4) {
                const float4 qa = *reinterpret_cast<const float4 *>(Qb + cur + l31 * QS + hh * (DP / 2) + 4 * s4);
                const float4 ga = *reinterpret_cast<const float4 *>(dOb + cur + l31 * QS + hh * (DP / 2) + 4 * s4);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, kff[4 * s4 + 0], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, kff[4 * s4 + 1], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, kff[4 * s4 + 2], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, kff[4 * s4 + 3], sacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.x, vff[4 * s4 + 0], pacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.y, vff[4 * s4 + 1], pacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.z, vff[4 * s4 + 2], pacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.w, vff[4 * s4 + 3], pacc, 0, 0, 0);
            }
        }
        // ---- P = exp2(S'), dS = P * dP' ------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(sacc[r]);
            if constexpr (DROP) {
                // the forward thinned P after normalising it: O = (keep * P / (1-p)) V.  With m = keep / (1-p):
                // dV takes m * P; dP = m * dO V^T, and pacc holds dO V^T - delta, so dS = P (m (pacc + delta) - delta)
                const int qrow = qb * 32 + rho(r) + 4 * hh;
                const float ndl = ndelta_s[qrow];
                const uint32_t h = drop_hash((((uint32_t)win * g.heads + head) * L + qrow) * (L / 2) + (key >> 1), drop.seed_lo, drop.seed_hi);
                const bool keep = ((key & 1) ? (h >> 16) : (h & 0xffffu)) >= drop.thr;
                const float dp = keep ? (pacc[r] - ndl) * drop.inv_keep : 0.f;
                sacc[r] = keep ? p * drop.inv_keep : 0.f;
                pacc[r] = p * (dp + ndl);
            } else {
                const float ds = p * pacc[r];
                sacc[r] = p;
                pacc[r] = ds;
            }
        }
        // ---- d(bias table), w = 16 ------------------------------------------------------------------------
        if constexpr (BIAS && WS == 16) {
            float all[2] = {0.f, 0.f}, wrapped[2] = {0.f, 0.f};
            diag_sums(std::make_integer_sequence<int, 16>{}, pacc, wrapf, all, wrapped);
#pragma unroll
            for (int G = 0; G < 2; ++G) {  // lanes 0-31 += lanes 32-63 (one VALU swap, no LDS round trip)
                all[G] = upper_rows_rot4(all[G]);
                wrapped[G] = upper_rows_rot4(wrapped[G]);
                all[G] += upper_half_to_lower(all[G]);
                wrapped[G] += upper_half_to_lower(wrapped[G]);
            }
            // Lanes 0-15 now hold key row kip = 0, lanes 16-31 kip = 1 (m = lane & 15 is -dj, or 16 - dj for
            // wrapped elements).  Row-group G of this block belongs to table row di = 2qb + G - kip - 2*tile,
            // the wave's row 2qb + G - kip + 1: the odd row 2qb+1 gets (G=0, kip=0) + (G=1, kip=1) and is
            // complete; the even row 2qb gets (G=0, kip=1) plus (G=1, kip=0) of the previous block, which
            // is carried in a register.  Every entry of the wave's table is thus stored exactly once.
            {
                const float a0 = all[0] - wrapped[0], a1 = all[1] - wrapped[1];
                const float a0k1 = rows_odd_to_even(a0), a1k1 = rows_odd_to_even(a1);
                const float w0k1 = rows_odd_to_even(wrapped[0]), w1k1 = rows_odd_to_even(wrapped[1]);
                if (lane < 16) {
                    float *ent = bgw + wave * BGW_FLOATS + (2 * qb) * BGW_STRIDE + 15 - lane;
                    ent[0] = a0k1 + bg_carry_a;              // row 2qb
                    ent[16] = w0k1 + bg_carry_w;
                    ent[BGW_STRIDE] = a0 + a1k1;             // row 2qb + 1
                    ent[BGW_STRIDE + 16] = wrapped[0] + w1k1;
                }
                bg_carry_a = a1;
                bg_carry_w = wrapped[1];
            }
        }
        if constexpr (BF) {
            // ---- dS^T to LDS for dQ, and dV^T += dO^T P, dK^T += Q^T dS (sum over the accumulator rows) --
            // registers 4g..4g+3 are queries 8g + 4hh + 0..3 of this lane's key: one 8-byte chunk of the
            // key's row; chunk c sits at position (c & 4) | ((c & 3) ^ ((key >> 2) & 3)), which keeps
            // both these writes and the transposing reads below off each other's banks
            T *dsrow = dSs + key * DSS;
            const int fk = (l31 >> 2) & 3;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uint4 pb = make_uint4(pack_bf16(sacc[8 * s + 0], sacc[8 * s + 1]), pack_bf16(sacc[8 * s + 2], sacc[8 * s + 3]),
                                            pack_bf16(sacc[8 * s + 4], sacc[8 * s + 5]), pack_bf16(sacc[8 * s + 6], sacc[8 * s + 7]));
                const uint4 db = make_uint4(pack_bf16(pacc[8 * s + 0], pacc[8 * s + 1]), pack_bf16(pacc[8 * s + 2], pacc[8 * s + 3]),
                                            pack_bf16(pacc[8 * s + 4], pacc[8 * s + 5]), pack_bf16(pacc[8 * s + 6], pacc[8 * s + 7]));
                {
                    const int c0 = 4 * s + hh, c1 = 4 * s + 2 + hh;  // chunks of registers 8s..8s+3 and 8s+4..8s+7
                    *reinterpret_cast<uint2 *>(dsrow + 4 * ((c0 & 4) | ((c0 & 3) ^ fk))) = make_uint2(db.x, db.y);
                    *reinterpret_cast<uint2 *>(dsrow + 4 * ((c1 & 4) | ((c1 & 3) ^ fk))) = make_uint2(db.z, db.w);
                }
#pragma unroll
                for (int u = 0; u < DP / 32; ++u) {
                    // A operands dO^T, Q^T (row = head dim 32u + l31): element j is query
                    // 16s + 8(j>>2) + 4hh + (j&3), fetched as two transposing reads of the row-major block
                    const uint2 g0 = lds_read_tr16(gtr + cur + (16 * s) * QS + 32 * u), g1 = lds_read_tr16(gtr + cur + (16 * s + 8) * QS + 32 * u);
                    const uint2 q0 = lds_read_tr16(qtr + cur + (16 * s) * QS + 32 * u), q1 = lds_read_tr16(qtr + cur + (16 * s + 8) * QS + 32 * u);
                    dVacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(g0.x, g0.y, g1.x, g1.y)), as_bf16x8(pb), dVacc[u], 0, 0, 0);
                    dKacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(q0.x, q0.y, q1.x, q1.y)), as_bf16x8(db), dKacc[u], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) dSs[(rho(r) + 4 * hh) * SS + key] = pacc[r];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int u = 0; u < DP / 32; ++u) {
                    const float ga = dOb[cur + (rho(r) + 4 * hh) * QS + 32 * u + l31];
                    const float qa = Qb[cur + (rho(r) + 4 * hh) * QS + 32 * u + l31];
                    dVacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, sacc[r], dVacc[u], 0, 0, 0);
                    dKacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa, pacc[r], dKacc[u], 0, 0, 0);
                }
            }
        }
        if constexpr (DBUF) {
            if (qb + 1 < NT) store_block((qb + 1) & 1);  // nobody reads that pair any more
        }
        lds_barrier();  // the whole dS block (and the next block's rows) is in LDS

        // element (query q of the block, key k) of the dS image
        auto ds_at = [&](int q, int k) -> float {
            if constexpr (BF) {
                const int c = q >> 2;
                return bf16_to_f32(dSs[k * DSS + 4 * ((c & 4) | ((c & 3) ^ ((k >> 2) & 3))) + (q & 3)]);
            } else {
                return dSs[q * SS + k];
            }
        };
        // ---- d(bias table), w = 8: every thread owns a few table entries (di, dj) and sums the dS
        // elements of this block that map to them, straight from the dS image - a gather in a fixed
        // order, so no atomics and bitwise reproducible.
        if constexpr (BIAS && WS != 16) {
            const int dj = (tid % (2 * WS)) - (WS - 1);   // lane WS*2-1 of each group is idle
            const int dig = tid / (2 * WS);                // groups of di values
            if (dj < WS) {
#pragma unroll
                for (int n = 0; n < BG_N; ++n) {
                    const int di = dig + (THREADS / (2 * WS)) * n - (WS - 1);
                    if (di < WS) {
                        float acc = 0.f;
#pragma unroll
                        for (int qr = 0; qr < 32 / WS; ++qr) {
                            const int ki2 = qb * (32 / WS) + qr - di;
                            if (ki2 >= 0 && ki2 < WS) {
#pragma unroll
                                for (int qj = 0; qj < WS; ++qj) {
                                    const int kj2 = qj - dj;
                                    if (kj2 >= 0 && kj2 < WS) acc += ds_at(qr * WS + qj, ki2 * WS + kj2);
                                }
                            }
                        }
                        bgacc[n] += acc;
                    }
                }
            }
        }

        // ---- dQ block = dS K * scale: 16x16 output tiles, K-dim = this wave's key range ------------------
        f32x4 qacc[QTW];
#pragma unroll
        for (int tq = 0; tq < QTW; ++tq) {
            const int t16 = qtile0 + tq;
            const int mq = t16 / (DP / 16), nd = t16 - mq * (DP / 16);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) {
                // k-slot j of lane group l4 in a 32-key step is key 4*l4 + (j&3) + 16*(j>>2): a half wave's
                // two transposing reads then cover 8 consecutive key rows (conflict-free at DSS = 48)
                const T *abase = dSs + (khalf * KEYS_PER_UNIT + 4 * l4 + trq) * DSS + 4 * (4 * mq + (trp ^ l4));
                const T *bbase = Kl + (16 * nd + l15) * KTS + khalf * KEYS_PER_UNIT + 4 * l4;
#pragma unroll 4
                for (int s = 0; s < KEYS_PER_UNIT / 32; ++s) {
                    const uint2 a0 = lds_read_tr16(abase + (32 * s) * DSS), a1 = lds_read_tr16(abase + (32 * s + 16) * DSS);
                    const uint2 b0 = *reinterpret_cast<const uint2 *>(bbase + 32 * s), b1 = *reinterpret_cast<const uint2 *>(bbase + 32 * s + 16);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(make_uint4(a0.x, a0.y, a1.x, a1.y)),
                                                                  as_bf16x8(make_uint4(b0.x, b0.y, b1.x, b1.y)), acc, 0, 0, 0);
                }
            } else {
#pragma unroll 4
                for (int s = 0; s < KEYS_PER_UNIT / 4; ++s) {
                    const int k0 = khalf * KEYS_PER_UNIT + 4 * s + l4;
                    const float a = dSs[(16 * mq + l15) * SS + k0];
                    const float b = Kl[k0 * QS + 16 * nd + l15];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
                }
            }
            qacc[tq] = acc;
        }
        if constexpr (KSPLIT > 1) {
            if (khalf == 0) {
#pragma unroll
                for (int tq = 0; tq < QTW; ++tq) qkeep[tq] = qacc[tq];
            } else {
#pragma unroll
                for (int tq = 0; tq < QTW; ++tq)
                    *reinterpret_cast<float4 *>(qpart + ((qtile0 + tq) * 64 + lane) * 4) = make_float4(qacc[tq][0], qacc[tq][1], qacc[tq][2], qacc[tq][3]);
            }
        } else {
            finish_dq(qb, qacc, false);
        }
    }
    if constexpr (BIAS && WS == 16) {
        if (lane < 16) {  // the last even row of the wave's table
            float *ent = bgw + wave * BGW_FLOATS + (2 * NT) * BGW_STRIDE + 15 - lane;
            ent[0] = bg_carry_a;
            ent[16] = bg_carry_w;
        }
    }
    __syncthreads();  // the last block's LDS reads are done, the upper halves' partials have landed
    if constexpr (KSPLIT > 1) {
        if (khalf == 0) finish_dq(NT - 1, qkeep, true);
    }

    if (head == g.heads - 1 && C3 > 3 * g.C) {  // the padding columns of this window's dqkv rows
        const int npad = C3 - 3 * g.C;
        for (int i = tid; i < L * npad; i += THREADS) dqkv[trow(i / npad) * C3 + 3 * g.C + i % npad] = (T)0;
    }
    // ---- epilogue: dK^T, dV^T (rows = head dim in registers, lane = key) ------------------------
    if constexpr (PAIR) {
        // each wave turns its tiles into [key][head dim] rows in a private slice of the (now free)
        // dS region, so that the global stores are runs of whole rows instead of one dword in each
        // of 32 rows; a wave's LDS accesses complete in order and nobody else touches its slice
        T *slice = dSs + wave * (32 * QS);
        constexpr int DWR = DP / 2, RPI = 64 / DWR;
        const int dw = lane % DWR, rsub = lane / DWR;
#pragma unroll
        for (int which = 0; which < 2; ++which) {  // 0: dK, 1: dV
#pragma unroll
            for (int u = 0; u < DP / 32; ++u) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = which == 0 ? dKacc[u][4 * g4 + j] * LN2 : dVacc[u][4 * g4 + j];
                    *reinterpret_cast<uint2 *>(slice + l31 * QS + 32 * u + 8 * g4 + 4 * hh) =
                        make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
                }
            }
#pragma unroll
            for (int i = 0; i < 32 / RPI; ++i) {
                const int row = i * RPI + rsub;
                uint32_t v = *reinterpret_cast<const uint32_t *>(slice + row * QS + 2 * dw);
                if (2 * dw < d) {
                    T *dst = dqkv + trow(32 * tile + row) * C3 + (which == 0 ? koff : voff) + 2 * dw;
                    if (which == 1 && delta_in != nullptr) {  // add the LePE share of dV left there by the pre-pass
                        const uint32_t old = *reinterpret_cast<const uint32_t *>(dst);
                        v = pack_bf16(__uint_as_float(v << 16) + __uint_as_float(old << 16),
                                      __uint_as_float(v & 0xffff0000u) + __uint_as_float(old & 0xffff0000u));
                    }
                    *reinterpret_cast<uint32_t *>(dst) = v;
                }
            }
        }
    } else {
        T *dkrow = dqkv + trow(key) * C3 + koff;
        T *dvrow = dqkv + trow(key) * C3 + voff;
#pragma unroll
        for (int u = 0; u < DP / 32; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ee = 32 * u + rho(r) + 4 * hh;
                if (ee < d) {
                    float v0 = dVacc[u][r];
                    if (delta_in != nullptr) v0 += io<T>::ld(dvrow + ee);
                    io<T>::st(dkrow + ee, dKacc[u][r] * LN2);
                    io<T>::st(dvrow + ee, v0);
                }
            }
        }
    }

    if constexpr (BIAS) {
        // the table entered the logits times log2(e) inside exp2, so d/dtable is dS itself
        float *dst = ws_table + ((size_t)win * g.heads + head) * (TR * TR);
        if constexpr (WS == 16) {
            // entry (di, dj): wave wv holds it in row di + 2*wv + 1 of its table when that row exists
            for (int it = tid; it < TR * TR; it += THREADS) {
                const int di = it / TR - (WS - 1), dj = it - (it / TR) * TR;
                float acc = 0.f;
#pragma unroll
                for (int wv = 0; wv < NW; ++wv) {
                    const int row = di + 2 * wv + 1;
                    if (row >= 0 && row < BGW_ROWS) acc += bgw[wv * BGW_FLOATS + row * BGW_STRIDE + dj];
                }
                dst[it] = acc;
            }
        } else {
            const int dj = (tid % (2 * WS)) - (WS - 1);
            const int dig = tid / (2 * WS);
            if (dj < WS) {
#pragma unroll
                for (int n = 0; n < BG_N; ++n) {
                    const int di = dig + (THREADS / (2 * WS)) * n - (WS - 1);
                    if (di < WS) dst[(di + WS - 1) * TR + dj + WS - 1] = bgacc[n];
                }
            }
        }
    }
}

// d_table[t][h] = sum over windows of ws[(win*heads + h)*T + t], in a fixed order: 64 window groups (window
// index mod 64) each summed by one thread with four loads in flight, then the 64 partials in group order.
// A block takes 16 consecutive entries t of one head: each read instruction covers 64-byte runs of ws.
__global__ __launch_bounds__(1024) void bias_grad_reduce_kernel(const float *__restrict__ ws, int n_win, int heads, int T,
                                                                float *__restrict__ d_table) {
    constexpr int GROUPS = 64;
    __shared__ float part[GROUPS][17];
    const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int chunks = (T + 15) / 16;
    const int h = blockIdx.x / chunks, t = (blockIdx.x - h * chunks) * 16 + e;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (t < T) {
        const float *p = ws + (size_t)h * T + t;
        const size_t step = (size_t)heads * T;
        int wdx = grp;
        for (; wdx + 3 * GROUPS < n_win; wdx += 4 * GROUPS) {
            a0 += p[(size_t)wdx * step];
            a1 += p[(size_t)(wdx + GROUPS) * step];
            a2 += p[(size_t)(wdx + 2 * GROUPS) * step];
            a3 += p[(size_t)(wdx + 3 * GROUPS) * step];
        }
        for (; wdx < n_win; wdx += GROUPS) a0 += p[(size_t)wdx * step];
    }
    part[grp][e] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (grp == 0 && t < T) {
        float tot = 0.f;
#pragma unroll 8
        for (int k2 = 0; k2 < GROUPS; ++k2) tot += part[k2][e];
        d_table[(size_t)t * heads + h] = tot;
    }
}

// ---- LePE backward pre-pass (msft_arch.py:113-125, :136): one block per (window, head) -------------
// With out = SDPA(q,k,v) + LePE(v):  delta = rowsum(dO * (out - LePE(v)))   (for the main kernel),
// dV_lepe = conv^T(dO) written into the v slots of dqkv (the main kernel adds its dV on top),
// and per-window partial sums of the depthwise weight / bias gradients.
// In LDS one element holds CH channels of one position: a bf16 pair in 4 bytes when CH = 2.
template <typename T, int CH> struct LepeElem { using type = T; };
template <> struct LepeElem<bf16_t, 2> { using type = uint32_t; };

// CH floats side by side: with CH = 2 the arithmetic below is packed fp32 (v_pk_fma_f32) on the channel pair
template <int CH> struct LepeVec { typedef float type __attribute__((ext_vector_type(CH))); };

// a window row of WS positions (16-byte aligned when the element has 4 bytes) as floats
template <typename T, int WS, int CH, typename E>
__device__ __forceinline__ void lepe_row(const E *p, typename LepeVec<CH>::type (&x)[WS]) {
    if constexpr (sizeof(E) == 4) {
#pragma unroll
        for (int j0 = 0; j0 < WS; j0 += 4) {
            const uint4 q = *reinterpret_cast<const uint4 *>(p + j0);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (CH == 2) {
                    x[j0 + u][0] = __uint_as_float(w[u] << 16);
                    x[j0 + u][1] = __uint_as_float(w[u] & 0xffff0000u);
                } else {
                    x[j0 + u][0] = __uint_as_float(w[u]);
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < WS; ++j) x[j][0] = io<T>::ld(p + j);
    }
}

// sum over the lanes that are a multiple of WS apart; lanes 0 .. WS-1 hold the total
template <int WS> __device__ __forceinline__ float sum_lanes_ws_apart(float x) {
    if constexpr (WS == 8) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xF, 0xF, false));  // row_ror:8
    x += rows_odd_to_even(x);
    x += upper_half_to_lower(x);
    return x;
}

// geometry of the pre-pass that host and kernel share
template <int WS, int CH> struct LepePre {
    static constexpr int L = WS * WS;
    static constexpr int LP = (WS + 2) * WS;  // V and dO: a row of zeros above and below the window
    static constexpr int PP = LP + 4;         // positions per channel row in LDS; the + 4 spreads the 16-byte reads
    static constexpr int PO = L + 4;          //   of the lanes that differ in the channel over the banks
    static constexpr int NPASS = WS / (4 * CH);  // items = (d / CH) * WS <= (64 / CH) * WS = NPASS * 256
    static_assert(NPASS >= 1, "items of one pass");
};

template <typename T, int WS, int CH>
__global__ __launch_bounds__(256) void lepe_bwd_pre_kernel(AttnGeom g, const T *__restrict__ qkv, const T *__restrict__ out,
                                                           const T *__restrict__ dout, const float *__restrict__ lepe_w,
                                                           const float *__restrict__ lepe_b, T *__restrict__ dqkv,
                                                           float *__restrict__ delta_out, float *__restrict__ ws_lepe, int wpb,
                                                           int nw) {
    // One block per (group of wpb consecutive windows, head); nw of the group's windows are in LDS at a time (nw > 1
    // when the items of one window fill only a quarter or half of the block).
    // A work item is (CH channels, window row i): it holds row i of dO and walks the three neighbouring rows of V and
    // dO once, which gives, from 7 row reads, the LePE values of its row (for delta), dV_lepe of its row, and its
    // share of the 9 tap + 1 bias gradients.  LDS holds [channel group][position], so a row is one or two 16-byte
    // reads at a fixed offset from the item's base address, and V and dO carry a row of zeros above and below the
    // window (no row predicates).
    // delta sums over channels (lanes WS apart, then the waves through LDS); the weight gradient sums over the
    // group's windows (in registers when one pass covers all items) and over rows: ONE partial row per group
    // leaves the block.
    using G = LepePre<WS, CH>;
    using E = typename LepeElem<T, CH>::type;
    constexpr int L = G::L, LP = G::LP, PP = G::PP, PO = G::PO, NPASS = G::NPASS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int d = g.d, dc = d / CH, C = g.C, C3 = g.ldq;
    E *Vs = reinterpret_cast<E *>(smem);  // [nw][dc][PP]
    E *Gs = Vs + nw * dc * PP;            // dO
    E *Os = Gs + nw * dc * PP;            // out, then dV_lepe: [nw][dc][PO]
    float *wl = reinterpret_cast<float *>(Os + nw * dc * PO);  // [d][10]: 9 taps + bias
    float *wsum = wl + d * 10;                                  // [d][10] gradient sums
    float *Pw = wsum + d * 10;                                  // [NPASS * 4][L] per-wave sums of delta
    long long *org = reinterpret_cast<long long *>(Pw + NPASS * 4 * L);  // [4] first token row of each window in LDS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = blockIdx.x / g.heads, head = blockIdx.x - grp * g.heads;
    const int n_items = dc * WS;
    const int ipw = 256 / nw;                       // threads per window (nw > 1 only when NPASS == 1)
    const int slot = tid / ipw, it0 = tid - slot * ipw;
    int qoff, koff, voff;
    qkv_offsets(g, head, qoff, koff, voff);
    const int pitch = g.H == 0 ? WS : g.W;
    for (int it = tid; it < d * 10; it += 256) {
        const int e = it / 10, k = it - e * 10;
        wl[it] = k < 9 ? lepe_w[(size_t)(head * d + e) * 9 + k] : lepe_b[head * d + e];
        wsum[it] = 0.f;
    }
    for (int it = tid; it < nw * dc * 2 * WS; it += 256) {  // the rows of zeros
        const int ch = it / (2 * WS), r = it - ch * 2 * WS;
        const int off = ch * PP + (r < WS ? r : LP - 2 * WS + r);
        Vs[off] = E(0);
        Gs[off] = E(0);
    }
    __syncthreads();
    using V = typename LepeVec<CH>::type;
    V we[10], wacc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            wacc[k][c] = 0.f;
            if constexpr (NPASS == 1) we[k][c] = wl[(CH * min(it0 / WS, dc - 1) + c) * 10 + k];
        }
    const int win_end = min(g.n_win, (grp + 1) * wpb);
    for (int w0 = grp * wpb; w0 < win_end; w0 += nw) {
        const int n_act = min(nw, win_end - w0);
        if (tid < n_act) {
            const int win = w0 + tid;
            if (g.H == 0) {
                org[tid] = (long long)win * L;
            } else {
                const int per = g.nWh * g.nWw;
                const int b = win / per, r = win - b * per;
                const int wr = r / g.nWw, wc = r - wr * g.nWw;
                org[tid] = ((long long)b * g.H + wr * WS) * g.W + wc * WS;
            }
        }
        __syncthreads();  // the previous windows' readers are done
        auto trow = [&](int rw) -> size_t {  // rw = slot * L + position in the window
            const int l = rw & (L - 1);
            return (size_t)org[rw / L] + (size_t)(l / WS) * pitch + (l % WS);
        };
        const int n = n_act * L * dc;  // elements of one array to stage, [slot][position][channel group]
        if constexpr (CH == 2) {
            // SB loads of each of the three arrays are in flight per thread before the first LDS write: with few
            // windows per block at a time, the bytes in flight are what hides the HBM latency
            constexpr int SB = 8;
            const int step_row = 256 / dc, step_e = 256 - step_row * dc;
            int rw = tid / dc, e2 = tid - rw * dc;
            for (int base = tid; base < n; base += 256 * SB) {
                uint32_t rv[SB], rg[SB], ro[SB];
                int lv[SB], lo[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    lv[u] = -1;
                    if (base + u * 256 < n) {
                        const size_t tr = trow(rw);
                        const int ch = (rw / L) * dc + e2, l = rw & (L - 1);
                        lv[u] = ch * PP + WS + l;
                        lo[u] = ch * PO + l;
                        rv[u] = *reinterpret_cast<const uint32_t *>(qkv + tr * C3 + voff + 2 * e2);
                        rg[u] = *reinterpret_cast<const uint32_t *>(dout + tr * g.ldg + head * d + 2 * e2);
                        ro[u] = *reinterpret_cast<const uint32_t *>(out + tr * g.ldo + head * d + 2 * e2);
                    }
                    rw += step_row;
                    e2 += step_e;
                    if (e2 >= dc) {
                        e2 -= dc;
                        ++rw;
                    }
                }
#pragma unroll
                for (int u = 0; u < SB; ++u)
                    if (lv[u] >= 0) {
                        Vs[lv[u]] = rv[u];
                        Gs[lv[u]] = rg[u];
                        Os[lo[u]] = ro[u];
                    }
            }
        } else {
            for (int it = tid; it < n; it += 256) {
                const int rw = it / dc, e = it - rw * dc;
                const size_t tr = trow(rw);
                const int ch = (rw / L) * dc + e, l = rw & (L - 1);
                Vs[ch * PP + WS + l] = qkv[tr * C3 + voff + e];
                Gs[ch * PP + WS + l] = dout[tr * g.ldg + head * d + e];
                Os[ch * PO + l] = out[tr * g.ldo + head * d + e];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int ps = 0; ps < NPASS; ++ps) {
            if (ps * 256 >= n_items) break;  // the same for the whole block
            const int item = it0 + ps * 256;
            const int i = item % WS;
            const bool act = item < n_items && slot < n_act;
            const int ep = act ? item / WS : 0;  // the item's channel group
            if constexpr (NPASS > 1) {
#pragma unroll
                for (int k = 0; k < 10; ++k)
#pragma unroll
                    for (int c = 0; c < CH; ++c) we[k][c] = wl[(CH * ep + c) * 10 + k];
            }
            const E *vrow = Vs + (slot * dc + ep) * PP + (i + 1) * WS, *grow = Gs + (slot * dc + ep) * PP + (i + 1) * WS;
            E *orow = Os + (slot * dc + ep) * PO + i * WS;
            // first the V rows (LePE of this row for delta, the tap gradients), then the dO rows (dV_lepe): the two
            // halves share only row i of dO and the taps, which keeps the registers of each half apart
            V gc[WS];
            float part[WS];
            lepe_row<T, WS, CH>(grow, gc);
            if (!act) {  // an idle lane adds nothing to delta and to the gradient sums
#pragma unroll
                for (int j = 0; j < WS; ++j) gc[j] = V(0.f);
            }
            if constexpr (NPASS > 1) {
#pragma unroll
                for (int k = 0; k < 10; ++k) wacc[k] = V(0.f);
            }
#pragma unroll
            for (int j = 0; j < WS; ++j) wacc[9] += gc[j];
            {
                V lp[WS];
#pragma unroll
                for (int j = 0; j < WS; ++j) lp[j] = we[9];
#pragma unroll
                for (int a = -1; a <= 1; ++a) {
                    V x[WS];
                    lepe_row<T, WS, CH>(vrow + a * WS, x);  // V of row i + a
#pragma unroll
                    for (int bb = -1; bb <= 1; ++bb) {
                        const int k = (a + 1) * 3 + bb + 1;
#pragma unroll
                        for (int j = 0; j < WS; ++j)
                            if (j + bb >= 0 && j + bb < WS) {
                                lp[j] += we[k] * x[j + bb];
                                wacc[k] += gc[j] * x[j + bb];
                            }
                    }
                }
                // delta[q] = sum_e dO[q][e] * (out[q][e] - LePE(v)[q][e])
                V o[WS];
                lepe_row<T, WS, CH>(orow, o);
#pragma unroll
                for (int j = 0; j < WS; ++j) {
                    const V t = gc[j] * (o[j] - lp[j]);
                    part[j] = t[0];
                    if constexpr (CH == 2) part[j] += t[1];
                    part[j] = sum_lanes_ws_apart<WS>(part[j]);
                }
            }
            asm volatile("" ::: "memory");
            {
                // dV_lepe[k] = sum_taps W[tap] * dO[k - tap]: dO of row i - a; it takes out's place in LDS (each
                // element of Os is read and written by this item alone)
                V dv[WS];
#pragma unroll
                for (int j = 0; j < WS; ++j) dv[j] = V(0.f);
#pragma unroll
                for (int a = -1; a <= 1; ++a) {
                    V x[WS];
                    if (a != 0) lepe_row<T, WS, CH>(grow - a * WS, x);
#pragma unroll
                    for (int bb = -1; bb <= 1; ++bb) {
                        const int k = (a + 1) * 3 + bb + 1;
#pragma unroll
                        for (int j = 0; j < WS; ++j)
                            if (j - bb >= 0 && j - bb < WS) dv[j] += we[k] * (a != 0 ? x[j - bb] : gc[j - bb]);
                    }
                }
                if (act) {
                    if constexpr (CH == 2) {
#pragma unroll
                        for (int j0 = 0; j0 < WS; j0 += 4)
                            *reinterpret_cast<uint4 *>(orow + j0) =
                                make_uint4(pack_bf16(dv[j0][0], dv[j0][1]), pack_bf16(dv[j0 + 1][0], dv[j0 + 1][1]),
                                           pack_bf16(dv[j0 + 2][0], dv[j0 + 2][1]), pack_bf16(dv[j0 + 3][0], dv[j0 + 3][1]));
                    } else {
#pragma unroll
                        for (int j = 0; j < WS; ++j) io<T>::st(orow + j, dv[j][0]);
                    }
                }
            }
            if (lane < WS) {  // lane == i here
#pragma unroll
                for (int j = 0; j < WS; ++j) Pw[(ps * 4 + wave) * L + lane * WS + j] = part[j];
            }
            if constexpr (NPASS > 1) {  // (one pass: the sums stay in registers until the group's windows are done)
#pragma unroll
                for (int o2 = 1; o2 < WS; o2 <<= 1)
#pragma unroll
                    for (int k = 0; k < 10; ++k)
#pragma unroll
                        for (int c = 0; c < CH; ++c) wacc[k][c] += __shfl_xor(wacc[k][c], o2, 64);
                if (i == 0 && act) {  // the one writer of these channels' sums
#pragma unroll
                    for (int k = 0; k < 10; ++k)
#pragma unroll
                        for (int c = 0; c < CH; ++c) wsum[(CH * ep + c) * 10 + k] += wacc[k][c];
                }
            }
        }
        __syncthreads();
        if constexpr (NPASS == 1) {
            if (tid < n_act * L) {  // nw * L <= 256 (nw > 1 only with 8 x 8 windows)
                const int sl = tid / L, q = tid & (L - 1);
                const int w_first = sl * (ipw >> 6), w_live = (n_items + 63) >> 6;
                float acc = 0.f;
                for (int wv = 0; wv < w_live; ++wv) acc += Pw[(w_first + wv) * L + q];
                delta_out[((size_t)(w0 + sl) * g.heads + head) * L + q] = acc;
            }
        } else {
            const int live_slots = (n_items + 63) >> 6;
            for (int q = tid; q < L; q += 256) {
                float acc = 0.f;
                for (int sl = 0; sl < live_slots; ++sl) acc += Pw[sl * L + q];
                delta_out[((size_t)w0 * g.heads + head) * L + q] = acc;
            }
        }
        for (int it = tid; it < n; it += 256) {
            const int rw = it / dc, e = it - rw * dc;
            const int ch = (rw / L) * dc + e, l = rw & (L - 1);
            if constexpr (CH == 2) *reinterpret_cast<uint32_t *>(dqkv + trow(rw) * C3 + voff + 2 * e) = Os[ch * PO + l];
            else dqkv[trow(rw) * C3 + voff + e] = Os[ch * PO + l];
        }
    }
    if constexpr (NPASS == 1) {
        // rows of a channel group are adjacent lanes; then the windows side by side in the block, one after the other
#pragma unroll
        for (int o2 = 1; o2 < WS; o2 <<= 1)
#pragma unroll
            for (int k = 0; k < 10; ++k)
#pragma unroll
                for (int c = 0; c < CH; ++c) wacc[k][c] += __shfl_xor(wacc[k][c], o2, 64);
        for (int sl = 0; sl < nw; ++sl) {
            if (slot == sl && it0 % WS == 0 && it0 < n_items) {
                const int e0 = CH * (it0 / WS);
#pragma unroll
                for (int k = 0; k < 10; ++k)
#pragma unroll
                    for (int c = 0; c < CH; ++c) wsum[(e0 + c) * 10 + k] += wacc[k][c];
            }
            __syncthreads();
        }
    } else {
        __syncthreads();
    }
    for (int it = tid; it < d * 10; it += 256) ws_lepe[((size_t)grp * C + head * d) * 10 + it] = wsum[it];
}

// windows per block of the pre-pass: enough blocks to fill the chip several times over, few enough partial rows for
// the reduction to be short; a multiple of the nw windows a block holds in LDS at a time
static int lepe_wpb(const AttnGeom &g, int nw) {
    const long long blocks = (long long)g.n_win * g.heads;
    long long w = blocks / 2048;
    w = w < 1 ? 1 : (w > 16 ? 16 : w);
    return (int)((w + nw - 1) / nw * nw);
}
static int lepe_groups(const AttnGeom &g, int nw) { return (g.n_win + lepe_wpb(g, nw) - 1) / lepe_wpb(g, nw); }

template <typename T, int WS, int CH>
static int launch_lepe_pre(const AttnGeom &g, const T *qkv, const T *out, const T *dout, const float *lepe_w,
                           const float *lepe_b, T *dqkv, float *ws_delta, float *ws_lepe, int *groups, hipStream_t st) {
    using G = LepePre<WS, CH>;
    using E = typename LepeElem<T, CH>::type;
    const int dc = g.d / CH, n_items = dc * WS;
    const int nw = G::NPASS > 1 ? 1 : (n_items <= 64 ? 4 : (n_items <= 128 ? 2 : 1));
    const size_t l2 = (size_t)nw * dc * (2 * G::PP + G::PO) * sizeof(E) + (size_t)(g.d * 20 + G::NPASS * 4 * G::L) * sizeof(float) +
                      4 * sizeof(long long);
    if (l2 > 160 * 1024) return fail(DSW_EUNSUPPORTED, "win_attn_bwd: LePE with w=%d head dim %d needs %zu B of LDS", WS, g.d, l2);
    auto pre = lepe_bwd_pre_kernel<T, WS, CH>;
    if (l2 > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)pre, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
        if (e != hipSuccess) return fail(DSW_ELAUNCH, "win_attn_bwd: LDS attribute: %s", hipGetErrorString(e));
    }
    *groups = lepe_groups(g, nw);
    hipLaunchKernelGGL(pre, dim3(*groups * g.heads), dim3(256), l2, st, g, qkv, out, dout, lepe_w, lepe_b, dqkv, ws_delta, ws_lepe,
                       lepe_wpb(g, nw), nw);
    return check_launch("lepe_bwd_pre_kernel");
}

// d_lepe_w[c][tap], d_lepe_b[c] = sum over window groups (ascending) of the partials
__global__ __launch_bounds__(1024) void lepe_grad_reduce_kernel(const float *__restrict__ ws, int n_win, int C, float *__restrict__ d_w,
                                                                float *__restrict__ d_b) {
    __shared__ float red[32][33];
    int i;
    float acc;
    if (!column_sum_32<32>(ws, n_win, C * 10, red, i, acc)) return;
    const int c = i / 10, k = i - c * 10;
    if (k < 9) d_w[c * 9 + k] = acc;
    else d_b[c] = acc;
}

// DSW_ATTN_BWD_GENERAL=1 in the environment sends every shape to the general kernel (tests compare the two)
static bool force_general_bwd() {
    const char *e = getenv("DSW_ATTN_BWD_GENERAL");
    return e != nullptr && e[0] == '1';
}

template <typename T, int WS, int DP>
static int launch_bwd(const AttnGeom &g, const void *qkv, const void *out, const void *dout, const float *table,
                      float scale, const float *lepe_w, const float *lepe_b, const float *lse, void *dqkv,
                      float *d_table, float *d_lepe_w, float *d_lepe_b, float *ws_table, float *ws_lepe,
                      float *ws_delta, const DropArgs &drop, hipStream_t st) {
    using Cfg = BwdCfg<T, WS, DP>;
    const bool bias = table != nullptr, lepe = lepe_w != nullptr;
    const size_t lds = Cfg::lds_bytes(bias);
    if (lds > 160 * 1024) return fail(DSW_EUNSUPPORTED, "win_attn_bwd: w=%d head dim %d needs %zu B of LDS", WS, g.d, lds);
    const dim3 grid(g.n_win * g.heads), block(Cfg::THREADS);
    int rc, lepe_rows = 0;  // lepe_rows: partial rows of the LePE weight gradient the pre-pass leaves
    if (lepe) {
        // 2-byte elements go through the kernel two channels at a time when every row starts on a 4-byte boundary
        // (with an even head dim every head's offset into a row is even in both layouts of qkv)
        if constexpr (sizeof(T) == 2) {
            if (((g.d | g.ldq | g.ldo | g.ldg) & 1) == 0)
                rc = launch_lepe_pre<T, WS, 2>(g, (const T *)qkv, (const T *)out, (const T *)dout, lepe_w, lepe_b, (T *)dqkv, ws_delta,
                                               ws_lepe, &lepe_rows, st);
            else
                rc = launch_lepe_pre<T, WS, 1>(g, (const T *)qkv, (const T *)out, (const T *)dout, lepe_w, lepe_b, (T *)dqkv, ws_delta,
                                               ws_lepe, &lepe_rows, st);
        } else {
            rc = launch_lepe_pre<T, WS, 1>(g, (const T *)qkv, (const T *)out, (const T *)dout, lepe_w, lepe_b, (T *)dqkv, ws_delta, ws_lepe,
                                           &lepe_rows, st);
        }
        if (rc) return rc;
    }
#define DSW_BWD_LAUNCH2(B_, P_, D_)                                                                              \
    do {                                                                                                         \
        auto kern = win_attn_bwd_kernel<T, WS, DP, B_, P_, D_>;                                                  \
        if (lds > 48 * 1024) {                                                                                   \
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return fail(DSW_ELAUNCH, "win_attn_bwd: LDS attribute: %s", hipGetErrorString(e)); \
        }                                                                                                        \
        hipLaunchKernelGGL(kern, grid, block, lds, st, g, (const T *)qkv, (const T *)out, (const T *)dout, table, \
                           scale, lse, (T *)dqkv, ws_table, lepe ? ws_delta : (const float *)nullptr, drop);    \
    } while (0)
#define DSW_BWD_LAUNCH(B_, P_)                       \
    do {                                             \
        if (drop.thr) DSW_BWD_LAUNCH2(B_, P_, true); \
        else DSW_BWD_LAUNCH2(B_, P_, false);         \
    } while (0)
    constexpr bool BF16 = sizeof(T) == 2;
    const bool pair = BF16 && (g.d % 2 == 0);
    bool launched = false;
    if constexpr (BF16 && WS == 16) {
        if (!lepe && !drop.thr && !force_general_bwd()) {  // the two-phase kernel (attn_bwd_w16.hip) takes even head dims <= DP - 2
            const int r2 = launch_bwd_w16_two_phase(g, qkv, out, dout, table, scale, lse, dqkv, ws_table, st);
            if (r2 > 0) return r2;
            launched = r2 == DSW_OK;
        }
    }
    if (launched) {
    } else if constexpr (BF16) {
        if (bias && pair) DSW_BWD_LAUNCH(true, true);
        else if (bias) DSW_BWD_LAUNCH(true, false);
        else if (pair) DSW_BWD_LAUNCH(false, true);
        else DSW_BWD_LAUNCH(false, false);
    } else {
        if (bias) DSW_BWD_LAUNCH(true, false);
        else DSW_BWD_LAUNCH(false, false);
    }
#undef DSW_BWD_LAUNCH
#undef DSW_BWD_LAUNCH2
    rc = check_launch("win_attn_bwd_kernel");
    if (rc) return rc;
    if (bias) {
        const int T2 = (2 * WS - 1) * (2 * WS - 1);
        hipLaunchKernelGGL(bias_grad_reduce_kernel, dim3(g.heads * ((T2 + 15) / 16)), dim3(1024), 0, st, ws_table,
                           g.n_win, g.heads, T2, d_table);
        rc = check_launch("bias_grad_reduce_kernel");
        if (rc) return rc;
    }
    if (lepe) {
        hipLaunchKernelGGL(lepe_grad_reduce_kernel, dim3((g.C * 10 + 31) / 32), dim3(1024), 0, st, ws_lepe, lepe_rows, g.C,
                           d_lepe_w, d_lepe_b);
        rc = check_launch("lepe_grad_reduce_kernel");
    }
    return rc;
}

int make_geom(const char *who, int n_img, int H, int W, int w, int C, int heads, int layout, int ld_qkv, int ld_out,
              int ld_dout, AttnGeom &g);

}  // namespace dsw

using namespace dsw;

static size_t table_ws_floats(const AttnGeom &g, int w) { return (size_t)g.n_win * g.heads * (2 * w - 1) * (2 * w - 1); }
static size_t lepe_ws_floats(const AttnGeom &g) { return (size_t)g.n_win * g.C * 10; }
static size_t delta_ws_floats(const AttnGeom &g, int w) { return (size_t)g.n_win * g.heads * w * w; }

extern "C" size_t dsw_win_attn_bwd_workspace(int n_img, int H, int W, int w, int C, int heads, int has_bias,
                                             int has_lepe) {
    AttnGeom g;
    if (make_geom("dsw_win_attn_bwd_workspace", n_img, H, W, w, C, heads, DSW_QKV_CHUNKED, 0, 0, 0, g)) return 0;
    size_t fl = 0;
    if (has_bias) fl += table_ws_floats(g, w);
    if (has_lepe) fl += lepe_ws_floats(g) + delta_ws_floats(g, w);
    return fl * sizeof(float);
}

extern "C" int dsw_win_attn_bwd(const void *qkv, const void *out, const void *dout, int dtype, int layout,
                                int n_img, int H, int W, int w, int C, int heads, int ld_qkv, int ld_out, int ld_dout,
                                const float *bias_table, float scale, const float *lepe_w, const float *lepe_b, const float *lse,
                                void *dqkv, float *d_bias_table, float *d_lepe_w, float *d_lepe_b,
                                void *workspace, size_t workspace_bytes, float dropout_p, unsigned long long dropout_seed,
                                const int *n_live, void *stream) {
    AttnGeom g;
    int rc = make_geom("dsw_win_attn_bwd", n_img, H, W, w, C, heads, layout, ld_qkv, ld_out, ld_dout, g);
    if (rc) return rc;
    DSW_REQUIRE(n_live == nullptr || (H == 0 && lepe_w == nullptr), "dsw_win_attn_bwd: n_live goes with packed windows (H = W = 0) and no LePE");
    g.n_live = n_live;
    DropArgs drop;
    rc = make_drop("dsw_win_attn_bwd", dropout_p, dropout_seed, drop);
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
        if (w == 16) return big ? launch_bwd<T, 16, 64>(g, qkv, out, dout, bias_table, scale, lepe_w, lepe_b, lse, dqkv, d_bias_table, d_lepe_w, d_lepe_b, ws_table, ws_lepe, ws_delta, drop, st)
                                : launch_bwd<T, 16, 32>(g, qkv, out, dout, bias_table, scale, lepe_w, lepe_b, lse, dqkv, d_bias_table, d_lepe_w, d_lepe_b, ws_table, ws_lepe, ws_delta, drop, st);
        return big ? launch_bwd<T, 8, 64>(g, qkv, out, dout, bias_table, scale, lepe_w, lepe_b, lse, dqkv, d_bias_table, d_lepe_w, d_lepe_b, ws_table, ws_lepe, ws_delta, drop, st)
                   : launch_bwd<T, 8, 32>(g, qkv, out, dout, bias_table, scale, lepe_w, lepe_b, lse, dqkv, d_bias_table, d_lepe_w, d_lepe_b, ws_table, ws_lepe, ws_delta, drop, st);
    });
    return DSW_OK;
}
