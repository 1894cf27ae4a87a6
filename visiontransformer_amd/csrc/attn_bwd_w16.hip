// Fused window attention, backward, for 16 x 16 windows in bf16: dq, dk, dv and d(bias table) for one
// (window, head) per workgroup with NO exchange of dS between the waves and no barrier inside the loops.
// Reference ops differentiated: DSwinT_arch.py:122-125, :171-172 (and msft_arch.py:136 without LePE).
//
// The two orientations of the score tile each serve one half of the gradients without a transpose:
//   phase 1, KEY on the MFMA lane (wave = one 32-key tile, loop over the 32-query blocks):
//     S = Q K^T and dP = dO V^T come out as (query = accumulator row, key = lane), which is the B operand
//     of dV^T += dO^T P and dK^T += Q^T dS; d(bias table) is summed along the table's diagonals in registers;
//   phase 2, QUERY on the lane (wave = one 32-query block, loop over the 32-key tiles):
//     S^T = K Q^T and dP^T = V dO^T come out as (key = row, query = lane), the B operand of dQ^T += K^T dS^T.
// S, dP and the exponentials are computed twice (the matrix cores are nowhere near busy in this kernel);
// in exchange nothing but the operands themselves lives in LDS: Q and dO (row-major, 16-byte slots XOR-
// swizzled so that both the row reads and the transposing reads are bank-conflict-free) during phase 1,
// then K and V - written straight from the B-operand registers the waves already hold - during phase 2.
// Four workgroup barriers in all.  The row constants ride in the padding of the head dimension:
// columns d, d+1 of Q hold -lse*log2(e) split into two bf16 (hi + lo) against ones in K, and those of dO hold
// -delta against ones in V, so the accumulators start from the bias alone and need no per-row arithmetic
// (which is why the head dim has to leave two padding columns: d <= DP - 2).
#include <utility>

#include "attn_common.hpp"

#pragma clang fp contract(fast)

namespace dsw {

template <int DP> struct Bwd2Cfg {
    static constexpr int WS = 16, L = 256, NT = 8, NW = 8, THREADS = 512;
    static constexpr int N_BUF = L * DP;  // elements of one operand image
    static constexpr int TBF = bias_lds<16>::FLOATS, BGW_FLOATS = BGW_ROWS * BGW_STRIDE;
    static constexpr size_t lds_bytes(bool bias) {
        return (size_t)2 * N_BUF * sizeof(bf16_t) + (size_t)L * sizeof(float) + (bias ? (size_t)(TBF + NW * BGW_FLOATS) * sizeof(float) : 0);
    }
};

template <int CTRL> __device__ __forceinline__ float dpp_zero_fill(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}

// ---- d(bias table): diagonal sums of a dS tile (query = register, key = lane) -------------------------
// Register R holds query column qj = QJ + 4 hh with QJ = (R & 3) + 8 ((R >> 2) & 1); its 16-lane rows hold
// kj = 0..15.  Shifting a row left by QJ lines up equal dj = qj - kj for the elements with kj >= QJ
// (sum A, lane m holds kj = m + QJ); the others, shifted right by 16 - QJ, form sum B (kj = m + QJ - 16).
// The lane halves differ by 4 in qj, which is settled once per block after the sums (finish_diag).
template <int R> __device__ __forceinline__ void diag_step2(const f32x16 &ds, float (&sa)[2], float (&sb)[2]) {
    constexpr int QJ = (R & 3) + 8 * ((R >> 2) & 1);
    if constexpr (QJ == 0) {
        sa[R >> 3] += ds[R];
    } else {
        sa[R >> 3] += dpp_zero_fill<0x100 + QJ>(ds[R]);          // row_shl:QJ
        sb[R >> 3] += dpp_zero_fill<0x110 + (16 - QJ)>(ds[R]);   // row_shr:16-QJ
    }
}
template <int... R>
__device__ __forceinline__ void diag_sums2(std::integer_sequence<int, R...>, const f32x16 &ds, float (&sa)[2], float (&sb)[2]) {
    (diag_step2<R>(ds, sa, sb), ...);
}

template <int DP, bool BIAS>
__global__ __launch_bounds__(512, (DP == 32 ? 4 : 2)) void win_attn_bwd_w16_kernel(
    AttnGeom g, const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ out, const bf16_t *__restrict__ dout,
    const float *__restrict__ table, float scale, const float *__restrict__ lse, bf16_t *__restrict__ dqkv,
    float *__restrict__ ws_table) {
    using Cfg = Bwd2Cfg<DP>;
    using T = bf16_t;
    constexpr int WS = 16, L = Cfg::L, NT = Cfg::NT, NW = Cfg::NW, THREADS = Cfg::THREADS;
    constexpr int TS = bias_lds<WS>::TS, TBF = Cfg::TBF, TR = bias_lds<WS>::ROWS, BGW_FLOATS = Cfg::BGW_FLOATS;
    constexpr int NS = DP / 16;  // 16-wide k steps of the head dimension
    constexpr int NU = DP / 32;  // 32-row blocks of the head dimension
    constexpr int BLK = 32 * DP; // elements of a 32-row block of an operand image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *bufA = reinterpret_cast<T *>(smem);  // phase 1: Q' (scaled, augmented)   phase 2: K (augmented)
    T *bufB = bufA + Cfg::N_BUF;            // phase 1: dO' (augmented)          phase 2: V (augmented)
    float *nls = reinterpret_cast<float *>(bufB + Cfg::N_BUF);  // [L] -lse * log2(e)
    float *tb = nls + L;                                        // BIAS: [TBF]
    float *bgw = tb + TBF;                                      // BIAS: [NW][BGW_FLOATS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int d = g.d, C = g.ldo, CG = g.ldg, C3 = g.ldq;
    const int total = g.n_win * g.heads;
    const float sc2 = scale * LOG2E;
    // x as two bf16 whose sum carries 16 mantissa bits
    auto hi_lo = [](float x) -> uint32_t {
        const bf16_t h = f32_to_bf16(x);
        const bf16_t l = f32_to_bf16(x - bf16_to_f32(h));
        return (uint32_t)h | ((uint32_t)l << 16);
    };
    // where a work item (window, head) lives: token row of window position 0, row pitch, column offsets
    struct Where {
        int win, head, qoff, koff, voff, pitch;
        size_t origin;
    };
    auto locate = [&](int item) -> Where {
        Where w;
        // (integer division runs on the vector unit: its wave-uniform quotients are handed back to the scalar
        // registers, so that everything derived from them - offsets, base addresses - stays out of the VGPRs)
        const int id = xcd_remap(item, total);
        w.win = __builtin_amdgcn_readfirstlane(id / g.heads);
        w.head = id - w.win * g.heads;
        qkv_offsets(g, w.head, w.qoff, w.koff, w.voff);
        if (g.H == 0) {
            w.origin = (size_t)w.win * L;
            w.pitch = WS;
        } else {
            const int per = g.nWh * g.nWw;
            const int b = __builtin_amdgcn_readfirstlane(w.win / per), r = w.win - b * per;
            const int wr = __builtin_amdgcn_readfirstlane(r / g.nWw), wc = r - wr * g.nWw;
            w.origin = ((size_t)b * g.H + wr * WS) * g.W + wc * WS;
            w.pitch = g.W;
        }
        return w;
    };
    // token row of window position l relative to the window's first token (fits 32 bits together with a column)
    auto rel_row = [](const Where &w, int l) -> unsigned { return (unsigned)((l / WS) * w.pitch + (l % WS)); };

    // ---- staging geometry ------------------------------------------------------------------------------
    // A head's slice of a token row is d contiguous elements of a much longer row: DP/2 lanes take one dword
    // each of one row (a wave instruction then covers 64 / (DP/2) rows with one contiguous run each), for all
    // five operands, issued before anything is consumed.
    constexpr int DW = DP / 2, RS = THREADS / DW, ITER = L / RS;  // lanes per row, rows per sweep, sweeps
    const int r0 = tid / DW, e = 2 * (tid % DW);
    const bool eok = e < d, eaug = e == d;
    const int ec = eok ? e : 0;  // lanes past the head dim re-read element 0 (a valid address) and drop it
    uint32_t qv[ITER], gv[ITER], ov[ITER], kv[ITER], vv[ITER];
    float lsev = 0.f;
    auto load_item = [&](const Where &w) {
        // wave-uniform 64-bit bases, 32-bit lane offsets (rows of one window span at most 16 image rows)
        const T *qkv0 = qkv + w.origin * C3, *dout0 = dout + w.origin * CG + (size_t)w.head * d, *out0 = out + w.origin * C + (size_t)w.head * d;
        static_assert(RS % WS == 0, "a sweep is a whole number of window rows");
        const unsigned rel0 = rel_row(w, r0), step = (unsigned)(RS / WS) * w.pitch;
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const unsigned rel = rel0 + i * step;
            kv[i] = *reinterpret_cast<const uint32_t *>(qkv0 + (rel * C3 + w.koff + ec));
            vv[i] = *reinterpret_cast<const uint32_t *>(qkv0 + (rel * C3 + w.voff + ec));
            qv[i] = *reinterpret_cast<const uint32_t *>(qkv0 + (rel * C3 + w.qoff + ec));
            gv[i] = *reinterpret_cast<const uint32_t *>(dout0 + (rel * CG + ec));
            ov[i] = *reinterpret_cast<const uint32_t *>(out0 + (rel * C + ec));
        }
        if (tid < L) lsev = lse[((size_t)w.win * g.heads + w.head) * L + tid];
    };

    // per-lane LDS offsets that do not change from block to block (the slot pattern repeats every 16 rows)
    const int key = 32 * wave + l31;
    int a_off[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) a_off[s] = img<DP>(l31, 16 * s + 8 * hh);
    // transposing reads: lane 4q+p of a 16-lane group passes row q, columns 4p..4p+3
    const int trq = (lane >> 2) & 3, trp = lane & 3, trg = (lane >> 4) & 1;
    // [16-row step][32-column block][rows +0 / +8] of a 32-row block
    auto t_off = [&](int s, int u, int h8) -> int { return img<DP>(16 * s + 4 * hh + trq + 8 * h8, 32 * u + 16 * trg + 4 * trp); };
    const int ki = key / WS, kj = key - ki * WS;
    const bool upw = hh == 1 && (lane & 15) >= 12;  // upper lane half: class-A sums that turn out to be dj > 0
    constexpr int DWR = DP / 2, RPI = 64 / DWR;     // result rows: dwords per row, rows per store instruction
    const int dw = lane % DWR, rsub = lane / DWR;

    const Where cur = locate(blockIdx.x);
    if (window_is_off(g, cur.win)) {  // uniform over the workgroup: zero gradients, nothing read
        zero_window_slice<T>(dqkv, C3, cur.origin, cur.pitch, WS, cur.qoff, d, false, 0, tid, THREADS);
        zero_window_slice<T>(dqkv, C3, cur.origin, cur.pitch, WS, cur.koff, d, false, 0, tid, THREADS);
        zero_window_slice<T>(dqkv, C3, cur.origin, cur.pitch, WS, cur.voff, d, cur.head == g.heads - 1, 3 * g.C, tid, THREADS);
        if constexpr (BIAS)
            for (int it = tid; it < TR * TR; it += THREADS) ws_table[((size_t)cur.win * g.heads + cur.head) * (TR * TR) + it] = 0.f;
        return;
    }
    load_item(cur);
    {
        // ---- K and V pass through the operand images, only to be picked up as this wave's B fragments ---
#pragma unroll
        for (int i = 0; i < ITER; ++i) {  // ones in the two augmented columns
            const int off = img<DP>(i * RS + r0, e);
            *reinterpret_cast<uint32_t *>(bufA + off) = eok ? kv[i] : (eaug ? 0x3F803F80u : 0u);
            *reinterpret_cast<uint32_t *>(bufB + off) = eok ? vv[i] : (eaug ? 0x3F803F80u : 0u);
        }
        if (tid < L) nls[tid] = -lsev * LOG2E;
        if constexpr (BIAS) {
            for (int it = tid; it < TR * TR; it += THREADS) {
                const int di = it / TR, dj = it - di * TR;
                tb[di * TS + dj] = table[(size_t)it * g.heads + cur.head] * LOG2E;
            }
        }
        lds_barrier();
        uint4 kfb[NS], vfb[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            kfb[s] = *reinterpret_cast<const uint4 *>(bufA + wave * BLK + a_off[s]);
            vfb[s] = *reinterpret_cast<const uint4 *>(bufB + wave * BLK + a_off[s]);
        }
        lds_barrier();
        // ---- Q' (times scale * log2 e, -lse in the padding) and dO' (-delta in the padding) ------------
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            // delta = rowsum(dO * O) over the DP/2 lanes of the row (a rotation all-reduce inside 16-lane rows)
            float part = eok ? __uint_as_float(ov[i] << 16) * __uint_as_float(gv[i] << 16) +
                                   __uint_as_float(ov[i] & 0xffff0000u) * __uint_as_float(gv[i] & 0xffff0000u)
                             : 0.f;
            part += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(part), 0x128, 0xF, 0xF, false));  // row_ror:8
            part += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(part), 0x124, 0xF, 0xF, false));  // row_ror:4
            part += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(part), 0x122, 0xF, 0xF, false));  // row_ror:2
            part += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(part), 0x121, 0xF, 0xF, false));  // row_ror:1
            if constexpr (DW == 32) part += __shfl_xor(part, 16, 64);
            const uint32_t q2 = pack_bf16(__uint_as_float(qv[i] << 16) * sc2, __uint_as_float(qv[i] & 0xffff0000u) * sc2);
            const int row = i * RS + r0, off = img<DP>(row, e);
            *reinterpret_cast<uint32_t *>(bufA + off) = eok ? q2 : (eaug ? hi_lo(nls[row]) : 0u);
            *reinterpret_cast<uint32_t *>(bufB + off) = eok ? gv[i] : (eaug ? hi_lo(-part) : 0u);
        }
        f32x16 dKacc[NU], dVacc[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dKacc[u][r] = 0.f;
                dVacc[u][r] = 0.f;
            }
        float bg_carry_a = 0.f, bg_carry_w = 0.f;  // (G = 1, kip = 0) sums of the previous block
        lds_barrier();  // Q', dO', the bias table are in LDS

        // =================== phase 1: key on the lane; dK^T, dV^T, d(bias table) ===================
#pragma unroll 1
        for (int qb = 0; qb < NT; ++qb) {
            const T *blkA = bufA + qb * BLK, *blkB = bufB + qb * BLK;
            f32x16 sacc, pacc;
            if constexpr (BIAS) {
                // table entry of (q, key): (qi-ki+15)*TS + (qj-kj+15); q = qb*32 + rho(r) + 4hh
                const int lane_base = (WS - 1) * (TS + 1) - ki * TS - kj + 4 * hh + (qb * 2) * TS;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = tb[lane_base + (rho(r) / WS) * TS + (rho(r) % WS)];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
            // S' = Q' K^T (base 2, lse and bias included), dP' = dO' V^T (delta included)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint4 qa = *reinterpret_cast<const uint4 *>(blkA + a_off[s]);
                const uint4 ga = *reinterpret_cast<const uint4 *>(blkB + a_off[s]);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(qa), as_bf16x8(kfb[s]), sacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ga), as_bf16x8(vfb[s]), pacc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(sacc[r]);
                sacc[r] = p;
                pacc[r] = p * pacc[r];
            }
            if constexpr (BIAS) {
                float sa[2] = {0.f, 0.f}, sb[2] = {0.f, 0.f};
                diag_sums2(std::make_integer_sequence<int, 16>{}, pacc, sa, sb);
                float nonw[2], wr[2];
#pragma unroll
                for (int G = 0; G < 2; ++G) {
                    // upper lane half: 4 more lanes to the left (its qj is 4 larger); the class-A elements that
                    // leave the row on the way (lanes 0-3, now 12-15) have dj > 0 like all of class B
                    const float a = upper_rows_rot4(sa[G]), b = upper_rows_rot4(sb[G]);
                    float n = upw ? 0.f : a, w = b + (upw ? a : 0.f);
                    n += upper_half_to_lower(n);
                    w += upper_half_to_lower(w);
                    nonw[G] = n;
                    wr[G] = w;
                }
                // Lanes 0-15 now hold key row kip = 0, lanes 16-31 kip = 1; lane m is dj = -m (nonw) or 16 - m (wr).
                // Row-group G of this block belongs to the wave's table row 2qb + G - kip + 1: the odd row 2qb+1
                // gets (G=0, kip=0) + (G=1, kip=1) and is complete; the even row 2qb gets (G=0, kip=1) plus
                // (G=1, kip=0) of the previous block, carried in a register.  Every entry is stored exactly once.
                const float a0k1 = rows_odd_to_even(nonw[0]), a1k1 = rows_odd_to_even(nonw[1]);
                const float w0k1 = rows_odd_to_even(wr[0]), w1k1 = rows_odd_to_even(wr[1]);
                if (lane < 16) {
                    float *ent = bgw + wave * BGW_FLOATS + (2 * qb) * BGW_STRIDE + 15 - lane;
                    ent[0] = a0k1 + bg_carry_a;              // row 2qb
                    ent[16] = w0k1 + bg_carry_w;
                    ent[BGW_STRIDE] = nonw[0] + a1k1;        // row 2qb + 1
                    ent[BGW_STRIDE + 16] = wr[0] + w1k1;
                }
                bg_carry_a = nonw[1];
                bg_carry_w = wr[1];
            }
            // dV^T += dO^T P, dK^T += Q^T dS (sum over the accumulator rows = queries).  Registers 8s..8s+7 of
            // a lane are queries 16s + 8(j>>2) + 4hh + (j&3): the A operands take the same order from two
            // transposing reads of the row-major images.
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uint4 pb = make_uint4(pack_bf16(sacc[8 * s + 0], sacc[8 * s + 1]), pack_bf16(sacc[8 * s + 2], sacc[8 * s + 3]),
                                            pack_bf16(sacc[8 * s + 4], sacc[8 * s + 5]), pack_bf16(sacc[8 * s + 6], sacc[8 * s + 7]));
                const uint4 db = make_uint4(pack_bf16(pacc[8 * s + 0], pacc[8 * s + 1]), pack_bf16(pacc[8 * s + 2], pacc[8 * s + 3]),
                                            pack_bf16(pacc[8 * s + 4], pacc[8 * s + 5]), pack_bf16(pacc[8 * s + 6], pacc[8 * s + 7]));
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const uint2 g0 = lds_read_tr16(blkB + t_off(s, u, 0)), g1 = lds_read_tr16(blkB + t_off(s, u, 1));
                    const uint2 q0 = lds_read_tr16(blkA + t_off(s, u, 0)), q1 = lds_read_tr16(blkA + t_off(s, u, 1));
                    dVacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(g0.x, g0.y, g1.x, g1.y)), as_bf16x8(pb), dVacc[u], 0, 0, 0);
                    dKacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(q0.x, q0.y, q1.x, q1.y)), as_bf16x8(db), dKacc[u], 0, 0, 0);
                }
            }
        }
        if constexpr (BIAS) {
            if (lane < 16) {  // the last even row of the wave's table
                float *ent = bgw + wave * BGW_FLOATS + (2 * NT) * BGW_STRIDE + 15 - lane;
                ent[0] = bg_carry_a;
                ent[16] = bg_carry_w;
            }
        }
        // phase 2's B operands: this wave's own query block, rows of Q' and dO' (column = query on the lane)
        uint4 qfb[NS], gfb[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            qfb[s] = *reinterpret_cast<const uint4 *>(bufA + wave * BLK + a_off[s]);
            gfb[s] = *reinterpret_cast<const uint4 *>(bufB + wave * BLK + a_off[s]);
        }
        lds_barrier();  // nobody reads Q' / dO' any more; every wave's bias-gradient table is complete

        // Results leave through the wave's own 32 rows of bufA (nobody else touches them until the next
        // barrier), turned into [token][head dim] rows there so that the global stores are runs of whole rows.
        auto store_rows = [&](const f32x16(&acc)[NU], float mul, int col0) {
            T *own = bufA + wave * BLK;
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    *reinterpret_cast<uint2 *>(own + img<DP>(l31, 32 * u + 8 * g4 + 4 * hh)) =
                        make_uint2(pack_bf16(acc[u][4 * g4] * mul, acc[u][4 * g4 + 1] * mul),
                                   pack_bf16(acc[u][4 * g4 + 2] * mul, acc[u][4 * g4 + 3] * mul));
#pragma unroll
            for (int i = 0; i < 32 / RPI; ++i) {
                const int row = i * RPI + rsub;
                const uint32_t v = *reinterpret_cast<const uint32_t *>(own + img<DP>(row, 2 * dw));
                if (2 * dw < d)
                    *reinterpret_cast<uint32_t *>(dqkv + cur.origin * C3 + (rel_row(cur, 32 * wave + row) * C3 + col0 + 2 * dw)) = v;
            }
        };
        store_rows(dKacc, LN2, cur.koff);  // Q' carries scale * log2(e): dK = Q'^T dS * ln 2
        store_rows(dVacc, 1.f, cur.voff);
        // K and V into LDS for phase 2, straight from the B-operand registers (16 bytes = one slot per register)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            *reinterpret_cast<uint4 *>(bufA + wave * BLK + a_off[s]) = kfb[s];
            *reinterpret_cast<uint4 *>(bufB + wave * BLK + a_off[s]) = vfb[s];
        }
        if constexpr (BIAS) {
            // the table entered the logits times log2(e) inside exp2, so d/dtable is dS itself.  Entry (di, dj):
            // wave wv holds it in row di + 2*wv + 1 of its table when that row exists.
            float *dst = ws_table + ((size_t)cur.win * g.heads + cur.head) * (TR * TR);
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
        }
        if (cur.head == g.heads - 1 && C3 > 3 * g.C) {  // the padding columns of this window's dqkv rows (an even count)
            const int npad2 = (C3 - 3 * g.C) >> 1;      // dwords per row; the lanes of a staging row share them
            T *pad0 = dqkv + cur.origin * C3 + 3 * g.C;
            const unsigned rel0 = rel_row(cur, r0), step = (unsigned)(RS / WS) * cur.pitch;
#pragma unroll 1
            for (int i = 0; i < ITER; ++i)
                for (int j = tid % DW; j < npad2; j += DW) *reinterpret_cast<uint32_t *>(pad0 + ((rel0 + i * step) * C3 + 2 * j)) = 0u;
        }
        lds_barrier();  // K, V are in LDS

        // =================== phase 2: query on the lane; dQ^T ===================
        f32x16 dQacc[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) dQacc[u][r] = 0.f;
        const int qq = 32 * wave + l31, qi = qq / WS, qj = qq - qi * WS;
#pragma unroll 1
        for (int kt = 0; kt < NT; ++kt) {
            const T *blkA = bufA + kt * BLK, *blkB = bufB + kt * BLK;
            f32x16 sacc, pacc;
            if constexpr (BIAS) {
                // key of register r: kt*32 + rho(r) + 4hh  ->  entry (qi-ki+15)*TS + (qj-kj+15)
                const int lane_base = (qi + WS - 1 - 2 * kt) * TS + qj + (WS - 1) - 4 * hh;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = tb[lane_base - (rho(r) / WS) * TS - (rho(r) % WS)];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint4 ka = *reinterpret_cast<const uint4 *>(blkA + a_off[s]);
                const uint4 va = *reinterpret_cast<const uint4 *>(blkB + a_off[s]);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ka), as_bf16x8(qfb[s]), sacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(va), as_bf16x8(gfb[s]), pacc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[r] = __builtin_amdgcn_exp2f(sacc[r]) * pacc[r];
            // dQ^T += K^T dS^T (sum over the accumulator rows = keys)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uint4 db = make_uint4(pack_bf16(pacc[8 * s + 0], pacc[8 * s + 1]), pack_bf16(pacc[8 * s + 2], pacc[8 * s + 3]),
                                            pack_bf16(pacc[8 * s + 4], pacc[8 * s + 5]), pack_bf16(pacc[8 * s + 6], pacc[8 * s + 7]));
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const uint2 k0 = lds_read_tr16(blkA + t_off(s, u, 0)), k1 = lds_read_tr16(blkA + t_off(s, u, 1));
                    dQacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(k0.x, k0.y, k1.x, k1.y)), as_bf16x8(db), dQacc[u], 0, 0, 0);
                }
            }
        }
        lds_barrier();  // nobody reads K / V any more
        store_rows(dQacc, scale, cur.qoff);
    }
}

int launch_bwd_w16_two_phase(const AttnGeom &g, const void *qkv, const void *out, const void *dout, const float *table,
                             float scale, const float *lse, void *dqkv, float *ws_table, hipStream_t st) {
    const int DPv = g.d > 32 ? 64 : 32;
    // pairs of bf16 per lane (even head dim and row strides); two padding columns for the row constants
    if ((g.d & 1) || ((g.ldq | g.ldo | g.ldg) & 1) || g.d > DPv - 2 || g.d < 2) return -1;
    const bool bias = table != nullptr;
    const dim3 grid(g.n_win * g.heads), block(512);
#define DSW_BWD2_LAUNCH(DP_, B_)                                                                                   \
    do {                                                                                                           \
        auto kern = win_attn_bwd_w16_kernel<DP_, B_>;                                                              \
        const size_t lds = Bwd2Cfg<DP_>::lds_bytes(B_);                                                            \
        if (lds > 48 * 1024) {                                                                                     \
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return fail(DSW_ELAUNCH, "win_attn_bwd: LDS attribute: %s", hipGetErrorString(e)); \
        }                                                                                                          \
        hipLaunchKernelGGL(kern, grid, block, lds, st, g, (const bf16_t *)qkv, (const bf16_t *)out, (const bf16_t *)dout, \
                           table, scale, lse, (bf16_t *)dqkv, ws_table);                                      \
    } while (0)
    if (DPv == 32) {
        if (bias) DSW_BWD2_LAUNCH(32, true);
        else DSW_BWD2_LAUNCH(32, false);
    } else {
        if (bias) DSW_BWD2_LAUNCH(64, true);
        else DSW_BWD2_LAUNCH(64, false);
    }
#undef DSW_BWD2_LAUNCH
    return check_launch("win_attn_bwd_w16_kernel");
}

}  // namespace dsw
