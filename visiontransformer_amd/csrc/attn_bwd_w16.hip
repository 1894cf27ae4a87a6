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

// ======================================================================================================
// The same two phases for head dim 30 (the DSWT shape), with less work for the vector unit around them:
//  * a head's slice of a token row moves as 16-byte slots (the last one read and written as the 16 bytes that END
//    with the slice and shifted in registers, see attn_fwd_w16.hip), 10 loads per thread instead of 40;
//  * a wave reads the K and V fragments of its own key tile from global memory straight into the B operand
//    registers; only Q' and dO' are staged (delta by v_dot2c_f32_bf16 and two quad-wide DPP adds), one barrier;
//  * the row constants sit in columns 30, 31 of the padded head dim whatever the slice holds;
//  * both loops are unrolled: every LDS address is a lane constant plus an immediate;
//  * d(bias table): the diagonal sums of a tile are added to tables in LDS - one per wave and lane half, so that no
//    two lanes of an instruction meet in an entry - by a read, an add and a write, four of each per tile, instead
//    of being finished across lanes (45 vector instructions per tile); ds_add_f32 would do it in one instruction
//    but takes about a hundred LDS cycles;
//  * phase 2 reads its bias values from a mirrored copy of the table, ascending like phase 1's;
//  * dK, dV and dQ leave through the wave's own rows of the images, complete 60-byte pieces of 16 rows per store
//    instruction (store_tile_rows_d30).
struct Bwd30Cfg {
    static constexpr int WS = 16, L = 256, DP = 32, NT = 8, NW = 8, THREADS = 512, D = 30;
    static constexpr int N_BUF = L * DP;
    static constexpr int TBF = bias_lds<16>::FLOATS;
    // per-wave d(table), one for each lane half: [qi - ki_local + 1][column], 32 columns a row with the two halves of
    // 16 swapped in odd rows (the two key rows of a tile go to neighbouring rows: 16 banks apart)
    static constexpr int BG_STRIDE = 32, BG_ROWS = 17, BG_HALF = BG_STRIDE * BG_ROWS, BG_FLOATS = 2 * BG_HALF;
    static constexpr size_t lds_bytes(bool bias) {
        return (size_t)2 * N_BUF * sizeof(bf16_t) + (bias ? (size_t)(2 * TBF + NW * BG_FLOATS) * sizeof(float) : 0);
    }
};

// BWD30_STAMPS (diagnostic builds only, tools/bwd_stamps.py): thread 0 of every workgroup records the shader clock at
// points of the head-dim-30 kernel into a device array that dsw_debug_read_stamps copies out; no output depends on it
#ifdef BWD30_STAMPS
__device__ unsigned long long g_bwd30_stamps[16 * 8192];
#define BWD30_STAMP(i)                                                                                       \
    do {                                                                                                     \
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_bwd30_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define BWD30_STAMP(i) ((void)0)
#endif

template <int CTRL> __device__ __forceinline__ float dpp_add_quad(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_hw, a), __builtin_bit_cast(bf16x2_hw, b), acc, false);
}

template <bool BIAS>
__global__ __launch_bounds__(512, 4) void win_attn_bwd_w16_d30_kernel(
    AttnGeom g, const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ out, const bf16_t *__restrict__ dout,
    const float *__restrict__ table, float scale, const float *__restrict__ lse, bf16_t *__restrict__ dqkv,
    float *__restrict__ ws_table) {
    using Cfg = Bwd30Cfg;
    using T = bf16_t;
    constexpr int WS = 16, L = Cfg::L, NT = Cfg::NT, NW = Cfg::NW, THREADS = Cfg::THREADS, DP = Cfg::DP, D = Cfg::D;
    constexpr int TS = bias_lds<WS>::TS, TBF = Cfg::TBF, TR = bias_lds<WS>::ROWS, TMAX = (2 * WS - 2) * (TS + 1);
    constexpr int BGS = Cfg::BG_STRIDE, BGF = Cfg::BG_FLOATS;
    constexpr int BLK = 32 * DP;  // elements of a 32-row block of an operand image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *bufA = reinterpret_cast<T *>(smem);  // phase 1: Q' (scaled, augmented)   phase 2: K (augmented)
    T *bufB = bufA + Cfg::N_BUF;            // phase 1: dO' (augmented)          phase 2: V (augmented)
    float *tb = reinterpret_cast<float *>(bufB + Cfg::N_BUF);  // BIAS: entry e at e
    float *tbm = tb + TBF;                                     // BIAS: entry e at TMAX - e
    float *bgw = tbm + TBF;                                    // BIAS: [NW][BGF]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int C = g.ldo, CG = g.ldg, C3 = g.ldq;
    const float sc2 = scale * LOG2E;
    BWD30_STAMP(0);
    const int id = xcd_remap(blockIdx.x, g.n_win * g.heads);
    // (wave-uniform quotients of the vector unit's division, handed back to scalar registers)
    const int win = __builtin_amdgcn_readfirstlane(id / g.heads), head = id - win * g.heads;
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
        zero_window_slice<T>(dqkv, C3, origin, pitch, WS, qoff, D, false, 0, tid, THREADS);
        zero_window_slice<T>(dqkv, C3, origin, pitch, WS, koff, D, false, 0, tid, THREADS);
        zero_window_slice<T>(dqkv, C3, origin, pitch, WS, voff, D, head == g.heads - 1, 3 * g.C, tid, THREADS);
        if constexpr (BIAS)
            for (int it = tid; it < TR * TR; it += THREADS) ws_table[((size_t)win * g.heads + head) * (TR * TR) + it] = 0.f;
        return;
    }
    auto rel_row = [&](int l) -> unsigned { return (unsigned)((l >> 4) * pitch + (l & 15)); };
    const T *qkv0 = qkv + origin * C3;  // wave-uniform 64-bit bases, 32-bit lane offsets
    const T *dout0 = dout + origin * CG + head * D, *out0 = out + origin * C + head * D;
    const float *lse0 = lse + ((size_t)win * g.heads + head) * L;

    // ---- loads -----------------------------------------------------------------------------------------------
    // this wave's key tile: K and V fragments (slots hh and 2 + hh of row 32 wave + l31), the B operands of phase 1
    const int krow = 32 * wave + l31;
    const unsigned krel = rel_row(krow);
    uint4 kfb[2], vfb[2];
    {
        const unsigned e0 = krel * C3 + 8 * hh;
        kfb[0] = load16_a4(qkv0 + (e0 + koff));
        vfb[0] = load16_a4(qkv0 + (e0 + voff));
        kfb[1] = load16_a4(qkv0 + (e0 + koff + 16 - 2 * hh));
        vfb[1] = load16_a4(qkv0 + (e0 + voff + 16 - 2 * hh));
    }
    // Q, dO, O and lse of two rows per thread, a slot each
    const int sc = tid & 3;
    const bool slast = sc == 3;
    uint4 qv[2], gv[2], ov[2];
    float lsev[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 2) + 128 * i;
        const unsigned rel = rel_row(row), e0 = 8 * sc - (slast ? 2 : 0);
        qv[i] = load16_a4(qkv0 + (rel * C3 + qoff + e0));
        gv[i] = load16_a4(dout0 + (rel * CG + e0));
        ov[i] = load16_a4(out0 + (rel * C + e0));
        lsev[i] = lse0[row];
    }
    // the head's column of the bias table, both sweeps requested at once behind the operands: one trip to memory
    static_assert(TR * TR <= 2 * THREADS, "two table entries per thread");
    float tabv[2] = {0.f, 0.f};
    if constexpr (BIAS) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (tid + THREADS * i < TR * TR) tabv[i] = table[(size_t)(tid + THREADS * i) * g.heads + head];
    }
    BWD30_STAMP(12);
    if constexpr (BIAS) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int it = tid + THREADS * i, di = it / TR, dj = it - di * TR;
            if (it < TR * TR) {
                const float v = tabv[i] * LOG2E;
                tb[di * TS + dj] = v;
                tbm[TMAX - (di * TS + dj)] = v;
            }
        }
        // this wave's d(table): zero
        float4 *z = reinterpret_cast<float4 *>(bgw + wave * BGF);
        for (int i = lane; i < BGF / 4; i += 64) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    BWD30_STAMP(1);
    if (hh) {  // columns 24-31 arrived as 22-29: shift, ones in the two last columns
        kfb[1] = make_uint4(kfb[1].y, kfb[1].z, kfb[1].w, 0x3F803F80u);
        vfb[1] = make_uint4(vfb[1].y, vfb[1].z, vfb[1].w, 0x3F803F80u);
    }
    // ---- Q' (times scale * log2 e, -lse in columns 30, 31) and dO' (-delta there) into the images -----------------
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 2) + 128 * i;
        if (slast) {
            qv[i] = make_uint4(qv[i].y, qv[i].z, qv[i].w, 0u);
            gv[i] = make_uint4(gv[i].y, gv[i].z, gv[i].w, 0u);
            ov[i] = make_uint4(ov[i].y, ov[i].z, ov[i].w, 0u);
        }
        // delta = rowsum(dO * O): 8 elements here, then the four lanes of the row
        float part = dot2_bf16(ov[i].x, gv[i].x, 0.f);
        part = dot2_bf16(ov[i].y, gv[i].y, part);
        part = dot2_bf16(ov[i].z, gv[i].z, part);
        part = dot2_bf16(ov[i].w, gv[i].w, part);
        part = dpp_add_quad<0xB1>(part);  // quad_perm [1,0,3,2]
        part = dpp_add_quad<0x4E>(part);  // quad_perm [2,3,0,1]
        const uint32_t in[4] = {qv[i].x, qv[i].y, qv[i].z, qv[i].w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack_bf16(__uint_as_float(in[j] << 16) * sc2, __uint_as_float(in[j] & 0xffff0000u) * sc2);
        if (slast) {
            o[3] = bf16_hi_lo(-lsev[i] * LOG2E);
            gv[i].w = bf16_hi_lo(-part);
        }
        const int off = img<DP>(row, 8 * sc);
        *reinterpret_cast<uint4 *>(bufA + off) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4 *>(bufB + off) = gv[i];
    }
    // per-lane LDS offsets (elements) inside a 32-row block of an image
    int a_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) a_off[s] = img<DP>(l31, 16 * s + 8 * hh);
    // transposing reads: lane 4q+p of a 16-lane group passes row q, columns 4p..4p+3; rows +0 / +8 of a 16-row step
    const int trq = (lane >> 2) & 3, trp = lane & 3, trg = (lane >> 4) & 1;
    int t_off[2];
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) t_off[h8] = img<DP>(4 * hh + trq + 8 * h8, 16 * trg + 4 * trp);

    f32x16 dKacc, dVacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        dKacc[r] = 0.f;
        dVacc[r] = 0.f;
    }
    BWD30_STAMP(2);
    lds_barrier();  // Q', dO', the bias tables are in LDS; every wave's d(table) is zero
    BWD30_STAMP(3);

    // =================== phase 1: key on the lane; dK^T, dV^T, d(bias table) ===================
    {
        const int ki = krow >> 4, kj = krow & 15;
        // table entry of (q, key): (qi-ki+15)*TS + (qj-kj+15); q = qb*32 + rho(r) + 4hh
        const int seed_base = (WS - 1) * (TS + 1) - ki * TS - kj + 4 * hh;
        // d(table) entry of this lane's sums: table (wave, hh), row 2qb + G - kip + 1, column 15 - m (sum A: dj = 4hh - m)
        // or 31 - m (sum B: dj = 16 + 4hh - m); odd rows keep their two column halves swapped.  Two lane addresses
        // serve all four: x (sum A of G = 0, sum B of G = 1 a row further on) and x ^ 16 (the other two).
        const int kip = (lane >> 4) & 1;
        const int x0 = wave * BGF + hh * Cfg::BG_HALF + (1 - kip) * BGS + ((15 - (lane & 15)) ^ ((1 - kip) << 4));
        float *px = bgw + x0, *py = bgw + (x0 ^ 16);
#pragma unroll
        for (int qb = 0; qb < NT; ++qb) {
            const T *blkA = bufA + qb * BLK, *blkB = bufB + qb * BLK;
            f32x16 sacc, pacc;
            if constexpr (BIAS) {
                // (the tile's first entry, counted from the start of LDS, goes through an empty asm: one address
                // register per tile and small immediates, instead of one register per read, all computed up front)
                int tile_base = (int)(tb - reinterpret_cast<float *>(smem)) + seed_base + 2 * qb * TS;
                asm volatile("" : "+v"(tile_base));
                const float *tp = reinterpret_cast<const float *>(smem) + tile_base;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = tp[(rho(r) >> 4) * TS + (rho(r) & 15)];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
            // S' = Q' K^T (base 2, lse and bias included), dP' = dO' V^T (delta included)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
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
                px[2 * qb * BGS] += sa[0];
                py[2 * qb * BGS] += sb[0];
                py[(2 * qb + 1) * BGS] += sa[1];
                px[(2 * qb + 1) * BGS] += sb[1];
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
                const uint2 g0 = lds_read_tr16(blkB + 16 * s * DP + t_off[0]), g1 = lds_read_tr16(blkB + 16 * s * DP + t_off[1]);
                const uint2 q0 = lds_read_tr16(blkA + 16 * s * DP + t_off[0]), q1 = lds_read_tr16(blkA + 16 * s * DP + t_off[1]);
                dVacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(g0.x, g0.y, g1.x, g1.y)), as_bf16x8(pb), dVacc, 0, 0, 0);
                dKacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(q0.x, q0.y, q1.x, q1.y)), as_bf16x8(db), dKacc, 0, 0, 0);
            }
        }
    }
    BWD30_STAMP(4);
    // phase 2's B operands: this wave's own query block, rows of Q' and dO' (column = query on the lane)
    // (everything from here on that depends on the lane is computed afresh from a copy of the lane index the
    // compiler cannot look through: kept alive across phase 1 instead, these values do not fit the 128 registers and
    // the spills would make the kernel one that uses scratch memory, which costs far more than recomputing them)
    int lane2 = lane;
    asm volatile("" : "+v"(lane2));
    const int l31b = lane2 & 31, hhb = lane2 >> 5, tid2 = wave * 64 + lane2, qrow2 = 32 * wave + l31b;
    int a_off2[2], t_off2[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) a_off2[s] = img<DP>(l31b, 16 * s + 8 * hhb);
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) t_off2[h8] = img<DP>(4 * hhb + ((lane2 >> 2) & 3) + 8 * h8, 16 * ((lane2 >> 4) & 1) + 4 * (lane2 & 3));
    uint4 qfb[2], gfb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        qfb[s] = *reinterpret_cast<const uint4 *>(bufA + wave * BLK + a_off2[s]);
        gfb[s] = *reinterpret_cast<const uint4 *>(bufB + wave * BLK + a_off2[s]);
    }
    lds_barrier();  // nobody reads Q' / dO' any more; every wave's d(table) is complete

    // results leave through this wave's own 32 rows of the images (store_tile_rows_d30): nobody else touches them
    // between the barriers
    bf16_t *dq0 = dqkv + origin * C3;
    store_tile_rows_d30(bufA + wave * BLK, dKacc, LN2, dq0 + koff, C3, pitch, wave, lane2);  // Q' carries scale * log2(e): dK = Q'^T dS * ln 2
    store_tile_rows_d30(bufB + wave * BLK, dVacc, 1.f, dq0 + voff, C3, pitch, wave, lane2);
    BWD30_STAMP(8);
    // K and V into LDS for phase 2, straight from the B-operand registers (16 bytes = one slot per register)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        *reinterpret_cast<uint4 *>(bufA + wave * BLK + a_off2[s]) = kfb[s];
        *reinterpret_cast<uint4 *>(bufB + wave * BLK + a_off2[s]) = vfb[s];
    }
    BWD30_STAMP(9);
    if constexpr (BIAS) {
        // the table entered the logits times log2(e) inside exp2, so d/dtable is dS itself.  Entry (di, dj):
        // wave wv holds it in row di + 2*wv + 1 of its table when that row exists.
        // A wave takes whole rows di of the table (wave, wave + 8, ...): which waves hold the row is then a scalar
        // loop range.  The lower lane half sums the tables of the lower lane halves (dj index = column), the upper
        // half those of the upper halves, whose columns are 4 further on in dj; one exchange adds the two.
        float *dst = ws_table + ((size_t)win * g.heads + head) * (TR * TR);
        const int dj = l31b, djh = dj - 4 * hhb;
        const bool have = djh >= 0;  // (column 31 of the lower half: read, never stored)
        for (int dix = __builtin_amdgcn_readfirstlane(wave); dix < TR; dix += NW) {
            const int di = dix - (WS - 1);
            // rows 0 <= di + 2 wv + 1 <= 16 of wave wv's tables
            const int lo = di < -1 ? (-di) >> 1 : 0, hi = min(NW - 1, (WS - 1 - di) >> 1);
            float acc = 0.f;
            for (int wv = lo; wv <= hi; ++wv) {
                const int row = di + 2 * wv + 1, sw = (row & 1) << 4;
                const float v = bgw[wv * BGF + hhb * Cfg::BG_HALF + row * BGS + ((have ? djh : 0) ^ sw)];
                acc += have ? v : 0.f;
            }
            const unsigned ai = __float_as_uint(acc);
            const auto sw2 = __builtin_amdgcn_permlane32_swap(ai, ai, false, false);
            acc = __uint_as_float(sw2[0]) + __uint_as_float(sw2[1]);
            if (lane2 < TR) dst[dix * TR + dj] = acc;
        }
    }
    BWD30_STAMP(10);
    if (head == g.heads - 1 && C3 > 3 * g.C) {  // the padding columns of this window's dqkv rows (an even count)
        const int np2 = (C3 - 3 * g.C) >> 1;    // dwords per row; two threads share a row
        T *prow = dqkv + (origin + rel_row(tid2 >> 1)) * C3 + 3 * g.C;
        for (int j = tid2 & 1; j < np2; j += 2) *reinterpret_cast<uint32_t *>(prow + 2 * j) = 0u;
    }
    BWD30_STAMP(11);
    lds_barrier();  // K, V are in LDS
    BWD30_STAMP(5);

    // =================== phase 2: query on the lane; dQ^T ===================
    f32x16 dQacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dQacc[r] = 0.f;
    {
        // key of register r: kt*32 + rho(r) + 4hh  ->  entry e = (qi-ki+15)*TS + (qj-kj+15), read at TMAX - e
        const int seed_base = TMAX - (((qrow2 >> 4) + WS - 1) * TS + (qrow2 & 15) + (WS - 1) - 4 * hhb);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const T *blkA = bufA + kt * BLK, *blkB = bufB + kt * BLK;
            f32x16 sacc, pacc;
            if constexpr (BIAS) {
                int tile_base = (int)(tbm - reinterpret_cast<float *>(smem)) + seed_base + 2 * kt * TS;
                asm volatile("" : "+v"(tile_base));
                const float *tp = reinterpret_cast<const float *>(smem) + tile_base;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = tp[(rho(r) >> 4) * TS + (rho(r) & 15)];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uint4 ka = *reinterpret_cast<const uint4 *>(blkA + a_off2[s]);
                const uint4 va = *reinterpret_cast<const uint4 *>(blkB + a_off2[s]);
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
                const uint2 k0 = lds_read_tr16(blkA + 16 * s * DP + t_off2[0]), k1 = lds_read_tr16(blkA + 16 * s * DP + t_off2[1]);
                dQacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(k0.x, k0.y, k1.x, k1.y)), as_bf16x8(db), dQacc, 0, 0, 0);
            }
        }
    }
    BWD30_STAMP(6);
    lds_barrier();  // nobody reads K / V any more
    store_tile_rows_d30(bufA + wave * BLK, dQacc, scale, dq0 + qoff, C3, pitch, wave, lane2);
    BWD30_STAMP(7);
}

// DSW_ATTN_BWD_W16_OLD=1 in the environment keeps head dim 30 on the kernel for any even head dim (tests compare the two)
static bool force_old_two_phase() {
    const char *e = getenv("DSW_ATTN_BWD_W16_OLD");
    return e != nullptr && e[0] == '1';
}

int launch_bwd_w16_two_phase(const AttnGeom &g, const void *qkv, const void *out, const void *dout, const float *table,
                             float scale, const float *lse, void *dqkv, float *ws_table, hipStream_t st) {
    if (g.d == 30 && ((g.ldq | g.ldo | g.ldg) & 1) == 0 && !force_old_two_phase()) {
        const bool bias = table != nullptr;
        const dim3 grid(g.n_win * g.heads), block(512);
        const size_t lds = Bwd30Cfg::lds_bytes(bias);
        if (bias) {
            auto kern = win_attn_bwd_w16_d30_kernel<true>;
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(DSW_ELAUNCH, "win_attn_bwd: LDS attribute: %s", hipGetErrorString(e));
            hipLaunchKernelGGL(kern, grid, block, lds, st, g, (const bf16_t *)qkv, (const bf16_t *)out, (const bf16_t *)dout, table, scale, lse,
                               (bf16_t *)dqkv, ws_table);
        } else {
            hipLaunchKernelGGL(win_attn_bwd_w16_d30_kernel<false>, grid, block, lds, st, g, (const bf16_t *)qkv, (const bf16_t *)out,
                               (const bf16_t *)dout, table, scale, lse, (bf16_t *)dqkv, ws_table);
        }
        return check_launch("win_attn_bwd_w16_d30_kernel");
    }
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

#ifdef BWD30_STAMPS
extern "C" int dsw_debug_read_stamps(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dsw::g_bwd30_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif
