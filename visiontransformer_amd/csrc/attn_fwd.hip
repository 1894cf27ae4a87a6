// Fused window attention, forward: softmax(q k^T * scale + bias) v (+ LePE) for one
// (window, head) per workgroup, on the gfx950 matrix cores.
// Reference: DSwinT_arch.py:122-125, :135-172, :303-308, :333-337; msft_arch.py:94-139.
//
// Structure.  The window's Q, K and V head slices are staged once in LDS (row-major; V is
// transposed for f32 and read with the transposing LDS load for bf16), zero-padded from the head
// dim d to DP.  Each wave owns 32-query blocks.  For a block it
// computes S^T = K Q^T for ALL keys of the window (32x32 MFMA tiles: key = row = register,
// query = column = lane), so a softmax row lives in one lane pair (l, l+32): max and sum are
// register reductions plus one cross-half exchange, and the bias lookup is one LDS read per
// element from the (2w-1)^2 table (never expanded to [L,L]).  The un-normalised P^T tiles are
// then, as they stand in the accumulators, the B operand of O^T = V^T P^T (rows of the
// accumulator = the summed key index), so P never touches LDS; 1/sum is applied to O^T per lane.
// bf16 inputs use v_mfma_f32_32x32x16_bf16, f32 inputs the exact-f32 v_mfma_f32_32x32x2_f32.
#include <math.h>

#include "attn_common.hpp"

#pragma clang fp contract(fast)

namespace dsw {

template <typename T, int WS, int DP> struct FwdCfg {
    static constexpr int L = WS * WS;
    static constexpr int NT = L / 32;               // key tiles = query blocks
    static constexpr bool WIDE = WS == 16 && DP == 32 && sizeof(T) == 2;  // one wave per query block
    static constexpr int NW = WIDE ? 8 : (NT < 4 ? NT : 4);               // waves per workgroup
    static constexpr int THREADS = NW * 64;
    static constexpr int QS = DP + lds_traits<T>::RPAD;  // row-major stride (elements)
    static constexpr int VS = L + 4;                      // transposed stride (elements)
    // two workgroups per CU wherever the LDS footprint allows it
    static constexpr int MIN_WAVES = WIDE ? 4 : (WS == 16 && (DP == 64 || sizeof(T) == 4)) ? 1 : 2;
    static constexpr int TP = WIDE ? 2 : (NT < 4 ? NT : 4);  // key tiles per softmax pass
    static constexpr int N_V = sizeof(T) == 2 ? L * QS : DP * VS;  // bf16: V row-major; f32: V^T
    static constexpr size_t lds_bytes(bool bias) {
        return (size_t)(2 * L * QS + N_V) * sizeof(T) + (bias ? bias_lds<WS>::FLOATS * sizeof(float) : 0);
    }
};

template <typename T, int WS, int DP, bool BIAS, bool LEPE, bool DROP>
__global__ __launch_bounds__((FwdCfg<T, WS, DP>::THREADS), (FwdCfg<T, WS, DP>::MIN_WAVES)) void win_attn_fwd_kernel(
    AttnGeom g, const T *__restrict__ qkv, const float *__restrict__ table, float scale,
    const float *__restrict__ lepe_w, const float *__restrict__ lepe_b, T *__restrict__ out,
    float *__restrict__ lse, DropArgs drop) {
    using Cfg = FwdCfg<T, WS, DP>;
    constexpr int L = Cfg::L, NT = Cfg::NT, NW = Cfg::NW, THREADS = Cfg::THREADS, QS = Cfg::QS, VS = Cfg::VS;
    constexpr int TS = bias_lds<WS>::TS, TMAX = (2 * WS - 2) * (TS + 1);
    constexpr bool BF = sizeof(T) == 2;
    constexpr int TP = Cfg::TP, NPASS = NT / TP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *Qs = reinterpret_cast<T *>(smem);
    T *Ks = Qs + L * QS;
    T *Vt = Ks + L * QS;  // bf16: V [L][QS] row-major;  f32: V^T [DP][VS]
    float *tb = reinterpret_cast<float *>(Vt + Cfg::N_V);

    const int tid = threadIdx.x;
    const int id = xcd_remap(blockIdx.x, g.n_win * g.heads);
    const int win = id / g.heads, head = id - win * g.heads;
    const int d = g.d, C = g.ldo, C3 = g.ldq;  // row strides of out and qkv
    int qoff, koff, voff;
    qkv_offsets(g, head, qoff, koff, voff);
    // token (i, j) of this window sits at row origin + i*pitch + j
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

    // ---- stage Q (times scale*log2(e)), K and V ---------------------------------------------------
    // When d < DP the first padding column of V is set to ones: row d of O^T = V^T P^T is then the
    // softmax denominator, computed by the matrix core instead of 256 VALU adds per lane.
    // Loads are issued in batches of 8 per thread before any of them is consumed: a loop that
    // loads and stores one item per trip pays a full memory round trip per trip.
    const float sc2 = scale * LOG2E;
    auto rowbase = [&](int row) -> size_t { return (origin + (size_t)(row / WS) * pitch + (row % WS)) * C3; };
    if constexpr (BF) {
        if ((d & 1) == 0) {
            // a thread keeps one element pair and one window column; a sweep of all threads covers
            // RS = THREADS / (DP/2) consecutive tokens (a whole or half window row)
            constexpr int DW = DP / 2, RS = THREADS / DW, ITER = L / RS, BATCH = ITER < 8 ? ITER : 8;
            const int r0 = tid / DW, e = (tid - r0 * DW) * 2;
            const bool ok = e < d;
            const T *src0 = qkv + origin * C3 + min(e, d - 2);  // clamped: padding lanes re-read a valid pair
            T *qdst = Qs + r0 * QS + e, *kdst = Ks + r0 * QS + e, *vdst = Vt + r0 * QS + e;
            const uint32_t vpad = e == d ? 0x3f80u : 0u;  // bf16 1.0 in the low half
            for (int i0 = 0; i0 < ITER; i0 += BATCH) {
                uint32_t qv[BATCH], kv[BATCH], vv[BATCH];
#pragma unroll
                for (int i = 0; i < BATCH; ++i) {
                    const int tok = (i0 + i) * RS + r0;
                    const T *src = src0 + ((size_t)(tok / WS) * pitch + (tok % WS)) * C3;
                    qv[i] = *reinterpret_cast<const uint32_t *>(src + qoff);
                    kv[i] = *reinterpret_cast<const uint32_t *>(src + koff);
                    vv[i] = *reinterpret_cast<const uint32_t *>(src + voff);
                }
#pragma unroll
                for (int i = 0; i < BATCH; ++i) {
                    const int o = (i0 + i) * RS * QS;
                    const uint32_t q2 = pack_bf16(__uint_as_float(qv[i] << 16) * sc2, __uint_as_float(qv[i] & 0xffff0000u) * sc2);
                    *reinterpret_cast<uint32_t *>(qdst + o) = ok ? q2 : 0u;
                    *reinterpret_cast<uint32_t *>(kdst + o) = ok ? kv[i] : 0u;
                    *reinterpret_cast<uint32_t *>(vdst + o) = ok ? vv[i] : vpad;
                }
            }
        } else {
            for (int it = tid; it < L * DP; it += THREADS) {
                const int row = it / DP, e = it - row * DP;
                const size_t base = rowbase(row);
                T qv = 0, kv = 0, vv = (e == d) ? (T)0x3f80u : (T)0;
                if (e < d) {
                    qv = f32_to_bf16(bf16_to_f32(qkv[base + qoff + e]) * sc2);
                    kv = qkv[base + koff + e];
                    vv = qkv[base + voff + e];
                }
                Qs[row * QS + e] = qv;
                Ks[row * QS + e] = kv;
                Vt[row * QS + e] = vv;
            }
        }
    } else {
        constexpr int ITER = L * DP / THREADS, BATCH = 8;
        for (int i0 = 0; i0 < ITER; i0 += BATCH) {
            float qv[BATCH], kv[BATCH], vv[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int it = tid + (i0 + i) * THREADS;
                const int row = it / DP, e = it - row * DP;
                const T *src = qkv + rowbase(row) + min(e, d - 1);
                qv[i] = src[qoff];
                kv[i] = src[koff];
                vv[i] = src[voff];
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int it = tid + (i0 + i) * THREADS;
                const int row = it / DP, e = it - row * DP;
                const bool ok = e < d;
                Qs[row * QS + e] = ok ? qv[i] * sc2 : 0.f;
                Ks[row * QS + e] = ok ? kv[i] : 0.f;
                Vt[e * VS + row] = ok ? vv[i] : (e == d ? 1.f : 0.f);
            }
        }
    }
    if constexpr (BIAS) {
        constexpr int R = bias_lds<WS>::ROWS;
        for (int it = tid; it < R * R; it += THREADS) {
            const int di = it / R, dj = it - di * R;
            tb[TMAX - (di * TS + dj)] = table[(size_t)it * g.heads + head] * LOG2E;  // mirrored, see lane_base
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const bool ones_row = d < DP;  // uniform
    // transposing reads of V: lane 4q+p of a 16-lane group points at key row 4hh + q, head-dim
    // columns 16*(group & 1) + 4p.. of the 4-key x 16-column block the group receives
    const T *vtr = Vt + (4 * hh + ((lane >> 2) & 3)) * QS + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    for (int qb = wave; qb < NT; qb += NW) {
        const int q = qb * 32 + l31;
        const int qi = q / WS, qj = q - qi * WS;
        // this lane's Q fragments (B operand of S^T = K Q^T), kept for the whole block
        uint4 qfb[BF ? DP / 16 : 1];
        float4 qff[BF ? 1 : DP / 8];
        if constexpr (BF) {
#pragma unroll
            for (int s = 0; s < DP / 16; ++s) qfb[s] = *reinterpret_cast<const uint4 *>(Qs + q * QS + 16 * s + 8 * hh);
        } else {
            // lane half hh supplies the head-dim slice [hh*DP/2, (hh+1)*DP/2), one element per k-step
#pragma unroll
            for (int s = 0; s < DP / 8; ++s) qff[s] = *reinterpret_cast<const float4 *>(Qs + q * QS + hh * (DP / 2) + 4 * s);
        }
        f32x16 oacc[DP / 32];
#pragma unroll
        for (int u = 0; u < DP / 32; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[u][r] = 0.f;
        // element (t, r) of this lane is key 32t + rho(r) + 4hh; its table entry is
        // (qi - ki + w-1)*TS + (qj - kj + w-1) = E - (ki_c*TS + kj_c), with ki_c, kj_c the compile-time
        // part of the key coordinates.  The table is stored mirrored (entry e at TMAX - e), so the
        // four consecutive keys of registers 4g..4g+3 are four ascending words: paired reads land
        // directly in the accumulator registers.
        const int lane_base = TMAX - (qi * TS + qj + (WS - 1) * (TS + 1) - 4 * hh);
        float mx = -INFINITY, sum = 0.f;

        // The keys are walked in passes of TP tiles (128 keys) with an online softmax across passes,
        // which keeps the live score block at 64 accumulators per lane.
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            f32x16 sacc[TP];
            // ---- accumulators start from the bias ------------------------------------------------
#pragma unroll
            for (int tt = 0; tt < TP; ++tt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key_c = 32 * (pass * TP + tt) + rho(r);
                    if constexpr (BIAS) sacc[tt][r] = tb[lane_base + ((key_c / WS) * TS + (key_c % WS))];
                    else sacc[tt][r] = 0.f;
                }
            }
            // ---- S^T = K Q^T (+ bias), already in the base-2 domain ----------------------------------
#pragma unroll
            for (int tt = 0; tt < TP; ++tt) {
                const int t = pass * TP + tt;
                if constexpr (BF) {
#pragma unroll
                    for (int s = 0; s < DP / 16; ++s) {
                        const uint4 kf = *reinterpret_cast<const uint4 *>(Ks + (32 * t + l31) * QS + 16 * s + 8 * hh);
                        sacc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(kf), as_bf16x8(qfb[s]), sacc[tt], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < DP / 8; ++s) {
                        const float4 kf = *reinterpret_cast<const float4 *>(Ks + (32 * t + l31) * QS + hh * (DP / 2) + 4 * s);
                        sacc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qff[s].x, sacc[tt], 0, 0, 0);
                        sacc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qff[s].y, sacc[tt], 0, 0, 0);
                        sacc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qff[s].z, sacc[tt], 0, 0, 0);
                        sacc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qff[s].w, sacc[tt], 0, 0, 0);
                    }
                }
            }
            // ---- running max over registers + the other lane half; rescale what is accumulated ------
            float pm = mx;
#pragma unroll
            for (int tt = 0; tt < TP; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) pm = fmaxf(pm, sacc[tt][r]);
            pm = fmaxf(pm, __shfl_xor(pm, 32, 64));
            if (pass > 0) {
                const float alpha = __builtin_amdgcn_exp2f(mx - pm);
                sum *= alpha;
#pragma unroll
                for (int u = 0; u < DP / 32; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[u][r] *= alpha;
            }
            mx = pm;
            // ---- P^T = exp2(S^T - max) and O^T += V^T P^T ----------------------------------------------
#pragma unroll
            for (int tt = 0; tt < TP; ++tt) {
                const int t = pass * TP + tt;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {  // packed subtract: two elements per instruction
                    const f32x2 sh = f32x2{sacc[tt][r], sacc[tt][r + 1]} - f32x2{mx, mx};
                    sacc[tt][r] = __builtin_amdgcn_exp2f(sh.x);
                    sacc[tt][r + 1] = __builtin_amdgcn_exp2f(sh.y);
                }
                if (!ones_row || DROP) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum += sacc[tt][r];
                }
                if constexpr (DROP) {
                    // the denominator above is of the full row; what goes on to the product with V is thinned.
                    // Registers r, r+1 (r even) are the adjacent keys 32t + rho(r) + 4hh (+1): one hash per pair.
                    const uint32_t row_pairs = (((uint32_t)win * g.heads + head) * L + q) * (L / 2) + ((32 * t + 4 * hh) >> 1);
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const uint32_t h = drop_hash(row_pairs + (rho(r) >> 1), drop.seed_lo, drop.seed_hi);
                        sacc[tt][r] = (h & 0xffffu) >= drop.thr ? sacc[tt][r] : 0.f;
                        sacc[tt][r + 1] = (h >> 16) >= drop.thr ? sacc[tt][r + 1] : 0.f;
                    }
                }
                if constexpr (BF) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const uint4 pb = make_uint4(pack_bf16(sacc[tt][8 * s + 0], sacc[tt][8 * s + 1]),
                                                    pack_bf16(sacc[tt][8 * s + 2], sacc[tt][8 * s + 3]),
                                                    pack_bf16(sacc[tt][8 * s + 4], sacc[tt][8 * s + 5]),
                                                    pack_bf16(sacc[tt][8 * s + 6], sacc[tt][8 * s + 7]));
#pragma unroll
                        for (int u = 0; u < DP / 32; ++u) {
                            // A operand = V^T (row = head dim 32u + l31): element j of this lane is key
                            // 32t + 16s + 8(j>>2) + 4hh + (j&3), fetched as two transposing reads of V
                            const T *vblk = vtr + (32 * t + 16 * s) * QS + 32 * u;
                            const uint2 lo = lds_read_tr16(vblk);
                            const uint2 hi = lds_read_tr16(vblk + 8 * QS);
                            const uint4 va = make_uint4(lo.x, lo.y, hi.x, hi.y);
                            oacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(va), as_bf16x8(pb), oacc[u], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
                        for (int u = 0; u < DP / 32; ++u) {
                            const float4 va = *reinterpret_cast<const float4 *>(Vt + (32 * u + l31) * VS + 32 * t + 8 * g4 + 4 * hh);
                            oacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(va.x, sacc[tt][4 * g4 + 0], oacc[u], 0, 0, 0);
                            oacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(va.y, sacc[tt][4 * g4 + 1], oacc[u], 0, 0, 0);
                            oacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(va.z, sacc[tt][4 * g4 + 2], oacc[u], 0, 0, 0);
                            oacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(va.w, sacc[tt][4 * g4 + 3], oacc[u], 0, 0, 0);
                        }
                    }
                }
            }
        }

        // ---- softmax denominator: row d of O^T (ones row of V^T), or the explicit sum ---------------
        if (ones_row && !DROP) {
#pragma unroll
            for (int u = 0; u < DP / 32; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += (32 * u + rho(r) + 4 * hh == d) ? oacc[u][r] : 0.f;
        }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = (DROP ? drop.inv_keep : 1.0f) / sum;
        if (hh == 0) lse[((size_t)win * g.heads + head) * L + q] = (mx + __builtin_amdgcn_logf(sum)) * LN2;

        // ---- epilogue: normalise, LePE, store (lane = query row, registers = head-dim) -------------
        T *orow = out + (origin + (size_t)qi * pitch + qj) * C + head * d;
#pragma unroll
        for (int u = 0; u < DP / 32; ++u) {
            float o[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dd = 32 * u + rho(r) + 4 * hh;
                float v = oacc[u][r] * inv;
                if constexpr (LEPE) {
                    if (dd < d) {
                        const int c = head * d + dd;
                        float acc = lepe_b[c];
#pragma unroll
                        for (int a = -1; a <= 1; ++a) {
#pragma unroll
                            for (int bb = -1; bb <= 1; ++bb) {
                                const int ni = qi + a, nj = qj + bb;
                                if (ni >= 0 && ni < WS && nj >= 0 && nj < WS)
                                    acc += lepe_w[c * 9 + (a + 1) * 3 + (bb + 1)] * io<T>::ld(BF ? Vt + (ni * WS + nj) * QS + dd : Vt + dd * VS + ni * WS + nj);
                            }
                        }
                        v += acc;
                    }
                }
                o[r] = v;
            }
            if constexpr (BF) {
                if ((d & 1) == 0) {
                    // through the block's own (spent) Q rows in LDS, so that the global stores below
                    // are runs of whole output rows instead of one dword in each of 32 rows
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        *reinterpret_cast<uint2 *>(Qs + q * QS + 32 * u + 8 * g4 + 4 * hh) =
                            make_uint2(pack_bf16(o[4 * g4], o[4 * g4 + 1]), pack_bf16(o[4 * g4 + 2], o[4 * g4 + 3]));
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dd = 32 * u + rho(r) + 4 * hh;
                        if (dd < d) io<T>::st(orow + dd, o[r]);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dd = 32 * u + rho(r) + 4 * hh;
                    if (dd < d) io<T>::st(orow + dd, o[r]);
                }
            }
        }
        if (head == g.heads - 1 && C > g.C) {  // this block's share of the output rows' padding columns
            for (int i = lane; i < 32 * (C - g.C); i += 64) {
                const int qq = qb * 32 + i / (C - g.C);
                out[(origin + (size_t)(qq / WS) * pitch + (qq % WS)) * C + g.C + i % (C - g.C)] = (T)0;
            }
        }
        if constexpr (BF) {
            if ((d & 1) == 0) {
                // a wave's LDS accesses complete in order, and only this wave touches these rows
                constexpr int DWR = DP / 2, RPI = 64 / DWR;  // dwords per row, rows per store instruction
                const int dw = lane % DWR, rsub = lane / DWR;
#pragma unroll
                for (int i = 0; i < 32 / RPI; ++i) {
                    const int qq = qb * 32 + i * RPI + rsub;
                    const uint32_t v = *reinterpret_cast<const uint32_t *>(Qs + qq * QS + 2 * dw);
                    if (2 * dw < d)
                        *reinterpret_cast<uint32_t *>(out + (origin + (size_t)(qq / WS) * pitch + (qq % WS)) * C + head * d + 2 * dw) = v;
                }
            }
        }
    }
}

template <typename T, int WS, int DP>
static int launch_fwd(const AttnGeom &g, const void *qkv, const float *table, float scale, const float *lepe_w,
                      const float *lepe_b, void *out, float *lse, const DropArgs &drop, hipStream_t st) {
    using Cfg = FwdCfg<T, WS, DP>;
    const bool bias = table != nullptr, lepe = lepe_w != nullptr;
    const size_t lds = Cfg::lds_bytes(bias);
    if (lds > 160 * 1024) return fail(DSW_EUNSUPPORTED, "win_attn_fwd: w=%d head dim %d needs %zu B of LDS", WS, g.d, lds);
    const dim3 grid(g.n_win * g.heads), block(Cfg::THREADS);
#define DSW_FWD_LAUNCH2(B_, L_, D_)                                                                              \
    do {                                                                                                         \
        auto kern = win_attn_fwd_kernel<T, WS, DP, B_, L_, D_>;                                                  \
        if (lds > 48 * 1024) {                                                                                   \
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return fail(DSW_ELAUNCH, "win_attn_fwd: LDS attribute: %s", hipGetErrorString(e)); \
        }                                                                                                        \
        hipLaunchKernelGGL(kern, grid, block, lds, st, g, (const T *)qkv, table, scale, lepe_w, lepe_b, (T *)out, lse, drop); \
    } while (0)
#define DSW_FWD_LAUNCH(B_, L_)                     \
    do {                                           \
        if (drop.thr) DSW_FWD_LAUNCH2(B_, L_, true); \
        else DSW_FWD_LAUNCH2(B_, L_, false);       \
    } while (0)
    if (bias && !lepe) DSW_FWD_LAUNCH(true, false);
    else if (!bias && lepe) DSW_FWD_LAUNCH(false, true);
    else if (!bias && !lepe) DSW_FWD_LAUNCH(false, false);
    else DSW_FWD_LAUNCH(true, true);
#undef DSW_FWD_LAUNCH
#undef DSW_FWD_LAUNCH2
    return check_launch("win_attn_fwd_kernel");
}

int make_geom(const char *who, int n_img, int H, int W, int w, int C, int heads, int layout, int ld_qkv, int ld_out,
              int ld_dout, AttnGeom &g) {
    if (n_img <= 0 || C <= 0 || heads <= 0 || C % heads) return fail(DSW_EINVAL, "%s: bad sizes n_img=%d C=%d heads=%d", who, n_img, C, heads);
    if (w != 8 && w != 16) return fail(DSW_EUNSUPPORTED, "%s: window size %d (built: 8, 16)", who, w);
    if (C / heads > 64) return fail(DSW_EUNSUPPORTED, "%s: head dim %d > 64", who, C / heads);
    if (layout != DSW_QKV_CHUNKED && layout != DSW_QKV_INTERLEAVED) return fail(DSW_EINVAL, "%s: layout %d", who, layout);
    if ((H == 0) != (W == 0) || H < 0 || W < 0 || H % w || W % w) return fail(DSW_EINVAL, "%s: H,W (%d,%d) must both be 0 or multiples of %d", who, H, W, w);
    g.H = H;
    g.W = W;
    g.nWh = H ? H / w : 0;
    g.nWw = W ? W / w : 0;
    g.n_win = H ? n_img * g.nWh * g.nWw : n_img;
    g.C = C;
    g.heads = heads;
    g.d = C / heads;
    g.layout = layout;
    g.ldq = ld_qkv ? ld_qkv : 3 * C;
    g.ldo = ld_out ? ld_out : C;
    g.ldg = ld_dout ? ld_dout : C;
    if (g.ldq < 3 * C || g.ldo < C || g.ldg < C || (g.d % 2 == 0 && ((g.ldq | g.ldo | g.ldg) & 1)))
        return fail(DSW_EINVAL, "%s: row strides (%d, %d, %d) must be >= 3C, C, C (and even for an even head dim)", who,
                    g.ldq, g.ldo, g.ldg);
    return DSW_OK;
}

// dropout arguments of the C ABI -> kernel form; p in [0, 1)
int make_drop(const char *who, float p, unsigned long long seed, DropArgs &d) {
    if (!(p >= 0.f) || p >= 1.f) return fail(DSW_EINVAL, "%s: dropout probability %g outside [0, 1)", who, (double)p);
    const long thr = lroundf(p * 65536.f);
    d.thr = (uint32_t)(thr > 65535 ? 65535 : thr);
    d.seed_lo = (uint32_t)(seed & 0xffffffffull);
    d.seed_hi = (uint32_t)(seed >> 32);
    d.inv_keep = 1.0f / (1.0f - p);
    return DSW_OK;
}

}  // namespace dsw

using namespace dsw;

extern "C" int dsw_win_attn_fwd(const void *qkv, int dtype, int layout, int n_img, int H, int W, int w, int C,
                                int heads, int ld_qkv, int ld_out, const float *bias_table, float scale,
                                const float *lepe_w, const float *lepe_b, void *out, float *lse, float dropout_p,
                                unsigned long long dropout_seed, void *stream) {
    AttnGeom g;
    int rc = make_geom("dsw_win_attn_fwd", n_img, H, W, w, C, heads, layout, ld_qkv, ld_out, 0, g);
    if (rc) return rc;
    DropArgs drop;
    rc = make_drop("dsw_win_attn_fwd", dropout_p, dropout_seed, drop);
    if (rc) return rc;
    DSW_REQUIRE(qkv && out && lse, "dsw_win_attn_fwd: null pointer");
    DSW_REQUIRE((lepe_w == nullptr) == (lepe_b == nullptr), "dsw_win_attn_fwd: lepe_w and lepe_b go together");
    hipStream_t st = as_stream(stream);
    const bool big = g.d > 32;
    DSW_DISPATCH_DTYPE(dtype, T, {
        if (w == 16) return big ? launch_fwd<T, 16, 64>(g, qkv, bias_table, scale, lepe_w, lepe_b, out, lse, drop, st)
                                : launch_fwd<T, 16, 32>(g, qkv, bias_table, scale, lepe_w, lepe_b, out, lse, drop, st);
        return big ? launch_fwd<T, 8, 64>(g, qkv, bias_table, scale, lepe_w, lepe_b, out, lse, drop, st)
                   : launch_fwd<T, 8, 32>(g, qkv, bias_table, scale, lepe_w, lepe_b, out, lse, drop, st);
    });
    return DSW_OK;
}
