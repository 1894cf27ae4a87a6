// The part of the fused window-attention forward that follows the staging of Q, K and V in LDS: shared by the
// kernel that reads projected q|k|v rows (attn_fwd.hip) and the one that projects them itself (attn_proj_fwd.hip).
#pragma once
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
    // LePE values of the window, fp32 [L][LPS], computed once per workgroup ahead of the attention (bf16 kernels)
    static constexpr int LPS = DP + 4;  // 16-byte reads of four head-dim columns, lanes (= queries) spread over the banks
    static constexpr size_t lepe_lds_bytes() { return (size_t)L * LPS * sizeof(float); }
};

// LePE of one window and head (msft_arch.py:113-125: a depthwise 3x3 convolution of V inside the window, zero
// padding at its border) from the bf16 V image in LDS ([L][QS] row-major) into lp [L][LPS] fp32, for an even head
// dim.  A work item is (channel pair, window row): it walks the three neighbouring rows of V once, 3 * WS LDS
// reads for 2 * WS outputs (the per-element form needs 9 reads and 9 weight loads per output).  The caller puts
// a workgroup barrier before (V staged) and after (lp complete).
template <typename T, int WS, int DP, int THREADS>
__device__ __forceinline__ void lepe_values(const T *Vt, float *lp, const float *__restrict__ lepe_w,
                                            const float *__restrict__ lepe_b, int head, int d, int tid) {
    using Cfg = FwdCfg<T, WS, DP>;
    constexpr int QS = Cfg::QS, LPS = Cfg::LPS, DW = DP / 2;
    typedef float v2 __attribute__((ext_vector_type(2)));
    for (int item = tid; item < DW * WS; item += THREADS) {
        const int ep = item % DW, i = item / DW, e = 2 * ep;  // lanes along the channel pairs: consecutive LDS words
        if (e >= d) continue;
        const float *w0 = lepe_w + (size_t)(head * d + e) * 9, *w1 = w0 + 9;
        v2 acc[WS];
        const v2 b2 = {lepe_b[head * d + e], lepe_b[head * d + e + 1]};
#pragma unroll
        for (int j = 0; j < WS; ++j) acc[j] = b2;
#pragma unroll
        for (int a = -1; a <= 1; ++a) {
            const int row = i + a;
            if (row < 0 || row >= WS) continue;
            v2 x[WS];
#pragma unroll
            for (int j = 0; j < WS; ++j) {
                const uint32_t pr = *reinterpret_cast<const uint32_t *>(Vt + (row * WS + j) * QS + e);
                x[j] = v2{__uint_as_float(pr << 16), __uint_as_float(pr & 0xffff0000u)};
            }
#pragma unroll
            for (int bb = -1; bb <= 1; ++bb) {
                const v2 wt = {w0[(a + 1) * 3 + bb + 1], w1[(a + 1) * 3 + bb + 1]};
#pragma unroll
                for (int j = 0; j < WS; ++j)
                    if (j + bb >= 0 && j + bb < WS) acc[j] += wt * x[j + bb];
            }
        }
#pragma unroll
        for (int j = 0; j < WS; ++j) *reinterpret_cast<v2 *>(lp + (i * WS + j) * LPS + e) = acc[j];
    }
}

// Expects, in LDS: Qs [L][QS] = q * scale * log2(e), Ks [L][QS], Vt (bf16: V [L][QS]; f32: V^T [DP][VS]), all
// zero-padded from d to DP with the first padding column of V set to one, and (BIAS) the mirrored
// relative-position table tb; a workgroup barrier after the staging is the caller's.  `origin` / `pitch`
// address token (i, j) of the window at row origin + i * pitch + j of `out`.  lse may be null (not wanted).
// LEPE: lp, when not null, holds the window's LePE values [L][Cfg::LPS] in LDS (lepe_values below) and the epilogue
// adds them; with lp null the epilogue evaluates the 9 taps per output element itself.
template <typename T, int WS, int DP, bool BIAS, bool LEPE, bool DROP>
__device__ __forceinline__ void win_attn_fwd_core(const AttnGeom &g, T *Qs, T *Ks, T *Vt, const float *tb, int win, int head,
                                                  size_t origin, int pitch, const float *__restrict__ lepe_w,
                                                  const float *__restrict__ lepe_b, T *__restrict__ out,
                                                  float *__restrict__ lse, const DropArgs &drop, const float *lp = nullptr) {
    using Cfg = FwdCfg<T, WS, DP>;
    constexpr int L = Cfg::L, NT = Cfg::NT, NW = Cfg::NW, QS = Cfg::QS, VS = Cfg::VS;
    constexpr int TS = bias_lds<WS>::TS, TMAX = (2 * WS - 2) * (TS + 1);
    constexpr bool BF = sizeof(T) == 2;
    constexpr int TP = Cfg::TP, NPASS = NT / TP;
    const int tid = threadIdx.x;
    const int d = g.d, C = g.ldo;
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
        if (hh == 0 && lse) lse[((size_t)win * g.heads + head) * L + q] = (mx + __builtin_amdgcn_logf(sum)) * LN2;

        // ---- epilogue: normalise, LePE, store (lane = query row, registers = head-dim) -------------
        T *orow = out + (origin + (size_t)qi * pitch + qj) * C + head * d;
#pragma unroll
        for (int u = 0; u < DP / 32; ++u) {
            float o[16];
            float lpv[16];
            if constexpr (LEPE) {
                if (lp != nullptr) {  // registers 4g..4g+3 are four consecutive head-dim columns
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float4 t4 = *reinterpret_cast<const float4 *>(lp + q * Cfg::LPS + 32 * u + 8 * g4 + 4 * hh);
                        lpv[4 * g4] = t4.x;
                        lpv[4 * g4 + 1] = t4.y;
                        lpv[4 * g4 + 2] = t4.z;
                        lpv[4 * g4 + 3] = t4.w;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dd = 32 * u + rho(r) + 4 * hh;
                float v = oacc[u][r] * inv;
                if constexpr (LEPE) {
                    if (lp != nullptr) {
                        v += lpv[r];
                    } else if (dd < d) {
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
                // a wave-uniform 64-bit base and 32-bit lane offsets (a window spans at most WS image rows)
                T *obase = out + origin * C + head * d;
#pragma unroll
                for (int i = 0; i < 32 / RPI; ++i) {
                    const int qq = qb * 32 + i * RPI + rsub;
                    const uint32_t v = *reinterpret_cast<const uint32_t *>(Qs + qq * QS + 2 * dw);
                    const unsigned off = ((unsigned)(qq / WS) * pitch + (unsigned)(qq % WS)) * C + 2 * dw;
                    if (2 * dw < d) *reinterpret_cast<uint32_t *>(obase + off) = v;
                }
            }
        }
    }
}

}  // namespace dsw
