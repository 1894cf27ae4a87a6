// Fused window attention, forward: softmax(q k^T * scale + bias) v (+ LePE) for one
// (window, head) per workgroup, on the gfx950 matrix cores.
// Reference: DSwinT_arch.py:122-125, :135-172, :303-308, :333-337; msft_arch.py:94-139.
//
// Structure.  The window's Q, K (row-major) and V (transposed) head slices are staged once in
// LDS, zero-padded from the head dim d to DP.  Each wave owns 32-query blocks.  For a block it
// computes S^T = K Q^T for ALL keys of the window (32x32 MFMA tiles: key = row = register,
// query = column = lane), so a softmax row lives in one lane pair (l, l+32): max and sum are
// register reductions plus one cross-half exchange, and the bias lookup is one LDS read per
// element from the (2w-1)^2 table (never expanded to [L,L]).  The un-normalised P^T tiles are
// then, as they stand in the accumulators, the B operand of O^T = V^T P^T (rows of the
// accumulator = the summed key index), so P never touches LDS; 1/sum is applied to O^T per lane.
// bf16 inputs use v_mfma_f32_32x32x16_bf16, f32 inputs the exact-f32 v_mfma_f32_32x32x2_f32.
#include "attn_common.hpp"

#pragma clang fp contract(fast)

namespace dsw {

template <typename T, int WS, int DP> struct FwdCfg {
    static constexpr int L = WS * WS;
    static constexpr int NT = L / 32;               // key tiles = query blocks
    static constexpr int NW = NT < 4 ? NT : 4;      // waves per workgroup
    static constexpr int THREADS = NW * 64;
    static constexpr int QS = DP + lds_traits<T>::RPAD;  // row-major stride (elements)
    static constexpr int VS = L + 4;                      // transposed stride (elements)
    // w = 16 keeps a 32 x 256 score block (128 accumulators) per lane: one wave per SIMD
    static constexpr int MIN_WAVES = WS == 16 ? 1 : 2;
    static constexpr size_t lds_bytes(bool bias) {
        return (size_t)(2 * L * QS + DP * VS) * sizeof(T) + (bias ? bias_lds<WS>::FLOATS * sizeof(float) : 0);
    }
};

template <typename T, int WS, int DP, bool BIAS, bool LEPE>
__global__ __launch_bounds__((FwdCfg<T, WS, DP>::THREADS), (FwdCfg<T, WS, DP>::MIN_WAVES)) void win_attn_fwd_kernel(
    AttnGeom g, const T *__restrict__ qkv, const float *__restrict__ table, float scale,
    const float *__restrict__ lepe_w, const float *__restrict__ lepe_b, T *__restrict__ out,
    float *__restrict__ lse) {
    using Cfg = FwdCfg<T, WS, DP>;
    constexpr int L = Cfg::L, NT = Cfg::NT, NW = Cfg::NW, THREADS = Cfg::THREADS, QS = Cfg::QS, VS = Cfg::VS;
    constexpr int TS = bias_lds<WS>::TS;
    constexpr bool BF = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *Qs = reinterpret_cast<T *>(smem);
    T *Ks = Qs + L * QS;
    T *Vt = Ks + L * QS;
    float *tb = reinterpret_cast<float *>(Vt + DP * VS);

    const int tid = threadIdx.x;
    const int id = xcd_remap(blockIdx.x, g.n_win * g.heads);
    const int win = id / g.heads, head = id - win * g.heads;
    const int d = g.d, C = g.C, C3 = 3 * g.C;
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

    // ---- stage Q, K (row-major) and V (transposed) -------------------------------------------
    if constexpr (BF) {
        if ((d & 1) == 0) {
            constexpr int DW = DP / 2;
            for (int it = tid; it < L * DW; it += THREADS) {
                const int row = it / DW, e = (it - row * DW) * 2;
                const size_t base = (origin + (size_t)(row / WS) * pitch + (row % WS)) * C3;
                uint32_t qv = 0, kv = 0, vv = 0;
                if (e < d) {
                    qv = *reinterpret_cast<const uint32_t *>(qkv + base + qoff + e);
                    kv = *reinterpret_cast<const uint32_t *>(qkv + base + koff + e);
                    vv = *reinterpret_cast<const uint32_t *>(qkv + base + voff + e);
                }
                *reinterpret_cast<uint32_t *>(Qs + row * QS + e) = qv;
                *reinterpret_cast<uint32_t *>(Ks + row * QS + e) = kv;
                Vt[e * VS + row] = (T)(vv & 0xffffu);
                Vt[(e + 1) * VS + row] = (T)(vv >> 16);
            }
        } else {
            for (int it = tid; it < L * DP; it += THREADS) {
                const int row = it / DP, e = it - row * DP;
                const size_t base = (origin + (size_t)(row / WS) * pitch + (row % WS)) * C3;
                T qv = 0, kv = 0, vv = 0;
                if (e < d) {
                    qv = qkv[base + qoff + e];
                    kv = qkv[base + koff + e];
                    vv = qkv[base + voff + e];
                }
                Qs[row * QS + e] = qv;
                Ks[row * QS + e] = kv;
                Vt[e * VS + row] = vv;
            }
        }
    } else {
        for (int it = tid; it < L * DP; it += THREADS) {
            const int row = it / DP, e = it - row * DP;
            const size_t base = (origin + (size_t)(row / WS) * pitch + (row % WS)) * C3;
            T qv = 0, kv = 0, vv = 0;
            if (e < d) {
                qv = qkv[base + qoff + e];
                kv = qkv[base + koff + e];
                vv = qkv[base + voff + e];
            }
            Qs[row * QS + e] = qv;
            Ks[row * QS + e] = kv;
            Vt[e * VS + row] = vv;
        }
    }
    if constexpr (BIAS) {
        constexpr int R = bias_lds<WS>::ROWS;
        for (int it = tid; it < R * R; it += THREADS) {
            const int di = it / R, dj = it - di * R;
            tb[di * TS + dj] = table[(size_t)it * g.heads + head] * LOG2E;
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const float sc2 = scale * LOG2E;

    for (int qb = wave; qb < NT; qb += NW) {
        const int q = qb * 32 + l31;
        const int qi = q / WS, qj = q - qi * WS;
        f32x16 sacc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[t][r] = 0.f;

        // ---- S^T = K Q^T ------------------------------------------------------------------------
        if constexpr (BF) {
            uint4 qf[DP / 16];
#pragma unroll
            for (int s = 0; s < DP / 16; ++s) qf[s] = *reinterpret_cast<const uint4 *>(Qs + q * QS + 16 * s + 8 * hh);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int s = 0; s < DP / 16; ++s) {
                    const uint4 kf = *reinterpret_cast<const uint4 *>(Ks + (32 * t + l31) * QS + 16 * s + 8 * hh);
                    sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(kf), as_bf16x8(qf[s]), sacc[t], 0, 0, 0);
                }
            }
        } else {
            // lane half hh supplies the head-dim slice [hh*DP/2, (hh+1)*DP/2), one element per k-step
            float4 qf[DP / 8];
#pragma unroll
            for (int s = 0; s < DP / 8; ++s) qf[s] = *reinterpret_cast<const float4 *>(Qs + q * QS + hh * (DP / 2) + 4 * s);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int s = 0; s < DP / 8; ++s) {
                    const float4 kf = *reinterpret_cast<const float4 *>(Ks + (32 * t + l31) * QS + hh * (DP / 2) + 4 * s);
                    sacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[s].x, sacc[t], 0, 0, 0);
                    sacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[s].y, sacc[t], 0, 0, 0);
                    sacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[s].z, sacc[t], 0, 0, 0);
                    sacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[s].w, sacc[t], 0, 0, 0);
                }
            }
        }

        // ---- softmax over the keys (registers + the other lane half), base 2 -----------------------
        // element (t, r) of this lane is key 32t + rho(r) + 4hh; its table entry is
        // (qi - ki + w-1)*TS + (qj - kj + w-1) = lane_base - (ki_c*TS + kj_c) with ki_c, kj_c
        // the compile-time part of the key coordinates.
        const int lane_base = qi * TS + qj + (WS - 1) * (TS + 1) - 4 * hh;
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key_c = 32 * t + rho(r);
                float v = sacc[t][r] * sc2;
                if constexpr (BIAS) v += tb[lane_base - ((key_c / WS) * TS + (key_c % WS))];
                sacc[t][r] = v;
                mx = fmaxf(mx, v);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep one tile's table reads in flight, not all
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
        uint32_t pk[BF ? NT : 1][8];  // bf16 P^T, packed pairwise: the PV B operand
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(sacc[t][r] - mx);
                sacc[t][r] = p;
                sum += p;
            }
            if constexpr (BF) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) pk[t][r >> 1] = pack_bf16(sacc[t][r], sacc[t][r + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        if (hh == 0) lse[((size_t)win * g.heads + head) * L + q] = (mx + __builtin_amdgcn_logf(sum)) * LN2;

        // ---- O^T = V^T P^T ------------------------------------------------------------------------
        f32x16 oacc[DP / 32];
#pragma unroll
        for (int u = 0; u < DP / 32; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[u][r] = 0.f;
        if constexpr (BF) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const uint4 pb = make_uint4(pk[t][4 * s + 0], pk[t][4 * s + 1], pk[t][4 * s + 2], pk[t][4 * s + 3]);
#pragma unroll
                    for (int u = 0; u < DP / 32; ++u) {
                        const T *vrow = Vt + (32 * u + l31) * VS + 32 * t + 16 * s + 4 * hh;
                        const uint2 lo = *reinterpret_cast<const uint2 *>(vrow);
                        const uint2 hi = *reinterpret_cast<const uint2 *>(vrow + 8);
                        const uint4 va = make_uint4(lo.x, lo.y, hi.x, hi.y);
                        oacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(va), as_bf16x8(pb), oacc[u], 0, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
                    for (int u = 0; u < DP / 32; ++u) {
                        const float4 va = *reinterpret_cast<const float4 *>(Vt + (32 * u + l31) * VS + 32 * t + 8 * g4 + 4 * hh);
                        oacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(va.x, sacc[t][4 * g4 + 0], oacc[u], 0, 0, 0);
                        oacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(va.y, sacc[t][4 * g4 + 1], oacc[u], 0, 0, 0);
                        oacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(va.z, sacc[t][4 * g4 + 2], oacc[u], 0, 0, 0);
                        oacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(va.w, sacc[t][4 * g4 + 3], oacc[u], 0, 0, 0);
                    }
                }
            }
        }

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
                                    acc += lepe_w[c * 9 + (a + 1) * 3 + (bb + 1)] * io<T>::ld(Vt + dd * VS + ni * WS + nj);
                            }
                        }
                        v += acc;
                    }
                }
                o[r] = v;
            }
            if constexpr (BF) {
                if ((d & 1) == 0) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int dd = 32 * u + rho(r) + 4 * hh;
                        if (dd < d) *reinterpret_cast<uint32_t *>(orow + dd) = pack_bf16(o[r], o[r + 1]);
                    }
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
    }
}

template <typename T, int WS, int DP>
static int launch_fwd(const AttnGeom &g, const void *qkv, const float *table, float scale, const float *lepe_w,
                      const float *lepe_b, void *out, float *lse, hipStream_t st) {
    using Cfg = FwdCfg<T, WS, DP>;
    const bool bias = table != nullptr, lepe = lepe_w != nullptr;
    const size_t lds = Cfg::lds_bytes(bias);
    if (lds > 160 * 1024) return fail(DSW_EUNSUPPORTED, "win_attn_fwd: w=%d head dim %d needs %zu B of LDS", WS, g.d, lds);
    const dim3 grid(g.n_win * g.heads), block(Cfg::THREADS);
#define DSW_FWD_LAUNCH(B_, L_)                                                                                   \
    do {                                                                                                         \
        auto kern = win_attn_fwd_kernel<T, WS, DP, B_, L_>;                                                      \
        if (lds > 48 * 1024) {                                                                                   \
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return fail(DSW_ELAUNCH, "win_attn_fwd: LDS attribute: %s", hipGetErrorString(e)); \
        }                                                                                                        \
        hipLaunchKernelGGL(kern, grid, block, lds, st, g, (const T *)qkv, table, scale, lepe_w, lepe_b, (T *)out, lse); \
    } while (0)
    if (bias && !lepe) DSW_FWD_LAUNCH(true, false);
    else if (!bias && lepe) DSW_FWD_LAUNCH(false, true);
    else if (!bias && !lepe) DSW_FWD_LAUNCH(false, false);
    else DSW_FWD_LAUNCH(true, true);
#undef DSW_FWD_LAUNCH
    return check_launch("win_attn_fwd_kernel");
}

int make_geom(const char *who, int n_img, int H, int W, int w, int C, int heads, int layout, AttnGeom &g) {
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
    return DSW_OK;
}

}  // namespace dsw

using namespace dsw;

extern "C" int dsw_win_attn_fwd(const void *qkv, int dtype, int layout, int n_img, int H, int W, int w, int C,
                                int heads, const float *bias_table, float scale, const float *lepe_w,
                                const float *lepe_b, void *out, float *lse, void *stream) {
    AttnGeom g;
    int rc = make_geom("dsw_win_attn_fwd", n_img, H, W, w, C, heads, layout, g);
    if (rc) return rc;
    DSW_REQUIRE(qkv && out && lse, "dsw_win_attn_fwd: null pointer");
    DSW_REQUIRE((lepe_w == nullptr) == (lepe_b == nullptr), "dsw_win_attn_fwd: lepe_w and lepe_b go together");
    hipStream_t st = as_stream(stream);
    const bool big = g.d > 32;
    DSW_DISPATCH_DTYPE(dtype, T, {
        if (w == 16) return big ? launch_fwd<T, 16, 64>(g, qkv, bias_table, scale, lepe_w, lepe_b, out, lse, st)
                                : launch_fwd<T, 16, 32>(g, qkv, bias_table, scale, lepe_w, lepe_b, out, lse, st);
        return big ? launch_fwd<T, 8, 64>(g, qkv, bias_table, scale, lepe_w, lepe_b, out, lse, st)
                   : launch_fwd<T, 8, 32>(g, qkv, bias_table, scale, lepe_w, lepe_b, out, lse, st);
    });
    return DSW_OK;
}
