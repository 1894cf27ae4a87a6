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
#include "attn_fwd_core.hpp"

#pragma clang fp contract(fast)

namespace dsw {


template <typename T, int WS, int DP, bool BIAS, bool LEPE, bool DROP>
__global__ __launch_bounds__((FwdCfg<T, WS, DP>::THREADS), (FwdCfg<T, WS, DP>::MIN_WAVES)) void win_attn_fwd_kernel(
    AttnGeom g, const T *__restrict__ qkv, const float *__restrict__ table, float scale,
    const float *__restrict__ lepe_w, const float *__restrict__ lepe_b, T *__restrict__ out,
    float *__restrict__ lse, DropArgs drop, int lepe_in_lds) {
    using Cfg = FwdCfg<T, WS, DP>;
    constexpr int L = Cfg::L, THREADS = Cfg::THREADS, QS = Cfg::QS, VS = Cfg::VS;
    constexpr int TS = bias_lds<WS>::TS, TMAX = (2 * WS - 2) * (TS + 1);
    constexpr bool BF = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *Qs = reinterpret_cast<T *>(smem);
    T *Ks = Qs + L * QS;
    T *Vt = Ks + L * QS;  // bf16: V [L][QS] row-major;  f32: V^T [DP][VS]
    float *tb = reinterpret_cast<float *>(Vt + Cfg::N_V);

    const int tid = threadIdx.x;
    const int id = xcd_remap(blockIdx.x, g.n_win * g.heads);
    const int win = __builtin_amdgcn_readfirstlane(id / g.heads), head = id - win * g.heads;  // (wave-uniform quotients of the vector unit's division, handed back to scalar registers)
    const int d = g.d, C3 = g.ldq;  // C3: row stride of qkv
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
        const int b = __builtin_amdgcn_readfirstlane(win / per), r = win - b * per;
        const int wr = __builtin_amdgcn_readfirstlane(r / g.nWw), wc = r - wr * g.nWw;
        origin = ((size_t)b * g.H + wr * WS) * g.W + wc * WS;
        pitch = g.W;
    }
    if (window_is_off(g, win)) {  // uniform over the workgroup
        zero_window_slice<T>(out, g.ldo, origin, pitch, WS, head * d, d, head == g.heads - 1, g.C, tid, THREADS);
        if (lse != nullptr)
            for (int l = tid; l < L; l += THREADS) lse[((size_t)win * g.heads + head) * L + l] = 0.f;
        return;
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
            // a wave-uniform 64-bit base and 32-bit lane offsets (the rows of a window span at most WS image rows):
            // the address arithmetic of a load is one 32-bit add instead of a 64-bit multiply-add
            const T *src0 = qkv + origin * C3;
            const unsigned ecol = (unsigned)min(e, d - 2);  // clamped: padding lanes re-read a valid pair
            static_assert(RS % WS == 0 || WS % RS == 0, "a sweep is a whole number of window rows or a part of one");
            const unsigned rel0 = ((unsigned)(r0 / WS) * pitch + (unsigned)(r0 % WS)) * C3 + ecol;
            const unsigned step = RS >= WS ? (unsigned)(RS / WS) * pitch * C3 : 0u;
            T *qdst = Qs + r0 * QS + e, *kdst = Ks + r0 * QS + e, *vdst = Vt + r0 * QS + e;
            const uint32_t vpad = e == d ? 0x3f80u : 0u;  // bf16 1.0 in the low half
            for (int i0 = 0; i0 < ITER; i0 += BATCH) {
                uint32_t qv[BATCH], kv[BATCH], vv[BATCH];
#pragma unroll
                for (int i = 0; i < BATCH; ++i) {
                    unsigned rel;
                    if constexpr (RS >= WS) {
                        rel = rel0 + (unsigned)(i0 + i) * step;
                    } else {
                        const int tok = (i0 + i) * RS + r0;
                        rel = ((unsigned)(tok / WS) * pitch + (unsigned)(tok % WS)) * C3 + ecol;
                    }
                    qv[i] = *reinterpret_cast<const uint32_t *>(src0 + (rel + qoff));
                    kv[i] = *reinterpret_cast<const uint32_t *>(src0 + (rel + koff));
                    vv[i] = *reinterpret_cast<const uint32_t *>(src0 + (rel + voff));
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
    const float *lp = nullptr;
    if constexpr (LEPE && BF) {
        if (lepe_in_lds) {  // uniform; the launcher sets it for an even head dim when the extra LDS fits
            float *lpw = tb + (BIAS ? bias_lds<WS>::FLOATS : 0);
            lepe_values<T, WS, DP, THREADS>(Vt, lpw, lepe_w, lepe_b, head, d, tid);
            __syncthreads();
            lp = lpw;
        }
    }

    win_attn_fwd_core<T, WS, DP, BIAS, LEPE, DROP>(g, Qs, Ks, Vt, tb, win, head, origin, pitch, lepe_w, lepe_b, out, lse, drop, lp);
}

template <typename T, int WS, int DP>
static int launch_fwd(const AttnGeom &g, const void *qkv, const float *table, float scale, const float *lepe_w,
                      const float *lepe_b, void *out, float *lse, const DropArgs &drop, hipStream_t st) {
    using Cfg = FwdCfg<T, WS, DP>;
    const bool bias = table != nullptr, lepe = lepe_w != nullptr;
    size_t lds = Cfg::lds_bytes(bias);
    if (lds > 160 * 1024) return fail(DSW_EUNSUPPORTED, "win_attn_fwd: w=%d head dim %d needs %zu B of LDS", WS, g.d, lds);
    // bf16, even head dim: the window's LePE values are computed once into LDS when they fit beside the operands
    const int lepe_in_lds = lepe && sizeof(T) == 2 && (g.d & 1) == 0 && lds + Cfg::lepe_lds_bytes() <= 160 * 1024;
    if (lepe_in_lds) lds += Cfg::lepe_lds_bytes();
    const dim3 grid(g.n_win * g.heads), block(Cfg::THREADS);
#define DSW_FWD_LAUNCH2(B_, L_, D_)                                                                              \
    do {                                                                                                         \
        auto kern = win_attn_fwd_kernel<T, WS, DP, B_, L_, D_>;                                                  \
        if (lds > 48 * 1024) {                                                                                   \
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return fail(DSW_ELAUNCH, "win_attn_fwd: LDS attribute: %s", hipGetErrorString(e)); \
        }                                                                                                        \
        hipLaunchKernelGGL(kern, grid, block, lds, st, g, (const T *)qkv, table, scale, lepe_w, lepe_b, (T *)out, lse, drop, lepe_in_lds); \
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
    g.n_live = nullptr;
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

// DSW_ATTN_FWD_GENERAL=1 in the environment sends every shape to the general kernel (tests compare the two)
bool force_general_fwd() {
    const char *e = getenv("DSW_ATTN_FWD_GENERAL");
    return e != nullptr && e[0] == '1';
}

}  // namespace dsw

using namespace dsw;

extern "C" int dsw_win_attn_fwd(const void *qkv, int dtype, int layout, int n_img, int H, int W, int w, int C,
                                int heads, int ld_qkv, int ld_out, const float *bias_table, float scale,
                                const float *lepe_w, const float *lepe_b, void *out, float *lse, float dropout_p,
                                unsigned long long dropout_seed, const int *n_live, void *stream) {
    AttnGeom g;
    int rc = make_geom("dsw_win_attn_fwd", n_img, H, W, w, C, heads, layout, ld_qkv, ld_out, 0, g);
    if (rc) return rc;
    DSW_REQUIRE(n_live == nullptr || H == 0, "dsw_win_attn_fwd: n_live goes with packed windows (H = W = 0)");
    g.n_live = n_live;
    DropArgs drop;
    rc = make_drop("dsw_win_attn_fwd", dropout_p, dropout_seed, drop);
    if (rc) return rc;
    DSW_REQUIRE(qkv && out && lse, "dsw_win_attn_fwd: null pointer");
    DSW_REQUIRE((lepe_w == nullptr) == (lepe_b == nullptr), "dsw_win_attn_fwd: lepe_w and lepe_b go together");
    hipStream_t st = as_stream(stream);
    const bool big = g.d > 32;
    if (dtype == DSW_BF16 && w == 16 && lepe_w == nullptr && drop.thr == 0 && !force_general_fwd()) {
        rc = launch_fwd_w16(g, qkv, bias_table, scale, out, lse, st);  // even head dims <= 30 (attn_fwd_w16.hip)
        if (rc >= 0) return rc;
    }
    DSW_DISPATCH_DTYPE(dtype, T, {
        if (w == 16) return big ? launch_fwd<T, 16, 64>(g, qkv, bias_table, scale, lepe_w, lepe_b, out, lse, drop, st)
                                : launch_fwd<T, 16, 32>(g, qkv, bias_table, scale, lepe_w, lepe_b, out, lse, drop, st);
        return big ? launch_fwd<T, 8, 64>(g, qkv, bias_table, scale, lepe_w, lepe_b, out, lse, drop, st)
                   : launch_fwd<T, 8, 32>(g, qkv, bias_table, scale, lepe_w, lepe_b, out, lse, drop, st);
    });
    return DSW_OK;
}
