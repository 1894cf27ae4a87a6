// Window attention with the q|k|v projection inside the kernel (forward, bf16, head dim <= 32): the nn.Linear of
// DSwinT_arch.py:334 (WindowedMSA.qkv) / :303 (DS-MSA in_proj) and the attention of :135-172 in one launch, so the
// [tokens, 3C] q|k|v matrix never exists in memory.  One (window, head) per workgroup, as in attn_fwd.hip.
//
// Prologue.  The head's 3 x 32 rows of the prepared weight (zero rows past the head dim) are staged in LDS - in the
// space that Q and K take afterwards when they fit there, so the footprint stays at two workgroups per CU for
// 16 x 16 windows.  Each wave owns 32 tokens and computes Q^T, K^T, V^T = W_h X^T for them on the matrix cores
// (A = weight rows from LDS, B = the tokens' activation rows straight from global memory, 16 bytes per lane and
// k-step, prefetched four k-steps ahead): the result has the token on the lane and the head dim in the registers,
// so a lane writes its token's row of Q, K and V to LDS in 8-byte pieces.  Bias, the bf16 rounding of the
// projection's result and the scaling of q are applied on the way, in the order the separate kernels apply them
// (round q to bf16, scale by scale*log2(e), round again).  The rest is the shared forward core.
// The six heads of a window read the same activation rows; the workgroup order (xcd_remap) puts them on one XCD, so
// five of the six reads are L2 hits.
#include "attn_fwd_core.hpp"

#pragma clang fp contract(fast)

namespace dsw {
namespace {

constexpr int PROJ_DP = 32;
constexpr int PROJ_WROWS = 3 * PROJ_DP;

template <int WS> struct ProjCfg {
    using F = FwdCfg<bf16_t, WS, PROJ_DP>;
    static_assert(F::NT == F::NW, "one 32-token block per wave");
    static constexpr int L = F::L, THREADS = F::THREADS, QS = F::QS;
    static constexpr size_t QK_BYTES = (size_t)2 * L * QS * sizeof(bf16_t);
};

template <int WS, bool BIAS, int KS>
__global__ __launch_bounds__((ProjCfg<WS>::THREADS), (ProjCfg<WS>::F::MIN_WAVES)) void win_attn_proj_fwd_kernel(
    AttnGeom g, const bf16_t *__restrict__ x, long long ldx, const bf16_t *__restrict__ w_p, int Kp,
    const float *__restrict__ bias, const float *__restrict__ table, float scale, bf16_t *__restrict__ out,
    float *__restrict__ lse, int w_lds_off) {
    using Cfg = ProjCfg<WS>;
    using F = typename Cfg::F;
    constexpr int L = Cfg::L, THREADS = Cfg::THREADS, QS = Cfg::QS, DP = PROJ_DP;
    constexpr int TS = bias_lds<WS>::TS, TMAX = (2 * WS - 2) * (TS + 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Qs = reinterpret_cast<bf16_t *>(smem);
    bf16_t *Ks = Qs + L * QS;
    bf16_t *Vt = Ks + L * QS;
    float *tb = reinterpret_cast<float *>(Vt + F::N_V);
    bf16_t *Ws = reinterpret_cast<bf16_t *>(smem + w_lds_off);
    const int WSTR = Kp + 8;  // 16-byte reads of 8 consecutive rows fall on disjoint banks (row pitch = 16 mod 128 bytes)

    const int tid = threadIdx.x;
    const int id = xcd_remap(blockIdx.x, g.n_win * g.heads);
    const int win = __builtin_amdgcn_readfirstlane(id / g.heads), head = id - win * g.heads;  // (wave-uniform quotients of the vector unit's division, handed back to scalar registers)
    const int d = g.d;
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
    if (window_is_off(g, win)) {  // uniform over the workgroup
        zero_window_slice<bf16_t>(out, g.ldo, origin, pitch, WS, head * d, d, head == g.heads - 1, g.C, threadIdx.x, THREADS);
        if (lse != nullptr)
            for (int l = threadIdx.x; l < L; l += THREADS) lse[((size_t)win * g.heads + head) * L + l] = 0.f;
        return;
    }

    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int tok = wave * 32 + l31;
    const bf16_t *xrow = x + (origin + (size_t)(tok / WS) * pitch + (tok % WS)) * ldx + 8 * hh;
    const int ksteps = KS ? KS : (Kp >> 4);  // k-steps of 16 channels

    // ---- the head's weight rows -> LDS: row 32p + e = row (offset_p + e) of w_p, zero for e >= d.  Loads are issued in
    // batches before any is consumed, behind the requests for this lane's activation fragments (KS known), which then
    // arrive while the weights are being staged. ------------------------------------------------------------------------
    uint4 xf[KS ? KS : 1];
    if constexpr (KS > 0) {
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[s] = *reinterpret_cast<const uint4 *>(xrow + 16 * s);
    }
    {
        constexpr int WB = 5;  // 16-byte pieces per thread and batch
        const int vpr = Kp >> 3, total = PROJ_WROWS * vpr;
        for (int base = 0; base < total; base += THREADS * WB) {
            uint4 wv[WB];
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                const int it = base + i * THREADS + tid;
                const int row = it / vpr, c = it - row * vpr;
                const int p = row >> 5, e = row & 31;
                const int off = p == 0 ? qoff : (p == 1 ? koff : voff);
                wv[i] = make_uint4(0u, 0u, 0u, 0u);
                if (it < total && e < d) wv[i] = *reinterpret_cast<const uint4 *>(w_p + (size_t)(off + e) * Kp + 8 * c);
            }
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                const int it = base + i * THREADS + tid;
                const int row = it / vpr, c = it - row * vpr;
                if (it < total) *reinterpret_cast<uint4 *>(Ws + row * WSTR + 8 * c) = wv[i];
            }
        }
    }
    if constexpr (BIAS) {
        constexpr int R = bias_lds<WS>::ROWS;
        for (int it = tid; it < R * R; it += THREADS) {
            const int di = it / R, dj = it - di * R;
            tb[TMAX - (di * TS + dj)] = table[(size_t)it * g.heads + head] * LOG2E;  // mirrored, see the core's lane_base
        }
    }

    // ---- Q^T, K^T, V^T of this wave's 32 tokens --------------------------------------------------------------------
    f32x16 aq, ak, av;
#pragma unroll
    for (int r = 0; r < 16; ++r) aq[r] = ak[r] = av[r] = 0.f;
    const bf16_t *wlane = Ws + l31 * WSTR + 8 * hh;
    auto k_step = [&](const uint4 &xs, int ks) {
        const bf16_t *wk = wlane + 16 * ks;
        const uint4 fq = *reinterpret_cast<const uint4 *>(wk);
        const uint4 fk = *reinterpret_cast<const uint4 *>(wk + 32 * WSTR);
        const uint4 fv = *reinterpret_cast<const uint4 *>(wk + 64 * WSTR);
        aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fq), as_bf16x8(xs), aq, 0, 0, 0);
        ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fk), as_bf16x8(xs), ak, 0, 0, 0);
        av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fv), as_bf16x8(xs), av, 0, 0, 0);
    };
    if constexpr (KS > 0) {
        lds_barrier();  // the activation loads stay in flight across it
        // Fully unrolled and software-pipelined by hand: the weight fragments of k-step s + 1 are requested before the
        // MFMAs of k-step s are issued, and a scheduling fence closes every step - left alone, the scheduler lifts
        // all 3 * KS weight reads to the top of the block and spills.
        constexpr int AHEAD = 1;
        uint4 wf[KS][3];
        auto fetch = [&](int ks) {
            const bf16_t *wk = wlane + 16 * ks;
            wf[ks][0] = *reinterpret_cast<const uint4 *>(wk);
            wf[ks][1] = *reinterpret_cast<const uint4 *>(wk + 32 * WSTR);
            wf[ks][2] = *reinterpret_cast<const uint4 *>(wk + 64 * WSTR);
        };
#pragma unroll
        for (int s = 0; s < AHEAD && s < KS; ++s) fetch(s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + AHEAD < KS) fetch(s + AHEAD);
            aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wf[s][0]), as_bf16x8(xf[s]), aq, 0, 0, 0);
            ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wf[s][1]), as_bf16x8(xf[s]), ak, 0, 0, 0);
            av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wf[s][2]), as_bf16x8(xf[s]), av, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // any number of channels (a multiple of 64): fragments fetched four k-steps ahead of their use
        uint4 xa[4], xb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xa[s] = *reinterpret_cast<const uint4 *>(xrow + 16 * s);
        __syncthreads();
        const int ng = ksteps >> 2;
        for (int grp = 0; grp < ng; grp += 2) {
            if (grp + 1 < ng) {
#pragma unroll
                for (int s = 0; s < 4; ++s) xb[s] = *reinterpret_cast<const uint4 *>(xrow + 16 * (4 * (grp + 1) + s));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) k_step(xa[s], 4 * grp + s);
            if (grp + 1 < ng) {
                if (grp + 2 < ng) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) xa[s] = *reinterpret_cast<const uint4 *>(xrow + 16 * (4 * (grp + 2) + s));
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) k_step(xb[s], 4 * (grp + 1) + s);
            }
        }
    }
    // this lane's bias values: register r holds head-dim element rho(r) + 4hh
    float bq[16], bk[16], bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int e = rho(r) + 4 * hh;
        const bool ok = e < d && bias != nullptr;
        bq[r] = ok ? bias[qoff + e] : 0.f;
        bk[r] = ok ? bias[koff + e] : 0.f;
        bv[r] = ok ? bias[voff + e] : (e == d ? 1.f : 0.f);  // ones column of V: the softmax denominator (see the core)
    }
    __syncthreads();  // every wave is done with the weight rows: Q and K may take their place

    // ---- rows of Q, K, V in LDS: registers 4i..4i+3 are the head-dim elements 8i + 4hh .. + 3 -----------------------
    const float sc2 = scale * LOG2E;
    auto rb = [](float v) { return bf16_to_f32(f32_to_bf16(v)); };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = tok * QS + 8 * i + 4 * hh;
        float q4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) q4[j] = rb(aq[4 * i + j] + bq[4 * i + j]) * sc2;
        *reinterpret_cast<uint2 *>(Qs + o) = make_uint2(pack_bf16(q4[0], q4[1]), pack_bf16(q4[2], q4[3]));
        *reinterpret_cast<uint2 *>(Ks + o) = make_uint2(pack_bf16(ak[4 * i] + bk[4 * i], ak[4 * i + 1] + bk[4 * i + 1]),
                                                        pack_bf16(ak[4 * i + 2] + bk[4 * i + 2], ak[4 * i + 3] + bk[4 * i + 3]));
        *reinterpret_cast<uint2 *>(Vt + o) = make_uint2(pack_bf16(av[4 * i] + bv[4 * i], av[4 * i + 1] + bv[4 * i + 1]),
                                                        pack_bf16(av[4 * i + 2] + bv[4 * i + 2], av[4 * i + 3] + bv[4 * i + 3]));
    }
    __syncthreads();
    DropArgs nodrop{0u, 0u, 0u, 1.f};
    win_attn_fwd_core<bf16_t, WS, DP, BIAS, false, false>(g, Qs, Ks, Vt, tb, win, head, origin, pitch, nullptr, nullptr, out, lse,
                                                          nodrop);
}

template <int WS>
int launch_proj_fwd(const AttnGeom &g, const void *x, long long ldx, const void *w_p, int Kp, const float *bias,
                    const float *table, float scale, void *out, float *lse, hipStream_t st) {
    using Cfg = ProjCfg<WS>;
    const bool has_table = table != nullptr;
    const size_t att = Cfg::F::lds_bytes(has_table);
    const size_t wbytes = (size_t)PROJ_WROWS * (Kp + 8) * sizeof(bf16_t);
    const size_t w_off = wbytes <= Cfg::QK_BYTES ? 0 : (att + 15) & ~(size_t)15;
    const size_t lds = w_off ? w_off + wbytes : att;
    if (lds > 160 * 1024) return fail(DSW_EUNSUPPORTED, "win_attn_proj_fwd: w=%d with %d input channels needs %zu B of LDS", WS, Kp, lds);
    const dim3 grid(g.n_win * g.heads), block(Cfg::THREADS);
#define DSW_PROJ_LAUNCH2(B_, KS_)                                                                                            \
    do {                                                                                                                 \
        auto kern = win_attn_proj_fwd_kernel<WS, B_, KS_>;                                                               \
        if (lds > 48 * 1024) {                                                                                           \
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return fail(DSW_ELAUNCH, "win_attn_proj_fwd: LDS attribute: %s", hipGetErrorString(e)); \
        }                                                                                                                \
        hipLaunchKernelGGL(kern, grid, block, lds, st, g, (const bf16_t *)x, ldx, (const bf16_t *)w_p, Kp, bias, table, scale, \
                           (bf16_t *)out, lse, (int)w_off);                                                              \
    } while (0)
#define DSW_PROJ_LAUNCH(B_)                            \
    do {                                               \
        if (Kp == 192) DSW_PROJ_LAUNCH2(B_, 12);       \
        else if (Kp == 128) DSW_PROJ_LAUNCH2(B_, 8);   \
        else if (Kp == 64) DSW_PROJ_LAUNCH2(B_, 4);    \
        else DSW_PROJ_LAUNCH2(B_, 0);                  \
    } while (0)
    if (has_table) DSW_PROJ_LAUNCH(true);
    else DSW_PROJ_LAUNCH(false);
#undef DSW_PROJ_LAUNCH
#undef DSW_PROJ_LAUNCH2
    return check_launch("win_attn_proj_fwd_kernel");
}

}  // namespace

int make_geom(const char *who, int n_img, int H, int W, int w, int C, int heads, int layout, int ld_qkv, int ld_out,
              int ld_dout, AttnGeom &g);

}  // namespace dsw

using namespace dsw;

extern "C" int dsw_win_attn_proj_fwd(const void *x, long long ldx, const void *w_p, int w_rows, int ldw, const float *bias,
                                     int layout, int n_img, int H, int W, int w, int C, int heads, int ld_out,
                                     const float *bias_table, float scale, void *out, float *lse, const int *n_live, void *stream) {
    AttnGeom g;
    const int rc = make_geom("dsw_win_attn_proj_fwd", n_img, H, W, w, C, heads, layout, 0, ld_out, 0, g);
    if (rc) return rc;
    DSW_REQUIRE(n_live == nullptr || H == 0, "dsw_win_attn_proj_fwd: n_live goes with packed windows (H = W = 0)");
    g.n_live = n_live;
    DSW_REQUIRE(x && w_p && out, "dsw_win_attn_proj_fwd: null pointer");
    if (g.d > PROJ_DP) return fail(DSW_EUNSUPPORTED, "dsw_win_attn_proj_fwd: head dim %d > %d", g.d, PROJ_DP);
    if (ldw < C || ldw % 64 || w_rows < 3 * C)
        return fail(DSW_EINVAL, "dsw_win_attn_proj_fwd: prepared weight [%d][%d] for C=%d (rows >= 3C, row length a multiple of 64 >= C)",
                    w_rows, ldw, C);
    if (ldx < ldw || ldx % 8) return fail(DSW_EINVAL, "dsw_win_attn_proj_fwd: ldx=%lld (needs >= %d and a multiple of 8)", ldx, ldw);
    if (((uintptr_t)x | (uintptr_t)w_p) & 15) return fail(DSW_EINVAL, "dsw_win_attn_proj_fwd: x and w_p must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    if (w == 16) return launch_proj_fwd<16>(g, x, ldx, w_p, ldw, bias, bias_table, scale, out, lse, st);
    return launch_proj_fwd<8>(g, x, ldx, w_p, ldw, bias, bias_table, scale, out, lse, st);
}
