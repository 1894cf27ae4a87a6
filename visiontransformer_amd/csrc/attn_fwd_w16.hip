// Fused window attention, forward, for 16 x 16 windows in bf16 with an even head dim <= 30 (no LePE, no dropout):
// softmax(q k^T * scale + bias) v for one (window, head) per workgroup of 8 waves, a wave per 32-query block.
// Reference: DSwinT_arch.py:122-125, :135-172, :303-308, :333-337.
//
// What sets it apart from the general kernel (attn_fwd.hip):
//  * a head's slice of a token row is read as 16-byte slots (4-byte aligned) instead of dwords;
//  * Q never enters LDS: a wave reads its own block's fragments from global memory straight into the B operand
//    registers and scales them there; K and V are XOR-swizzled images (slot_flip) whose 16-byte row reads and
//    transposing reads are bank-conflict-free; 39 KB of LDS per workgroup, several workgroups per CU;
//  * the row reference r of the softmax (what is subtracted from the logits before the exponential) rides in the
//    two last padding columns of the head dimension: Q' holds -r split into two bf16 (hi + lo) there against ones
//    in K, so the matrix core delivers S - r and the exponential follows without a subtraction.  r is the row
//    maximum of the FIRST key tile: no maximum, no rescaling of the running output for the other seven, and the
//    result is the exact softmax whatever r is, since numerator and denominator share it - unless a later logit
//    exceeds r by so much that an exponential overflows, which the denominator shows; the block is then done again
//    with the maximum over all keys;
//  * the denominator is row 31 of O^T (a column of ones in V's padding), computed by the matrix core;
//  * the output leaves through the wave's rows of the (spent) K image, so that one store instruction carries the
//    complete 60-byte pieces of 16 token rows (store_tile_rows_d30).
#include "attn_common.hpp"

#pragma clang fp contract(fast)

namespace dsw {

struct FwdW16Cfg {
    static constexpr int WS = 16, L = 256, DP = 32, NT = 8, THREADS = 512;
    static constexpr int N_IMG = L * DP;  // elements of one operand image
    static constexpr int TBF = bias_lds<16>::FLOATS;
    static constexpr size_t lds_bytes(bool bias) { return (size_t)2 * N_IMG * sizeof(bf16_t) + (bias ? (size_t)TBF * sizeof(float) : 0); }
};

// The head dim is 30: a head's slice of a token row is three whole 16-byte slots and one of 12 bytes.  The last
// slot is read (and written) as the 16 bytes that END with the slice - one dword early, so that no byte outside the
// slice is touched - and shifted in registers; its fourth dword, columns 30 and 31 of the padded head dim, is ours.
#ifndef FWD_W16_WAVES
#define FWD_W16_WAVES 6
#endif
template <bool BIAS>
__global__ __launch_bounds__(512, FWD_W16_WAVES) void win_attn_fwd_w16_kernel(AttnGeom g, const bf16_t *__restrict__ qkv,
                                                                  const float *__restrict__ table, float scale,
                                                                  bf16_t *__restrict__ out, float *__restrict__ lse) {
    using Cfg = FwdW16Cfg;
    using T = bf16_t;
    constexpr int WS = 16, L = Cfg::L, DP = Cfg::DP, NT = Cfg::NT, THREADS = Cfg::THREADS, D = 30;
    constexpr int TS = bias_lds<WS>::TS, TR = bias_lds<WS>::ROWS, TMAX = (2 * WS - 2) * (TS + 1);
    constexpr int BLK = 32 * DP;  // elements of a 32-row block of an image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *Ki = reinterpret_cast<T *>(smem);
    T *Vi = Ki + Cfg::N_IMG;
    float *tb = reinterpret_cast<float *>(Vi + Cfg::N_IMG);  // BIAS: mirrored table, entry e at TMAX - e

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int C = g.ldo, C3 = g.ldq;
    const int pitch = g.H == 0 ? WS : g.W;
    // where the work item of workgroup number `blk` lives: window, head, column offsets, and the token row of window
    // position 0 (token (i, j) of the window sits at row org + i*pitch + j)
    auto locate = [&](int blk, int &win_, int &head_, int &qo, int &ko, int &vo, size_t &org) {
        const int id = xcd_remap(blk, g.n_win * g.heads);
        // (wave-uniform quotients of the vector unit's division, handed back to scalar registers)
        win_ = __builtin_amdgcn_readfirstlane(id / g.heads);
        head_ = id - win_ * g.heads;
        qkv_offsets(g, head_, qo, ko, vo);
        if (g.H == 0) {
            org = (size_t)win_ * L;
        } else {
            const int per = g.nWh * g.nWw;
            const int b = __builtin_amdgcn_readfirstlane(win_ / per), r = win_ - b * per;
            const int wr = __builtin_amdgcn_readfirstlane(r / g.nWw), wc = r - wr * g.nWw;
            org = ((size_t)b * g.H + wr * WS) * g.W + wc * WS;
        }
    };
    int win, head, qoff, koff, voff;
    size_t origin;
    locate(blockIdx.x, win, head, qoff, koff, voff, origin);
    // windows from this number on are switched off (read once)
    const int n_live = g.n_live != nullptr ? __builtin_amdgcn_readfirstlane(*g.n_live) : 0x7fffffff;
    if (win >= n_live) {  // uniform over the workgroup
        zero_window_slice<T>(out, g.ldo, origin, pitch, WS, head * D, D, head == g.heads - 1, g.C, tid, THREADS);
        for (int l = tid; l < L; l += THREADS) lse[((size_t)win * g.heads + head) * L + l] = 0.f;
        return;
    }
    auto rel_row = [&](int l) -> unsigned { return (unsigned)((l >> 4) * pitch + (l & 15)); };
    const T *qkv0 = qkv + origin * C3;  // wave-uniform 64-bit base, 32-bit lane offsets

    // ---- loads: this wave's Q fragments (slots hh and 2 + hh of row 32 wave + l31), then K and V --------------
    // (the head's column of the bias table first, both sweeps at once: back before the operands)
    static_assert(TR * TR <= 2 * THREADS, "two table entries per thread");
    float tabv[2] = {0.f, 0.f};
    if constexpr (BIAS) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (tid + THREADS * i < TR * TR) tabv[i] = table[(size_t)(tid + THREADS * i) * g.heads + head];
    }
    const int qrow = 32 * wave + l31;
    const unsigned qrel = rel_row(qrow) * C3 + qoff + 8 * hh;
    uint4 qraw[2];
    qraw[0] = load16_a4(qkv0 + qrel);
    qraw[1] = load16_a4(qkv0 + (qrel + 16 - 2 * hh));
    {
        const int c = tid & 3;
        const bool last = c == 3;
        uint4 kv[2], vv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned rel = rel_row((tid >> 2) + 128 * i) * C3 + 8 * c - (last ? 2 : 0);
            kv[i] = load16_a4(qkv0 + (rel + koff));
            vv[i] = load16_a4(qkv0 + (rel + voff));
        }
        if constexpr (BIAS) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int it = tid + THREADS * i, di = it / TR, dj = it - di * TR;
                if (it < TR * TR) tb[TMAX - (di * TS + dj)] = tabv[i] * LOG2E;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 2) + 128 * i;
            if (last) {  // columns 30, 31: ones in K (they pick up -r from Q'), V's column 31 the denominator
                kv[i] = make_uint4(kv[i].y, kv[i].z, kv[i].w, 0x3F803F80u);
                vv[i] = make_uint4(vv[i].y, vv[i].z, vv[i].w, 0x3F800000u);
            }
            *reinterpret_cast<uint4 *>(Ki + img<DP>(row, 8 * c)) = kv[i];
            *reinterpret_cast<uint4 *>(Vi + img<DP>(row, 8 * c)) = vv[i];
        }
    }
    // Q' = bf16(q * scale * log2 e): the B operand of S^T = K Q'^T, kept for the whole block
    const float sc2 = scale * LOG2E;
    uint4 qfb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint32_t in[4] = {qraw[s].x, qraw[s].y, qraw[s].z, qraw[s].w};
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = pack_bf16(__uint_as_float(in[i] << 16) * sc2, __uint_as_float(in[i] & 0xffff0000u) * sc2);
        qfb[s] = make_uint4(o[0], o[1], o[2], o[3]);
    }
    if (hh) qfb[1] = make_uint4(qfb[1].y, qfb[1].z, qfb[1].w, 0u);
    // per-lane LDS offsets (elements) inside a 32-row block of an image
    int a_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) a_off[s] = img<DP>(l31, 16 * s + 8 * hh);
    // transposing reads: lane 4q+p of a 16-lane group passes row q, columns 4p..4p+3
    const int trq = (lane >> 2) & 3, trp = lane & 3, trg = (lane >> 4) & 1;
    int t_off[2];  // rows +0 / +8 of a 16-row step (a step further down is 16 whole rows on: the slot pattern repeats)
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) t_off[h8] = img<DP>(4 * hh + trq + 8 * h8, 16 * trg + 4 * trp);
    const int lane_base = TMAX - ((qrow >> 4) * TS + (qrow & 15) + (WS - 1) * (TS + 1) - 4 * hh);
    __syncthreads();

    f32x16 oacc;
    float rref = 0.f;  // the row reference: always the sum of two bf16, so that Q' carries it exactly
    // S^T - r for key tile t: the accumulators start from the bias, the matrix core adds K Q'^T and -r
    // the bias of key tile t against this lane's query: register r of lane half hh is key 32t + rho(r) + 4hh, four
    // consecutive registers are four ascending words of the mirrored table
    auto seeds = [&](int t, f32x16 &sacc) {
        // (the tile's base goes through an empty asm: one address register for the whole pass instead of one per
        // reach of the reads' 8-bit offsets, all of them computed up front and kept alive)
        int tile_base = (int)(tb - reinterpret_cast<float *>(smem)) + lane_base + 2 * TS * t;
        asm volatile("" : "+v"(tile_base));
        const float *tp = reinterpret_cast<const float *>(smem) + tile_base;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (BIAS) sacc[r] = tp[(rho(r) >> 4) * TS + (rho(r) & 15)];
            else sacc[r] = 0.f;
        }
    };
    // S^T - r for key tile t: the accumulators start from the bias, the matrix core adds K Q'^T and -r
    auto products = [&](int t, f32x16 &sacc) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint4 kf = *reinterpret_cast<const uint4 *>(Ki + t * BLK + a_off[s]);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(kf), as_bf16x8(qfb[s]), sacc, 0, 0, 0);
        }
    };
    auto scores = [&](int t, f32x16 &sacc) {
        seeds(t, sacc);
        products(t, sacc);
    };
    auto row_max = [&](const f32x16 &sacc) -> float {
        // (plain fmaxf: an instruction inside inline asm that reads what a matrix instruction has just written is not
        // given the wait states it needs)
        float tm = fmaxf(fmaxf(sacc[0], sacc[1]), sacc[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) tm = fmaxf(fmaxf(tm, sacc[r]), sacc[r + 1]);
        return max_across_halves(fmaxf(tm, sacc[15]));
    };
    // moves the reference to (about) rnew_wish; returns by how much it moved
    auto set_reference = [&](float wish) -> float {
        const uint32_t hl = bf16_hi_lo(wish);
        const float rnew = __uint_as_float(hl << 16) + __uint_as_float(hl & 0xffff0000u), sh = rnew - rref;
        rref = rnew;
        if (hh) qfb[1].w = hl ^ 0x80008000u;  // columns 30, 31 of Q': -hi, -lo
        return sh;
    };
    // P^T = exp2(S^T - r), as it stands the B operand of O^T += V^T P^T
    auto accumulate = [&](int t, f32x16 &sacc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = __builtin_amdgcn_exp2f(sacc[r]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint4 pb = make_uint4(pack_bf16(sacc[8 * s + 0], sacc[8 * s + 1]), pack_bf16(sacc[8 * s + 2], sacc[8 * s + 3]),
                                        pack_bf16(sacc[8 * s + 4], sacc[8 * s + 5]), pack_bf16(sacc[8 * s + 6], sacc[8 * s + 7]));
            // A operand = V^T (row = head dim l31): element j of this lane is key 32t + 16s + 8(j>>2) + 4hh + (j&3)
            const uint2 lo = lds_read_tr16(Vi + t * BLK + 16 * s * DP + t_off[0]), hi = lds_read_tr16(Vi + t * BLK + 16 * s * DP + t_off[1]);
            oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(lo.x, lo.y, hi.x, hi.y)), as_bf16x8(pb), oacc, 0, 0, 0);
        }
    };
    // the denominator is row 31 of O^T: register 15 of the upper lane half, handed to both halves
    auto denominator = [&]() -> float {
        const unsigned si = __float_as_uint(oacc[15]);
        const auto r = __builtin_amdgcn_permlane32_swap(si, si, false, false);
        return __uint_as_float(r[1]);
    };

    // ---- r = row maximum of the first key tile, no maximum after that ----------------------------------------------
    // Exact whatever r is as long as nothing overflows: exp2(S - r) reaches 2^100 only when a later logit exceeds
    // every logit of the first tile by 100 (base 2).  The check on the denominator catches that (one wave-uniform
    // branch, not taken on ordinary logits) and the block is done again with r = the maximum over ALL keys, found by
    // a pass of its own over the score tiles - the same code below runs a second time, nothing is kept alive for it.
    float sum;
    bool again = false;
#pragma unroll 1
    for (;;) {
        float top;
        if (!again) {
            f32x16 sacc;
            scores(0, sacc);
            top = row_max(sacc);
        } else {
            top = -INFINITY;
#pragma unroll 1
            for (int t = 0; t < NT; ++t) {
                f32x16 sacc;
                scores(t, sacc);
                top = fmaxf(top, row_max(sacc));
            }
        }
        // (S - r as it stands now) + r = S: the new reference is the maximum of the logits themselves
        set_reference(rref + top);
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x16 sacc;
            scores(t, sacc);
            accumulate(t, sacc);
        }
        sum = denominator();
        if (again || __builtin_amdgcn_ballot_w64(!(sum < 0x1p100f)) == 0) break;
        again = true;
    }

    // ---- epilogue: normalise, store -----------------------------------------------------------------------------
    const float inv = __builtin_amdgcn_rcpf(sum);  // (1 ulp: the result is rounded to bf16)
    if (hh == 0) lse[((size_t)win * g.heads + head) * L + qrow] = (rref + __builtin_amdgcn_logf(sum)) * LN2;
    // the output tile leaves through this wave's 32 rows of the K image, once no wave reads K or V any more
    __syncthreads();
    store_tile_rows_d30(Ki + wave * BLK, oacc, inv, out + origin * C + head * D, C, pitch, wave, lane);
    if (head == g.heads - 1 && C > g.C) {  // this wave's share of the output rows' padding columns
        // (C and g.C are even: whole dwords; a lane takes every other dword of its token's padding)
        const int np2 = (C - g.C) >> 1;
        T *prow = out + (origin + rel_row(qrow)) * C + g.C;
        for (int j = hh; j < np2; j += 2) *reinterpret_cast<uint32_t *>(prow + 2 * j) = 0u;
    }
}

int launch_fwd_w16(const AttnGeom &g, const void *qkv, const float *table, float scale, void *out, float *lse, hipStream_t st) {
    // head dim 30 (the two last columns of the padded head dim are taken), 4-byte-aligned rows
    if (g.d != 30 || ((g.ldq | g.ldo) & 1)) return -1;
    const bool bias = table != nullptr;
    const dim3 grid(g.n_win * g.heads), block(FwdW16Cfg::THREADS);
    const size_t lds = FwdW16Cfg::lds_bytes(bias);
    if (bias) hipLaunchKernelGGL(win_attn_fwd_w16_kernel<true>, grid, block, lds, st, g, (const bf16_t *)qkv, table, scale, (bf16_t *)out, lse);
    else hipLaunchKernelGGL(win_attn_fwd_w16_kernel<false>, grid, block, lds, st, g, (const bf16_t *)qkv, table, scale, (bf16_t *)out, lse);
    return check_launch("win_attn_fwd_w16_kernel");
}

}  // namespace dsw
