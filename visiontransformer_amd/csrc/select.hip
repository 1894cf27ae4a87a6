// Window selection for DS-MSA: per-image (Gumbel-)softmax, top-k by rank counting, hard gate,
// ordered compaction and window-corner arithmetic, with no host synchronisation.
// Reference: DSwinT_arch.py:201-237 and :280-295.
#include "common.hpp"

namespace dsw {

constexpr int SEL_THREADS = 256;
constexpr int SEL_SPLIT = 4;  // lanes that rank one window together (a power of two)
constexpr int SEL_MAX_WINDOWS = 36864;  // z[] must fit LDS: 36864 * 4 B = 144 KiB

// the reference's per-op rounding in the activation dtype
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// Kernel A.  grid (ceil(n/256), B).  Every block stages the n logits of its image in LDS and
// derives the softmax statistics itself; each thread then ranks one window by counting the
// windows that beat it (higher logit, or equal logit and lower index).  rank < k <=> selected,
// and the rank is the position in the descending top-k list.
// sigmoid of channel 1 + j of sd at a grid cell, rounded to the activation dtype: what
// `sd[:, 1:].sigmoid()` holds (DSwinT_arch.py:269-270, :280); read from `sig` when the caller made it
template <typename T>
__device__ __forceinline__ float gate_value(const T *__restrict__ sd, const T *__restrict__ sig, int b, int j, int n, int cell) {
    if (sig != nullptr) return io<T>::ld(sig + ((size_t)b * 3 + j) * n + cell);
    const float v = io<T>::ld(sd + ((size_t)b * 4 + 1 + j) * n + cell);
    return round_to<T>(1.0f / (1.0f + expf(-v)));
}

template <typename T>
__global__ __launch_bounds__(SEL_THREADS) void select_rank_kernel(
    const T *__restrict__ sd, const T *__restrict__ sig, const float *__restrict__ noise, float tau,
    int n, int k, float *__restrict__ probs, int *__restrict__ cand_idx, float *__restrict__ cand_p,
    int *__restrict__ cand_pass) {
    extern __shared__ __attribute__((aligned(16))) float z[];
    float *red = z + n;
    const int b = blockIdx.y, tid = threadIdx.x;
    const T *scores = sd + (size_t)b * 4 * n;  // channel 0
    float lmax = -INFINITY;
    for (int j = tid; j < n; j += SEL_THREADS) {
        const float s = io<T>::ld(scores + j);
        float v;
        if (noise != nullptr) v = __fdiv_rn(__fadd_rn(s, noise[(size_t)b * n + j]), tau);  // fp32 under autocast
        else v = round_to<T>(__fdiv_rn(s, tau));
        z[j] = v;
        lmax = fmaxf(lmax, v);
    }
    const float zmax = block_max(lmax, red);
    float lsum = 0.f;
    for (int j = tid; j < n; j += SEL_THREADS) lsum += expf(z[j] - zmax);
    const float zsum = block_sum(lsum, red);
    // SEL_SPLIT neighbouring lanes rank one window together, each counting over its share of the windows (a frame
    // has thousands of cells per image and one image: the count is the whole kernel there); the shares are
    // contiguous ranges, summed over the lanes of the group
    const int i = (blockIdx.x * SEL_THREADS + tid) / SEL_SPLIT, part = tid % SEL_SPLIT;
    const int ic = min(i, n - 1);  // lanes past the end count along (the group sum needs them) and drop out below
    const float zi = z[ic];
    const int per = (n + SEL_SPLIT - 1) / SEL_SPLIT, j0 = part * per, j1 = min(j0 + per, n);
    int rank = 0;
    for (int j = j0; j < j1; ++j) {
        const float zj = z[j];
        rank += (zj > zi || (zj == zi && j < ic)) ? 1 : 0;
    }
#pragma unroll
    for (int o = 1; o < SEL_SPLIT; o <<= 1) rank += __shfl_xor(rank, o, 64);
    if (i >= n || part != 0) return;
    const float p = expf(zi - zmax) / zsum;
    probs[(size_t)b * n + i] = p;
    if (rank < k) {
        const size_t o = (size_t)b * k + rank;
        cand_idx[o] = i;
        cand_p[o] = p;
        const float g = gate_value<T>(sd, sig, b, 0, n, i);
        cand_pass[o] = g > 0.5f ? 1 : 0;
    }
}

// Kernel B.  grid B.  Block b counts the passing candidates of the images before it, then
// compacts its own k candidates in rank order and finishes the per-window arithmetic.
template <typename T>
__global__ __launch_bounds__(SEL_THREADS) void select_compact_kernel(
    const T *__restrict__ sd, const T *__restrict__ sig, const int *__restrict__ cand_idx,
    const float *__restrict__ cand_p, const int *__restrict__ cand_pass, int B, int Hw, int Ww, int k, int w, int H, int W,
    int *__restrict__ n_pass, int *__restrict__ win_b, int *__restrict__ win_cell,
    int *__restrict__ win_h1, int *__restrict__ win_w1, float *__restrict__ scores_delta,
    float *__restrict__ gates_delta, int *__restrict__ cellmap) {
    __shared__ int wave_cnt[SEL_THREADS / 64];
    __shared__ int s_base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = Hw * Ww;
    // passing candidates of images 0..b-1
    int c = 0;
    for (int j = tid; j < b * k; j += SEL_THREADS) c += cand_pass[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) wave_cnt[wid] = c;
    for (int j = tid; j < n; j += SEL_THREADS) cellmap[(size_t)b * n + j] = -1;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int i = 0; i < SEL_THREADS / 64; ++i) t += wave_cnt[i];
        s_base = t;
    }
    __syncthreads();
    int running = s_base;
    for (int r0 = 0; r0 < k; r0 += SEL_THREADS) {
        const int r = r0 + tid;
        const int flag = (r < k) ? cand_pass[(size_t)b * k + r] : 0;
        const unsigned long long m = __ballot(flag);
        const int below = __popcll(m & ((1ull << lane) - 1ull));
        __syncthreads();  // wave_cnt reuse
        if (lane == 0) wave_cnt[wid] = __popcll(m);
        __syncthreads();
        int woff = 0, total = 0;
        for (int i = 0; i < SEL_THREADS / 64; ++i) {
            if (i < wid) woff += wave_cnt[i];
            total += wave_cnt[i];
        }
        if (flag) {
            const int id = running + woff + below;
            const int cell = cand_idx[(size_t)b * k + r];
            const int hc = cell / Ww, wc = cell - hc * Ww;
            const float fw = (float)w, half = (float)w * 0.5f;
            const float oh = gate_value<T>(sd, sig, b, 1, n, cell);
            const float ow = gate_value<T>(sd, sig, b, 2, n, cell);
            // offsets*w - w/2: two ops, each rounded in the activation dtype; .long() truncates
            const float th = round_to<T>(__fsub_rn(round_to<T>(__fmul_rn(oh, fw)), half));
            const float tw = round_to<T>(__fsub_rn(round_to<T>(__fmul_rn(ow, fw)), half));
            int h1 = hc * w + (int)th, w1 = wc * w + (int)tw;
            h1 = min(max(h1, 0), H - w);
            w1 = min(max(w1, 0), W - w);
            win_b[id] = b;
            win_cell[id] = cell;
            win_h1[id] = h1;
            win_w1[id] = w1;
            const float g = gate_value<T>(sd, sig, b, 0, n, cell);
            const float p = cand_p[(size_t)b * k + r];
            gates_delta[id] = __fadd_rn(__fsub_rn(1.0f, g), g);   // (hard - g).detach() + g, hard = 1
            scores_delta[id] = __fadd_rn(__fsub_rn(1.0f, p), p);  // (1 - p).detach() + p
            cellmap[(size_t)b * n + cell] = id;
        }
        running += total;
    }
    if (b == B - 1) {
        if (tid == 0) *n_pass = running;
        // the slots behind the passing windows: a harmless window (image 0, corner (0, 0)) with zero straight-through
        // values that the cell map never points to, for callers that run on all B * k slots without reading n_pass
        for (int i = running + tid; i < B * k; i += SEL_THREADS) {
            win_b[i] = 0;
            win_cell[i] = 0;
            win_h1[i] = 0;
            win_w1[i] = 0;
            scores_delta[i] = 0.f;
            gates_delta[i] = 0.f;
        }
    }
}

// Backward: softmax Jacobian on the selected probabilities + gate pass-through.  grid B.
__global__ __launch_bounds__(SEL_THREADS) void select_bwd_kernel(
    const float *__restrict__ probs, const float *__restrict__ d_scores_delta,
    const float *__restrict__ d_gates_delta, const int *__restrict__ win_b,
    const int *__restrict__ win_cell, int n_win, float tau, int n, float *__restrict__ d_scores,
    float *__restrict__ d_gate, const int *__restrict__ n_live) {
    extern __shared__ __attribute__((aligned(16))) float dp[];
    float *red = dp + n;
    if (n_live != nullptr) n_win = min(n_win, *n_live);  // the slots behind the passing windows hold nothing
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < n; j += SEL_THREADS) {
        dp[j] = 0.f;
        d_gate[(size_t)b * n + j] = 0.f;
    }
    __syncthreads();
    for (int t = tid; t < n_win; t += SEL_THREADS) {
        if (win_b[t] == b) {
            const int cell = win_cell[t];
            dp[cell] = d_scores_delta[t];
            d_gate[(size_t)b * n + cell] = d_gates_delta[t];
        }
    }
    __syncthreads();
    float part = 0.f;
    for (int j = tid; j < n; j += SEL_THREADS) part += probs[(size_t)b * n + j] * dp[j];
    const float dot = block_sum(part, red);
    for (int j = tid; j < n; j += SEL_THREADS)
        d_scores[(size_t)b * n + j] = probs[(size_t)b * n + j] * (dp[j] - dot) / tau;
}

// The same backward, continued through `sd[:, 1:].sigmoid()` and written as the gradient of sd
// itself, [B, 4, n] in the activation dtype: channel 0 = d(logits), channel 1 = d(gate) * s (1 - s),
// channels 2, 3 (offsets) = 0 (they reach the output through .long() only, DSwinT_arch.py:232-233).
template <typename T>
__global__ __launch_bounds__(SEL_THREADS) void select_bwd_sd_kernel(
    const float *__restrict__ probs, const float *__restrict__ d_scores_delta,
    const float *__restrict__ d_gates_delta, const int *__restrict__ win_b,
    const int *__restrict__ win_cell, int n_win, float tau, int n, const T *__restrict__ sd, T *__restrict__ d_sd,
    const int *__restrict__ n_live) {
    extern __shared__ __attribute__((aligned(16))) float dp[];  // [n] d(p), [n] d(gate), reduction scratch
    float *dg = dp + n, *red = dg + n;
    if (n_live != nullptr) n_win = min(n_win, *n_live);  // the slots behind the passing windows hold nothing
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < n; j += SEL_THREADS) {
        dp[j] = 0.f;
        dg[j] = 0.f;
    }
    __syncthreads();
    for (int t = tid; t < n_win; t += SEL_THREADS) {
        if (win_b[t] == b) {
            const int cell = win_cell[t];
            dp[cell] = d_scores_delta[t];
            dg[cell] = d_gates_delta[t];
        }
    }
    __syncthreads();
    float part = 0.f;
    for (int j = tid; j < n; j += SEL_THREADS) part += probs[(size_t)b * n + j] * dp[j];
    const float dot = block_sum(part, red);
    const T *sdb = sd + (size_t)b * 4 * n;
    T *out = d_sd + (size_t)b * 4 * n;
    for (int j = tid; j < n; j += SEL_THREADS) {
        io<T>::st(out + j, probs[(size_t)b * n + j] * (dp[j] - dot) / tau);
        const float sg = round_to<T>(1.0f / (1.0f + expf(-io<T>::ld(sdb + n + j))));
        io<T>::st(out + n + j, round_to<T>(dg[j]) * sg * (1.0f - sg));
        io<T>::st(out + 2 * n + j, 0.f);
        io<T>::st(out + 3 * n + j, 0.f);
    }
}

}  // namespace dsw

using namespace dsw;

extern "C" int dsw_select_fwd(const void *sd, const void *sig, int sd_dtype, const float *noise,
                              float tau, int B, int Hw, int Ww, int k, int w, int H, int W,
                              float *probs, int *cand_idx, float *cand_p, int *cand_pass,
                              int *n_pass, int *win_b, int *win_cell, int *win_h1, int *win_w1,
                              float *scores_delta, float *gates_delta, int *cellmap,
                              void *stream) {
    DSW_REQUIRE(sd && probs && cand_idx && cand_p && cand_pass && n_pass && win_b && win_cell &&
                    win_h1 && win_w1 && scores_delta && gates_delta && cellmap,
                "dsw_select_fwd: null pointer");
    DSW_REQUIRE(B > 0 && Hw > 0 && Ww > 0 && w > 0, "dsw_select_fwd: bad sizes B=%d Hw=%d Ww=%d w=%d", B, Hw, Ww, w);
    DSW_REQUIRE(H == Hw * w && W == Ww * w, "dsw_select_fwd: H,W (%d,%d) must equal Hw*w,Ww*w (%d,%d)", H, W, Hw * w, Ww * w);
    const int n = Hw * Ww;
    DSW_REQUIRE(k >= 1 && k <= n, "dsw_select_fwd: k=%d outside [1,%d]", k, n);
    DSW_REQUIRE(tau > 0.f, "dsw_select_fwd: tau must be positive");
    if (n > SEL_MAX_WINDOWS) return fail(DSW_EUNSUPPORTED, "dsw_select_fwd: %d windows per image exceeds %d", n, SEL_MAX_WINDOWS);
    hipStream_t st = as_stream(stream);
    const size_t lds = (size_t)(n + 8) * sizeof(float);
    dim3 gridA((n * SEL_SPLIT + SEL_THREADS - 1) / SEL_THREADS, B);
    DSW_DISPATCH_DTYPE(sd_dtype, T, {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)select_rank_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(DSW_ELAUNCH, "select: LDS attribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(select_rank_kernel<T>, gridA, dim3(SEL_THREADS), lds, st, (const T *)sd, (const T *)sig,
                           noise, tau, n, k, probs, cand_idx, cand_p, cand_pass);
        int rc = check_launch("select_rank_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL(select_compact_kernel<T>, dim3(B), dim3(SEL_THREADS), 0, st, (const T *)sd, (const T *)sig, cand_idx,
                           cand_p, cand_pass, B, Hw, Ww, k, w, H, W, n_pass, win_b, win_cell, win_h1, win_w1,
                           scores_delta, gates_delta, cellmap);
        return check_launch("select_compact_kernel");
    });
    return DSW_OK;
}

extern "C" int dsw_select_bwd(const float *probs, const float *d_scores_delta,
                              const float *d_gates_delta, const int *win_b, const int *win_cell,
                              int n_win, float tau, int B, int Hw, int Ww, float *d_scores,
                              float *d_gate, const int *n_live, void *stream) {
    DSW_REQUIRE(probs && d_scores && d_gate, "dsw_select_bwd: null pointer");
    DSW_REQUIRE(n_win == 0 || (d_scores_delta && d_gates_delta && win_b && win_cell), "dsw_select_bwd: null window arrays");
    DSW_REQUIRE(B > 0 && Hw > 0 && Ww > 0 && n_win >= 0 && tau > 0.f, "dsw_select_bwd: bad sizes");
    const int n = Hw * Ww;
    if (n > SEL_MAX_WINDOWS) return fail(DSW_EUNSUPPORTED, "dsw_select_bwd: %d windows per image exceeds %d", n, SEL_MAX_WINDOWS);
    const size_t lds = (size_t)(n + 8) * sizeof(float);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)select_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DSW_ELAUNCH, "select_bwd: LDS attribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(select_bwd_kernel, dim3(B), dim3(SEL_THREADS), lds, as_stream(stream), probs, d_scores_delta,
                       d_gates_delta, win_b, win_cell, n_win, tau, n, d_scores, d_gate, n_live);
    return check_launch("select_bwd_kernel");
}

extern "C" int dsw_select_bwd_sd(const float *probs, const float *d_scores_delta, const float *d_gates_delta,
                                 const int *win_b, const int *win_cell, int n_win, float tau, int B, int Hw, int Ww,
                                 const void *sd, int sd_dtype, void *d_sd, const int *n_live, void *stream) {
    DSW_REQUIRE(probs && sd && d_sd, "dsw_select_bwd_sd: null pointer");
    DSW_REQUIRE(n_win == 0 || (d_scores_delta && d_gates_delta && win_b && win_cell), "dsw_select_bwd_sd: null window arrays");
    DSW_REQUIRE(B > 0 && Hw > 0 && Ww > 0 && n_win >= 0 && tau > 0.f, "dsw_select_bwd_sd: bad sizes");
    const int n = Hw * Ww;
    if (n > SEL_MAX_WINDOWS) return fail(DSW_EUNSUPPORTED, "dsw_select_bwd_sd: %d windows per image exceeds %d", n, SEL_MAX_WINDOWS);
    const size_t lds = (size_t)(2 * n + 8) * sizeof(float);
    DSW_DISPATCH_DTYPE(sd_dtype, T, {
        auto kern = select_bwd_sd_kernel<T>;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(DSW_ELAUNCH, "select_bwd_sd: LDS attribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(kern, dim3(B), dim3(SEL_THREADS), lds, as_stream(stream), probs, d_scores_delta, d_gates_delta,
                           win_b, win_cell, n_win, tau, n, (const T *)sd, (T *)d_sd, n_live);
    });
    return check_launch("select_bwd_sd_kernel");
}
