// Bilinear window gather (roi_align specialised to integer-aligned w x w windows) and the
// gated scatter-mean write-back of DS-MSA, forward and backward, NHWC, atomic-free.
// Reference: DSwinT_arch.py:239-253, :259, :295-297, :303, :308-315.
//
// Geometry both directions rely on: a window that starts from grid cell (r, c) has its corner
// h1 in [r*w - w/2, r*w + w/2] (the offset is trunc(sigmoid*w - w/2), and the clamp to [0, H-w]
// only moves border windows inwards), so pixel row y can only be covered by windows of cell rows
// floor((y + w/2)/w) - 1 and floor((y + w/2)/w); the same holds per column.  With one window per
// cell at most, a pixel has <= 4 covering windows, found through cellmap.
#include "common.hpp"

namespace dsw {

constexpr int GS_THREADS = 256;
constexpr int TILE_PIX = 32;  // pixels of one image row handled by a block

struct Cover {
    int id[4];   // covering window ids, ascending
    int tok[4];  // token index inside that window: (y - h1) * w + (x - w1)
    int n;
};

// The covering windows of pixel (y, x), ascending by window id.  Both rounds of reads are issued as a group:
// the four candidate cells first (cells outside the grid count as empty), then - only if one of them holds a
// window - the corners of the four candidates (an empty candidate reads window 0, which exists then, and is
// dropped), so a lookup costs two memory latencies whatever it finds.  SG: also the product
// scores_delta * gates_delta of each covering window, in the same order.
template <bool SG>
__device__ __forceinline__ Cover find_cover_sg(const int *__restrict__ cellmap, const int *__restrict__ win_h1,
                                               const int *__restrict__ win_w1, const float *__restrict__ scores_delta,
                                               const float *__restrict__ gates_delta, int b, int y, int x, int Hw, int Ww,
                                               int w, float (&sg)[4]) {
    Cover cv;
    cv.n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { cv.id[i] = 0x7fffffff; cv.tok[i] = 0; sg[i] = 0.f; }
    const int r_hi = min((y + (w >> 1)) / w, Hw - 1);
    const int c_hi = min((x + (w >> 1)) / w, Ww - 1);
    int ids[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r_hi - 1 + (q >> 1), c = c_hi - 1 + (q & 1);
        const int v = cellmap[((size_t)b * Hw + max(r, 0)) * Ww + max(c, 0)];
        ids[q] = (r >= 0 && c >= 0) ? v : -1;
    }
    if ((ids[0] & ids[1] & ids[2] & ids[3]) < 0) return cv;  // every candidate is empty
    int h1[4], w1[4];
    float prod[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = max(ids[q], 0);
        h1[q] = win_h1[j];
        w1[q] = win_w1[j];
        if constexpr (SG) prod[q] = scores_delta[j] * gates_delta[j];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int id = ids[q];
        const int ly = y - h1[q], lx = x - w1[q];
        if (id < 0 || ly < 0 || ly >= w || lx < 0 || lx >= w) continue;
        // insertion into the ascending list (<= 4 entries; free slots hold the largest id): one pass of
        // compare-and-exchange with fixed indices, so the list stays in registers
        int nid = id, ntok = ly * w + lx;
        float nsg = SG ? prod[q] : 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool sw = nid < cv.id[s];
            const int tid_ = cv.id[s], ttok = cv.tok[s];
            cv.id[s] = sw ? nid : tid_;
            cv.tok[s] = sw ? ntok : ttok;
            nid = sw ? tid_ : nid;
            ntok = sw ? ttok : ntok;
            if constexpr (SG) {
                const float tsg = sg[s];
                sg[s] = sw ? nsg : tsg;
                nsg = sw ? tsg : nsg;
            }
        }
        cv.n += 1;
    }
    return cv;
}
__device__ __forceinline__ Cover find_cover(const int *__restrict__ cellmap, const int *__restrict__ win_h1,
                                            const int *__restrict__ win_w1, int b, int y, int x, int Hw,
                                            int Ww, int w) {
    float sg[4];
    return find_cover_sg<false>(cellmap, win_h1, win_w1, nullptr, nullptr, b, y, x, Hw, Ww, w, sg);
}

// torchvision's bilinear sample at (y + .5, x + .5): taps (y, x), (y, xh), (yh, x), (yh, xh)
// with weights hy*hx, hy*lx, ly*hx, ly*lx; on the last row/column the high tap folds onto the
// low one with weight 0.
struct Taps {
    int yh, xh;
    float w00, w01, w10, w11;
};
__device__ __forceinline__ Taps make_taps(int y, int x, int H, int W) {
    Taps t;
    const float ly = (y < H - 1) ? 0.5f : 0.f, lx = (x < W - 1) ? 0.5f : 0.f;
    const float hy = 1.f - ly, hx = 1.f - lx;
    t.yh = (y < H - 1) ? y + 1 : y;
    t.xh = (x < W - 1) ? x + 1 : x;
    t.w00 = hy * hx;
    t.w01 = hy * lx;
    t.w10 = ly * hx;
    t.w11 = ly * lx;
    return t;
}

template <typename T>
__device__ __forceinline__ float4 box_sample(const T *__restrict__ img, int y, int x, int H, int W, int C, int c) {
    const Taps t = make_taps(y, x, H, W);
    const float4 v00 = io<T>::ld4(img + ((size_t)y * W + x) * C + c);
    const float4 v01 = io<T>::ld4(img + ((size_t)y * W + t.xh) * C + c);
    const float4 v10 = io<T>::ld4(img + ((size_t)t.yh * W + x) * C + c);
    const float4 v11 = io<T>::ld4(img + ((size_t)t.yh * W + t.xh) * C + c);
    return ((v00 * t.w00 + v01 * t.w01) + v10 * t.w10) + v11 * t.w11;
}

// ---- gather forward: one block per (window, window row) -------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(GS_THREADS) void gather_fwd_kernel(
    const TI *__restrict__ x, const int *__restrict__ win_b, const int *__restrict__ win_h1,
    const int *__restrict__ win_w1, int H, int W, int C, int w, int ldo, TO *__restrict__ out) {
    // out rows are ldo >= C elements apart; the columns C..ldo-1 are set to zero
    const int bid = xcd_remap(blockIdx.x, gridDim.x);  // the rows of a window on one XCD: they share x rows
    const int n = __builtin_amdgcn_readfirstlane(bid / w), i = bid - n * w;
    const int b = win_b[n], y = win_h1[n] + i, x0 = win_w1[n];
    const TI *img = x + (size_t)b * H * W * C;
    TO *dst = out + ((size_t)n * w * w + (size_t)i * w) * ldo;
    const int CV = C >> 2, OV = ldo >> 2;
    for (int it = threadIdx.x; it < w * OV; it += GS_THREADS) {
        const int j = it / OV, c = (it - j * OV) << 2;
        io<TO>::st4(dst + (size_t)j * ldo + c, c < C ? box_sample<TI>(img, y, x0 + j, H, W, C, c) : f4zero());
    }
    (void)CV;
}

// ---- gather backward: one block per TILE_PIX pixels of an image row -------------------------------
// dx(y,x) = sum over the sample positions (y',x') in {y-1,y} x {x-1,x} whose taps include (y,x) of
//           weight * sum over windows covering (y',x') of dout[window, token(y',x')].
// ACC: add into an existing dx (pixels no window touches are left alone) instead of writing all of it.
//
// The windows that can reach the tile are few (they start from the <= 3 x 4 grid cells around it at w = 16) and
// the same for every thread of the block: they are looked up once (one thread per candidate cell, both rounds
// of reads as groups), sorted by id into a list in LDS, and the streaming phase walks that list - a uniform loop -
// with each thread testing its own four sample positions against the window's corner by arithmetic.  A tile no
// window reaches is a plain zero fill (or nothing at all when accumulating).  The terms of a pixel are added
// window by window in ascending id, sample positions (row, column) in order within a window.
constexpr int GB_TILE = 32;   // pixels of one image row handled by a block of the gather backward
constexpr int GB_MAXC = 128;  // candidate cells of a tile: <= 3 rows x (GB_TILE / w + 2) columns, w >= 1
template <typename TG, typename TO, bool ACC>
__global__ __launch_bounds__(GS_THREADS) void gather_bwd_kernel(
    const TG *__restrict__ dout, const int *__restrict__ cellmap, const int *__restrict__ win_h1,
    const int *__restrict__ win_w1, int H, int W, int C, int w, int ldg, int tiles_per_row, TO *__restrict__ dx) {
    __shared__ int cand_id[GB_MAXC];
    __shared__ int wl_id[GB_MAXC], wl_h1[GB_MAXC], wl_w1[GB_MAXC];
    const int Hw = H / w, Ww = W / w;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);  // an XCD takes whole images: vertical neighbours share its L2
    // (wave-uniform quotients of the vector unit's division, handed back to scalar registers)
    const int trow = __builtin_amdgcn_readfirstlane(bid / tiles_per_row), tile = bid - trow * tiles_per_row;
    const int b = __builtin_amdgcn_readfirstlane(trow / H), y = trow - b * H;
    const int x0 = tile * GB_TILE;
    const int npix = min(GB_TILE, W - x0);
    // candidate cells: a window that covers sample position (yy, xx) starts from cell row r_hi(yy) - 1 or r_hi(yy),
    // r_hi(v) = min((v + w/2) / w, Hw - 1), and likewise per column
    const int hw2 = w >> 1;
    const int rmin = max(min((max(y - 1, 0) + hw2) / w, Hw - 1) - 1, 0), rmax = min((y + hw2) / w, Hw - 1);
    const int cmin = max(min((max(x0 - 1, 0) + hw2) / w, Ww - 1) - 1, 0), cmax = min((x0 + npix - 1 + hw2) / w, Ww - 1);
    const int ncols = cmax - cmin + 1, ncell = (rmax - rmin + 1) * ncols;  // <= GB_MAXC
    int my_id = 0x7fffffff, my_h1 = 0, my_w1 = 0;
    if ((int)threadIdx.x < ncell) {
        const int r = rmin + (int)threadIdx.x / ncols, c = cmin + (int)threadIdx.x % ncols;
        const int id = cellmap[((size_t)b * Hw + r) * Ww + c];
        if (id >= 0) {
            const int h1 = win_h1[id], w1 = win_w1[id];
            // reaches the tile: a sample row among {y-1, y} and a sample column among x0-1 .. x0+npix-1 inside it
            if (h1 <= y && y <= h1 + w && w1 <= x0 + npix - 1 && x0 - 1 < w1 + w) {
                my_id = id;
                my_h1 = h1;
                my_w1 = w1;
            }
        }
    }
    if (threadIdx.x < GB_MAXC) cand_id[threadIdx.x] = my_id;
    const int nwin = __syncthreads_count(my_id != 0x7fffffff);
    const int CV = C >> 2, L = w * w, total = npix * CV;
    TO *drow = dx + (((size_t)b * H + y) * W + x0) * C;
    if (nwin == 0) {  // uniform
        if constexpr (!ACC) {
            for (int i = threadIdx.x; i < total; i += GS_THREADS) io<TO>::st4(drow + (size_t)i * 4, f4zero());
        }
        return;
    }
    if (my_id != 0x7fffffff) {  // position in the ascending list = number of smaller ids (ids are distinct)
        int rank = 0;
        for (int u = 0; u < ncell; ++u) rank += cand_id[u] < my_id;
        wl_id[rank] = my_id;
        wl_h1[rank] = my_h1;
        wl_w1[rank] = my_w1;
    }
    __syncthreads();
    const float wy_self = (y < H - 1) ? 0.5f : 1.0f;
    for (int it = threadIdx.x; it < total; it += GS_THREADS) {
        const int p = it / CV, c = (it - p * CV) << 2;
        const int x = x0 + p;
        const float wx_self = (x < W - 1) ? 0.5f : 1.0f;
        float4 acc = f4zero();
        bool touched = false;
        float4 old = f4zero();
        if constexpr (ACC) {
            // whether any listed window holds one of this pixel's sample positions: decides the read of dx up front
            for (int k = 0; k < nwin; ++k) {
                const int ly = y - wl_h1[k], lx = x - wl_w1[k];
                touched |= ly >= 0 && ly <= w && lx >= 0 && lx <= w;
            }
            if (touched) old = io<TO>::ld4(drow + (size_t)p * C + c);
        }
        for (int k = 0; k < nwin; ++k) {
            const int ly1 = y - wl_h1[k], lx1 = x - wl_w1[k];  // sample position (y, x) inside the window; (y-1, x-1) is one less
            if (ly1 < 0 || ly1 > w || lx1 < 0 || lx1 > w) continue;
            const TG *wbase = dout + (size_t)wl_id[k] * L * ldg + c;
            float4 v[4];
            bool ok[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ly = ly1 - 1 + (q >> 1), lx = lx1 - 1 + (q & 1);
                ok[q] = ly >= 0 && ly < w && lx >= 0 && lx < w;
                v[q] = io<TG>::ld4(wbase + (size_t)(ok[q] ? ly * w + lx : 0) * ldg);  // token 0 of the window when outside: not added
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float wt = ((q >> 1) ? wy_self : 0.5f) * ((q & 1) ? wx_self : 0.5f);
                if (ok[q]) acc = acc + v[q] * wt;
            }
        }
        TO *dst = drow + (size_t)p * C + c;
        if constexpr (ACC) {
            if (touched) io<TO>::st4(dst, old + acc);
        } else {
            io<TO>::st4(dst, acc);
        }
    }
}

// ---- scatter-mean + residual forward ----------------------------------------------------------------
template <typename TX, typename TA, typename TY>
__global__ __launch_bounds__(GS_THREADS) void scatter_fwd_kernel(
    const TX *__restrict__ x, const TA *__restrict__ after, const float *__restrict__ gates_delta,
    const float *__restrict__ scores_delta, const int *__restrict__ cellmap, const int *__restrict__ win_h1,
    const int *__restrict__ win_w1, int H, int W, int C, int w, int lda, int tiles_per_row, TY *__restrict__ yout,
    const int *__restrict__ skip_if_zero) {
    // skip_if_zero: a device counter (the number of windows that passed); when it holds 0 nothing is written - the
    // caller's next kernel reads x itself in that case (dsw_dwconv3x3_fwd_src), and the pass over the map is saved
    if (skip_if_zero != nullptr && *skip_if_zero == 0) return;
    __shared__ Cover cov[TILE_PIX];
    const int Hw = H / w, Ww = W / w;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);  // an XCD takes whole images: vertical neighbours share its L2
    // (wave-uniform quotients of the vector unit's division, handed back to scalar registers)
    const int trow = __builtin_amdgcn_readfirstlane(bid / tiles_per_row), tile = bid - trow * tiles_per_row;
    const int b = __builtin_amdgcn_readfirstlane(trow / H), y = trow - b * H;
    const int x0 = tile * TILE_PIX;
    const int npix = min(TILE_PIX, W - x0);
    if (threadIdx.x < npix) cov[threadIdx.x] = find_cover(cellmap, win_h1, win_w1, b, y, x0 + threadIdx.x, Hw, Ww, w);
    __syncthreads();
    const int CV = C >> 2, L = w * w;
    const TX *img = x + (size_t)b * H * W * C;
    for (int it = threadIdx.x; it < npix * CV; it += GS_THREADS) {
        const int p = it / CV, c = (it - p * CV) << 2;
        const int xx = x0 + p;
        const Cover &cv = cov[p];
        const size_t off = ((size_t)y * W + xx) * C + c;
        float4 v = io<TX>::ld4(img + off);
        if (cv.n > 0) {
            const float4 before = box_sample<TX>(img, y, xx, H, W, C, c);
            float4 acc = f4zero();
            for (int s = 0; s < cv.n; ++s) {
                const int id = cv.id[s];
                const float4 a = io<TA>::ld4(after + ((size_t)id * L + cv.tok[s]) * lda + c);
                acc = acc + ((a - before) * gates_delta[id]) * scores_delta[id];
            }
            v = v + acc / (float)cv.n;
        }
        io<TY>::st4(yout + (size_t)b * H * W * C + off, v);
    }
}

// ---- scatter backward, pixel side: dx = dy - boxT( dy/count * sum_n(s_n*g_n) ) --------------------
template <typename TG, typename TO, bool ACC>
__global__ __launch_bounds__(GS_THREADS) void scatter_bwd_dx_kernel(
    const TG *__restrict__ dy, const float *__restrict__ gates_delta, const float *__restrict__ scores_delta,
    const int *__restrict__ cellmap, const int *__restrict__ win_h1, const int *__restrict__ win_w1, int H,
    int W, int C, int w, int tiles_per_row, TO *dx, const TO *acc_src) {
    // (ACC: the result is acc_src + the gradient; acc_src is dx itself for accumulation in place, or another map of
    // the same shape - a gradient that reaches the same tensor by another route - which is then left untouched)
    __shared__ float fac[2][TILE_PIX + 1];  // sum_n(s_n*g_n)/count at the sample position, 0 if uncovered
    const int Hw = H / w, Ww = W / w;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);  // an XCD takes whole images: vertical neighbours share its L2
    // (wave-uniform quotients of the vector unit's division, handed back to scalar registers)
    const int trow = __builtin_amdgcn_readfirstlane(bid / tiles_per_row), tile = bid - trow * tiles_per_row;
    const int b = __builtin_amdgcn_readfirstlane(trow / H), y = trow - b * H;
    const int x0 = tile * TILE_PIX;
    const int npix = min(TILE_PIX, W - x0);
    const int CV = C >> 2, total = npix * CV;
    const float wy_self = (y < H - 1) ? 0.5f : 1.0f;
    const TG *gimg = dy + (size_t)b * H * W * C;
    const TG *grow = gimg + (size_t)y * W * C;                     // image row y
    const TG *gup = gimg + (size_t)(y > 0 ? y - 1 : 0) * W * C;    // row y - 1 (its factors are 0 when y == 0)
    TO *drow = dx + (((size_t)b * H + y) * W) * C;
    const TO *arow = ACC ? acc_src + (((size_t)b * H + y) * W) * C : nullptr;
    // A thread takes SB_U items (pixel, 4 channels) per trip.  Their own dy values (and the destination when
    // accumulating) are asked for together one trip ahead - the first trip's before the covering windows are
    // looked up, so that the lookup's two memory latencies pass under them - and an item with a covered sample
    // position among (y-1, x-1), (y-1, x), (y, x-1) asks for those three together (clamped addresses, factor 0
    // where there is nothing).  The fourth sample position is the pixel itself.  The terms are subtracted in a
    // fixed order with the factors (weight * f); a term whose f is 0 leaves the value as it is.
    constexpr int SB_U = 3;
    struct Trip {
        float4 ctr[SB_U], old[SB_U];
        int p[SB_U], c[SB_U];
    };
    auto issue = [&](int base, Trip &t) {
#pragma unroll
        for (int u = 0; u < SB_U; ++u) {
            const int it = min(base + u * GS_THREADS, total - 1);  // past the end: a valid item, dropped later
            const int p = it / CV, c = (it - p * CV) << 2;
            t.p[u] = p;
            t.c[u] = c;
            t.ctr[u] = io<TG>::ld4(grow + (size_t)(x0 + p) * C + c);
            if constexpr (ACC) t.old[u] = io<TO>::ld4(arow + (size_t)(x0 + p) * C + c);
        }
    };
    Trip cur;
    issue(threadIdx.x, cur);
    for (int t = threadIdx.x; t < 2 * (TILE_PIX + 1); t += GS_THREADS) {
        const int row = t / (TILE_PIX + 1), col = t - row * (TILE_PIX + 1);
        const int yy = y - 1 + row, xx = x0 - 1 + col;
        float f = 0.f;
        if (yy >= 0 && xx >= 0 && xx < W) {
            float sg[4];
            const Cover cv = find_cover_sg<true>(cellmap, win_h1, win_w1, scores_delta, gates_delta, b, yy, xx, Hw, Ww, w, sg);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (s < cv.n) f += sg[s];
            if (cv.n > 0) f = f / (float)cv.n;
        }
        fac[row][col] = f;
    }
    __syncthreads();
    for (int base = threadIdx.x; base < total; base += SB_U * GS_THREADS) {
        Trip nxt;
        issue(base + SB_U * GS_THREADS, nxt);
#pragma unroll
        for (int u = 0; u < SB_U; ++u) {
            if (base + u * GS_THREADS >= total) break;
            const int p = cur.p[u], c = cur.c[u], x = x0 + p;
            const float wx_self = (x < W - 1) ? 0.5f : 1.0f;
            const float f00 = fac[0][p], f01 = fac[0][p + 1], f10 = fac[1][p], f11 = fac[1][p + 1];
            float4 acc = cur.ctr[u];
            if (f00 != 0.f || f01 != 0.f || f10 != 0.f) {
                const int xm = x > 0 ? x - 1 : 0;
                const float4 n00 = io<TG>::ld4(gup + (size_t)xm * C + c);
                const float4 n01 = io<TG>::ld4(gup + (size_t)x * C + c);
                const float4 n10 = io<TG>::ld4(grow + (size_t)xm * C + c);
                if (f00 != 0.f) acc = acc - n00 * ((0.5f * 0.5f) * f00);
                if (f01 != 0.f) acc = acc - n01 * ((0.5f * wx_self) * f01);
                if (f10 != 0.f) acc = acc - n10 * ((wy_self * 0.5f) * f10);
            }
            if (f11 != 0.f) acc = acc - cur.ctr[u] * ((wy_self * wx_self) * f11);
            if constexpr (ACC) acc = acc + cur.old[u];
            io<TO>::st4(drow + (size_t)x * C + c, acc);
        }
        cur = nxt;
    }
}

// ---- scatter backward, window side: d_after and per-row partials of d_gates / d_scores -----------
// one block per (window, window row)
template <typename TG, typename TX, typename TA, typename TO>
__global__ __launch_bounds__(GS_THREADS) void scatter_bwd_win_kernel(
    const TG *__restrict__ dy, const TX *__restrict__ x, const TA *__restrict__ after,
    const float *__restrict__ gates_delta, const float *__restrict__ scores_delta,
    const int *__restrict__ cellmap, const int *__restrict__ win_b, const int *__restrict__ win_h1,
    const int *__restrict__ win_w1, int H, int W, int C, int w, int lda, int ldd, TO *__restrict__ d_after,
    float *__restrict__ partial, const int *__restrict__ n_live) {
    __shared__ float inv_cnt[64];
    __shared__ float red[GS_THREADS / 64];
    const int Hw = H / w, Ww = W / w;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n = __builtin_amdgcn_readfirstlane(bid / w), i = bid - n * w;
    if (n_live != nullptr && n >= *n_live) {
        // a window slot behind the passing ones (callers that never read the count on the host hand over all
        // B*k slots): nothing was written back from it, so its gradients are zero
        for (int it = threadIdx.x; it < w * (ldd >> 2); it += GS_THREADS)
            io<TO>::st4(d_after + ((size_t)n * w * w + (size_t)i * w) * ldd + ((size_t)it << 2), f4zero());
        if (threadIdx.x == 0) {
            partial[((size_t)n * w + i) * 2 + 0] = 0.f;
            partial[((size_t)n * w + i) * 2 + 1] = 0.f;
        }
        return;
    }
    const int b = win_b[n], y = win_h1[n] + i, x0 = win_w1[n];
    if (threadIdx.x < w) {
        const Cover cv = find_cover(cellmap, win_h1, win_w1, b, y, x0 + threadIdx.x, Hw, Ww, w);
        inv_cnt[threadIdx.x] = (float)cv.n;  // >= 1: this window covers the pixel
    }
    __syncthreads();
    const float g = gates_delta[n], s = scores_delta[n];
    const TX *img = x + (size_t)b * H * W * C;
    const TG *gimg = dy + (size_t)b * H * W * C;
    const size_t tok0 = ((size_t)n * w * w + (size_t)i * w);
    const int CV = C >> 2;
    float pg = 0.f, ps = 0.f;
    for (int it = threadIdx.x; it < w * CV; it += GS_THREADS) {
        const int j = it / CV, c = (it - j * CV) << 2;
        const float4 dgated = io<TG>::ld4(gimg + ((size_t)y * W + x0 + j) * C + c) / inv_cnt[j];
        const float4 u = dgated * s;  // d(delta*g)
        io<TO>::st4(d_after + (tok0 + j) * ldd + c, u * g);
        const float4 before = box_sample<TX>(img, y, x0 + j, H, W, C, c);
        const float4 delta = io<TA>::ld4(after + (tok0 + j) * lda + c) - before;
        pg += (u.x * delta.x + u.y * delta.y) + (u.z * delta.z + u.w * delta.w);
        const float4 dg = delta * g;
        ps += (dgated.x * dg.x + dgated.y * dg.y) + (dgated.z * dg.z + dgated.w * dg.w);
    }
    for (int it = threadIdx.x; it < w * ((ldd - C) >> 2); it += GS_THREADS) {  // padding columns of d_after
        const int PV = (ldd - C) >> 2, j = it / PV;
        io<TO>::st4(d_after + (tok0 + j) * ldd + C + ((it - j * PV) << 2), f4zero());
    }
    const float tg = block_sum(pg, red);
    const float ts = block_sum(ps, red);
    if (threadIdx.x == 0) {
        partial[((size_t)n * w + i) * 2 + 0] = tg;
        partial[((size_t)n * w + i) * 2 + 1] = ts;
    }
}

__global__ void scatter_bwd_reduce_kernel(const float *__restrict__ partial, int n_win, int w,
                                          float *__restrict__ d_gates_delta, float *__restrict__ d_scores_delta) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_win) return;
    // row order, eight rows' partials asked for at a time
    const float2 *row = reinterpret_cast<const float2 *>(partial) + (size_t)n * w;
    float tg = 0.f, ts = 0.f;
    int i = 0;
    for (; i + 8 <= w; i += 8) {
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = row[i + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            tg += v[j].x;
            ts += v[j].y;
        }
    }
    for (; i < w; ++i) {
        tg += row[i].x;
        ts += row[i].y;
    }
    d_gates_delta[n] = tg;
    d_scores_delta[n] = ts;
}

inline int check_geometry(const char *who, int B, int H, int W, int C, int w) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || w <= 0) return fail(DSW_EINVAL, "%s: bad sizes B=%d H=%d W=%d C=%d w=%d", who, B, H, W, C, w);
    if (H % w || W % w) return fail(DSW_EINVAL, "%s: H,W (%d,%d) must be multiples of the window size %d", who, H, W, w);
    if (C % 4) return fail(DSW_EUNSUPPORTED, "%s: channel count %d must be a multiple of 4", who, C);
    if (w > 64) return fail(DSW_EUNSUPPORTED, "%s: window size %d > 64", who, w);
    return DSW_OK;
}

// row stride of a window-token matrix: a multiple of 4 elements, at least C
inline int check_ld(const char *who, int ld, int C) {
    if (ld < C || ld % 4) return fail(DSW_EINVAL, "%s: row stride %d must be a multiple of 4 and >= %d", who, ld, C);
    return DSW_OK;
}

}  // namespace dsw

using namespace dsw;

extern "C" int dsw_gather_fwd(const void *x, int x_dtype, const int *win_b, const int *win_h1,
                              const int *win_w1, int n_win, int B, int H, int W, int C, int w,
                              void *out, int out_dtype, int ld_out, void *stream) {
    int rc = check_geometry("dsw_gather_fwd", B, H, W, C, w);
    if (rc) return rc;
    if (!ld_out) ld_out = C;
    rc = check_ld("dsw_gather_fwd", ld_out, C);
    if (rc) return rc;
    DSW_REQUIRE(n_win >= 0, "dsw_gather_fwd: n_win < 0");
    if (n_win == 0) return DSW_OK;
    DSW_REQUIRE(x && win_b && win_h1 && win_w1 && out, "dsw_gather_fwd: null pointer");
    DSW_DISPATCH_DTYPE(x_dtype, TI, DSW_DISPATCH_DTYPE(out_dtype, TO, {
        hipLaunchKernelGGL((gather_fwd_kernel<TI, TO>), dim3(n_win * w), dim3(GS_THREADS), 0, as_stream(stream),
                           (const TI *)x, win_b, win_h1, win_w1, H, W, C, w, ld_out, (TO *)out);
    }));
    return check_launch("gather_fwd_kernel");
}

extern "C" int dsw_gather_bwd(const void *dout, int dout_dtype, const int *cellmap, const int *win_h1,
                              const int *win_w1, int n_win, int B, int H, int W, int C, int w,
                              int ld_dout, void *dx, int dx_dtype, int accumulate, void *stream) {
    int rc = check_geometry("dsw_gather_bwd", B, H, W, C, w);
    if (rc) return rc;
    if (!ld_dout) ld_dout = C;
    rc = check_ld("dsw_gather_bwd", ld_dout, C);
    if (rc) return rc;
    DSW_REQUIRE(cellmap && dx && (n_win == 0 || (dout && win_h1 && win_w1)), "dsw_gather_bwd: null pointer");
    const int tiles = (W + GB_TILE - 1) / GB_TILE;
    DSW_DISPATCH_DTYPE(dout_dtype, TG, DSW_DISPATCH_DTYPE(dx_dtype, TO, {
        if (accumulate)
            hipLaunchKernelGGL((gather_bwd_kernel<TG, TO, true>), dim3(B * H * tiles), dim3(GS_THREADS), 0, as_stream(stream),
                               (const TG *)dout, cellmap, win_h1, win_w1, H, W, C, w, ld_dout, tiles, (TO *)dx);
        else
            hipLaunchKernelGGL((gather_bwd_kernel<TG, TO, false>), dim3(B * H * tiles), dim3(GS_THREADS), 0, as_stream(stream),
                               (const TG *)dout, cellmap, win_h1, win_w1, H, W, C, w, ld_dout, tiles, (TO *)dx);
    }));
    return check_launch("gather_bwd_kernel");
}

static int scatter_mean_fwd_impl(const void *x, int x_dtype, const void *after, int after_dtype, const float *gates_delta,
                                 const float *scores_delta, const int *cellmap, const int *win_h1, const int *win_w1, int n_win,
                                 int B, int H, int W, int C, int w, int ld_after, void *y, int y_dtype, const int *skip_if_zero,
                                 void *stream) {
    int rc = check_geometry("dsw_scatter_mean_fwd", B, H, W, C, w);
    if (rc) return rc;
    if (!ld_after) ld_after = C;
    rc = check_ld("dsw_scatter_mean_fwd", ld_after, C);
    if (rc) return rc;
    DSW_REQUIRE(x && y && cellmap && (n_win == 0 || (after && gates_delta && scores_delta && win_h1 && win_w1)),
                "dsw_scatter_mean_fwd: null pointer");
    const int tiles = (W + TILE_PIX - 1) / TILE_PIX;
    DSW_DISPATCH_DTYPE(x_dtype, TX, DSW_DISPATCH_DTYPE(after_dtype, TA, DSW_DISPATCH_DTYPE(y_dtype, TY, {
        hipLaunchKernelGGL((scatter_fwd_kernel<TX, TA, TY>), dim3(B * H * tiles), dim3(GS_THREADS), 0,
                           as_stream(stream), (const TX *)x, (const TA *)after, gates_delta, scores_delta, cellmap,
                           win_h1, win_w1, H, W, C, w, ld_after, tiles, (TY *)y, skip_if_zero);
    })));
    return check_launch("scatter_fwd_kernel");
}

extern "C" int dsw_scatter_mean_fwd(const void *x, int x_dtype, const void *after, int after_dtype,
                                    const float *gates_delta, const float *scores_delta,
                                    const int *cellmap, const int *win_h1, const int *win_w1, int n_win,
                                    int B, int H, int W, int C, int w, int ld_after, void *y, int y_dtype,
                                    void *stream) {
    return scatter_mean_fwd_impl(x, x_dtype, after, after_dtype, gates_delta, scores_delta, cellmap, win_h1, win_w1, n_win, B, H, W, C, w,
                                 ld_after, y, y_dtype, nullptr, stream);
}

extern "C" int dsw_scatter_mean_fwd_live(const void *x, int x_dtype, const void *after, int after_dtype, const float *gates_delta,
                                         const float *scores_delta, const int *cellmap, const int *win_h1, const int *win_w1,
                                         int n_win, int B, int H, int W, int C, int w, int ld_after, void *y, int y_dtype,
                                         const int *n_live, void *stream) {
    return scatter_mean_fwd_impl(x, x_dtype, after, after_dtype, gates_delta, scores_delta, cellmap, win_h1, win_w1, n_win, B, H, W, C, w,
                                 ld_after, y, y_dtype, n_live, stream);
}

static int scatter_mean_bwd_impl(const void *dy, int dy_dtype, const void *x, int x_dtype,
                                 const void *after, int after_dtype, const float *gates_delta,
                                 const float *scores_delta, const int *cellmap, const int *win_b,
                                 const int *win_h1, const int *win_w1, int n_win, int B, int H, int W,
                                 int C, int w, int ld_after, void *dx, int dx_dtype, int accumulate_dx, const void *dx_add,
                                 void *d_after, int d_after_dtype, int ld_d_after, float *d_gates_delta,
                                 float *d_scores_delta, float *partial, const int *n_live, void *stream) {
    int rc = check_geometry("dsw_scatter_mean_bwd", B, H, W, C, w);
    if (rc) return rc;
    if (!ld_after) ld_after = C;
    if (!ld_d_after) ld_d_after = C;
    rc = check_ld("dsw_scatter_mean_bwd", ld_after, C);
    if (rc) return rc;
    rc = check_ld("dsw_scatter_mean_bwd", ld_d_after, C);
    if (rc) return rc;
    DSW_REQUIRE(dy && dx && cellmap, "dsw_scatter_mean_bwd: null pointer");
    DSW_REQUIRE(n_win == 0 || (x && after && gates_delta && scores_delta && win_b && win_h1 && win_w1 && d_after &&
                               d_gates_delta && d_scores_delta && partial),
                "dsw_scatter_mean_bwd: null window arrays");
    DSW_REQUIRE((reinterpret_cast<uintptr_t>(partial) & 7) == 0, "dsw_scatter_mean_bwd: partial must be 8-byte aligned");
    DSW_REQUIRE(x_dtype == dy_dtype || true, "unused");
    const int tiles = (W + TILE_PIX - 1) / TILE_PIX;
    hipStream_t st = as_stream(stream);
    DSW_DISPATCH_DTYPE(dy_dtype, TG, DSW_DISPATCH_DTYPE(dx_dtype, TO, {
        if (accumulate_dx || dx_add != nullptr)
            hipLaunchKernelGGL((scatter_bwd_dx_kernel<TG, TO, true>), dim3(B * H * tiles), dim3(GS_THREADS), 0, st,
                               (const TG *)dy, gates_delta, scores_delta, cellmap, win_h1, win_w1, H, W, C, w, tiles,
                               (TO *)dx, dx_add != nullptr ? (const TO *)dx_add : (const TO *)dx);
        else
            hipLaunchKernelGGL((scatter_bwd_dx_kernel<TG, TO, false>), dim3(B * H * tiles), dim3(GS_THREADS), 0, st,
                               (const TG *)dy, gates_delta, scores_delta, cellmap, win_h1, win_w1, H, W, C, w, tiles,
                               (TO *)dx, (const TO *)nullptr);
    }));
    rc = check_launch("scatter_bwd_dx_kernel");
    if (rc || n_win == 0) return rc;
    // the window-side kernel is instantiated for dy dtype == d_after dtype pairs that occur:
    // fp32 gradients everywhere, or bf16 'after' with fp32 dy.
    DSW_DISPATCH_DTYPE(dy_dtype, TG, DSW_DISPATCH_DTYPE(x_dtype, TX, DSW_DISPATCH_DTYPE(after_dtype, TA, {
        if (d_after_dtype == DSW_F32) {
            hipLaunchKernelGGL((scatter_bwd_win_kernel<TG, TX, TA, float>), dim3(n_win * w), dim3(GS_THREADS), 0, st,
                               (const TG *)dy, (const TX *)x, (const TA *)after, gates_delta, scores_delta, cellmap,
                               win_b, win_h1, win_w1, H, W, C, w, ld_after, ld_d_after, (float *)d_after, partial, n_live);
        } else if (d_after_dtype == DSW_BF16) {
            hipLaunchKernelGGL((scatter_bwd_win_kernel<TG, TX, TA, bf16_t>), dim3(n_win * w), dim3(GS_THREADS), 0, st,
                               (const TG *)dy, (const TX *)x, (const TA *)after, gates_delta, scores_delta, cellmap,
                               win_b, win_h1, win_w1, H, W, C, w, ld_after, ld_d_after, (bf16_t *)d_after, partial, n_live);
        } else {
            return fail(DSW_EUNSUPPORTED, "dsw_scatter_mean_bwd: d_after dtype %d", d_after_dtype);
        }
    })));
    rc = check_launch("scatter_bwd_win_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(scatter_bwd_reduce_kernel, dim3((n_win + 63) / 64), dim3(64), 0, st, partial, n_win, w,
                       d_gates_delta, d_scores_delta);
    return check_launch("scatter_bwd_reduce_kernel");
}

extern "C" int dsw_scatter_mean_bwd(const void *dy, int dy_dtype, const void *x, int x_dtype,
                                    const void *after, int after_dtype, const float *gates_delta,
                                    const float *scores_delta, const int *cellmap, const int *win_b,
                                    const int *win_h1, const int *win_w1, int n_win, int B, int H, int W,
                                    int C, int w, int ld_after, void *dx, int dx_dtype, int accumulate_dx,
                                    void *d_after, int d_after_dtype, int ld_d_after, float *d_gates_delta,
                                    float *d_scores_delta, float *partial, const int *n_live, void *stream) {
    return scatter_mean_bwd_impl(dy, dy_dtype, x, x_dtype, after, after_dtype, gates_delta, scores_delta, cellmap, win_b, win_h1, win_w1,
                                 n_win, B, H, W, C, w, ld_after, dx, dx_dtype, accumulate_dx, nullptr, d_after, d_after_dtype, ld_d_after,
                                 d_gates_delta, d_scores_delta, partial, n_live, stream);
}

extern "C" int dsw_scatter_mean_bwd_add(const void *dy, int dy_dtype, const void *x, int x_dtype,
                                        const void *after, int after_dtype, const float *gates_delta,
                                        const float *scores_delta, const int *cellmap, const int *win_b,
                                        const int *win_h1, const int *win_w1, int n_win, int B, int H, int W,
                                        int C, int w, int ld_after, void *dx, int dx_dtype, const void *dx_add,
                                        void *d_after, int d_after_dtype, int ld_d_after, float *d_gates_delta,
                                        float *d_scores_delta, float *partial, const int *n_live, void *stream) {
    DSW_REQUIRE(dx_add != nullptr && dx_add != dx, "dsw_scatter_mean_bwd_add: dx_add must be a map other than dx");
    return scatter_mean_bwd_impl(dy, dy_dtype, x, x_dtype, after, after_dtype, gates_delta, scores_delta, cellmap, win_b, win_h1, win_w1,
                                 n_win, B, H, W, C, w, ld_after, dx, dx_dtype, 0, dx_add, d_after, d_after_dtype, ld_d_after,
                                 d_gates_delta, d_scores_delta, partial, n_live, stream);
}
