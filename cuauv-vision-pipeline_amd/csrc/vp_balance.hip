// Colour balance on the GPU: utils/color_correction/color_balance.cpp:343-780 `process_frame` (called by
// modules/color_balance.py:93-110 `balance`, modules/preprocessor.py:87-88) and the 8-bit HSV -> BGR conversion it uses.
//
// The reference walks the frame nine times on one core (split, three histogram passes, clips, running means, gains, merge,
// cvtColor, two more histogram passes, stretch, cvtColor back, split, merge).  Every per-pixel step of it is a function of
// the pixel's own channel values once a handful of frame statistics are known, so here a frame is read three times and
// written once:
//   k_cb_hist      B/G/R histograms per equalisation tile (one pass)
//   k_cb_plan1     per frame: percentile clip bounds, channel means, per-tile colour-cast gains, RGB contrast stretch
//                  -> one 256-entry table per (tile, channel)             (double precision, one block per frame)
//   k_cb_hsvhist   table lookup -> BGR2HSV (OpenCV's integer form) -> S and V histograms (second pass)
//   k_cb_plan2     S / V percentile bounds -> stretch tables
//   k_cb_apply     table lookup -> BGR2HSV -> S/V tables -> HSV2BGR (OpenCV's float form) -> store (third pass)
// 12 B/px of algorithmic traffic with the HSV stage (3 reads + 1 write of the frame), 9 B/px without it.
//
// The reference's tile means are a sequential fold (avg += (x - avg) / count, cpp:452-467), not a sum / count.  k_cb_plan1 gets the
// same tables without walking the tile: from the exact mean M and a rigorous bound e >= |fold - M| (derived at cb_fold_bound) it
// checks that every statement that reads the means - the 1/6 test of cpp:474, the choice of the cast, every entry of the gain
// tables - comes out the same for all values in [M - e, M + e]; then the fold's value, whatever it is, gives these tables.  When
// one of them could differ (a comparison within e of its boundary, a product gain * v within e of an integer: about one frame in
// 10^5) three threads run the fold itself, in the reference's order, and the tables are built from its result.
#include "vp_internal.h"
#include <mutex>
#include <cmath>
#include <cstdlib>

#define CB_MAX_TILES 1024
#define CB_COPIES 8

struct cb_params {
    int w, h, n;
    int flags;
    int hb, vb;      // tiles per row / column (already validated: they divide w / h)
    int bw, bh;      // tile size
};

struct HsvTab { int32_t sdiv[256]; int32_t hdiv[256]; };

__device__ __forceinline__ void cb_bgr2hsv(const HsvTab& t, int b, int g, int r, int& H, int& S, int& V)
{
    const int v = max(max(b, g), r), vmin = min(min(b, g), r), diff = v - vmin;
    S = (diff * t.sdiv[v] + 2048) >> 12;
    int hh = (v == r) ? (g - b) : ((v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff));
    hh = (hh * t.hdiv[diff] + 2048) >> 12;
    hh += hh < 0 ? 180 : 0;
    H = hh < 0 ? 0 : (hh > 255 ? 255 : hh);
    V = v;
}

__device__ __forceinline__ int cb_sat_round(float x)   // cv::saturate_cast<uchar>(float): round half to even, clamp
{
    const int v = (int)rintf(x);
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// OpenCV color_hsv.simd.hpp HSV2RGB_b -> HSV2RGB_f, vector arithmetic form (v - v*s, v - (v*s)*h, (v - v*s) + (v*s)*h), hrange 180.
// Built with -ffp-contract=off: every product and sum rounds separately, as the universal intrinsics do.
__device__ __forceinline__ void cb_hsv2bgr(int H, int S, int V, int& b, int& g, int& r)
{
    float h = (float)H * (6.f / 180.f);
    const float s = (float)S * (1.f / 255.f), v = (float)V * (1.f / 255.f);
    const float pre = truncf(h);
    h = h - pre;
    const float vs = v * s;
    const float vsh = vs * h;
    const float t1 = v - vs, t2 = v - vsh, t3 = (v - vs) + vsh;
    float sec = truncf(pre * (1.0f / 6.0f));
    sec = pre - sec * 6.0f;
    const int sector = (int)sec;
    // (b, g, r) = tab[{1,3,0}, {1,0,2}, {3,0,1}, {0,2,1}, {0,1,3}, {2,1,0}][sector], tab = {v, t1, t2, t3}
    float fb, fg, fr;
    switch (sector) {
    case 0: fb = t1; fg = t3; fr = v; break;
    case 1: fb = t1; fg = v; fr = t2; break;
    case 2: fb = t3; fg = v; fr = t1; break;
    case 3: fb = v; fg = t2; fr = t1; break;
    case 4: fb = v; fg = t1; fr = t3; break;
    default: fb = t2; fg = t1; fr = v; break;
    }
    b = cb_sat_round(fb * 255.f);
    g = cb_sat_round(fg * 255.f);
    r = cb_sat_round(fr * 255.f);
}

__device__ __forceinline__ int cb_tile_of(const cb_params& P, size_t pix_in_frame)
{
    const int y = (int)(pix_in_frame / (size_t)P.w), x = (int)(pix_in_frame - (size_t)y * P.w);
    return (y / P.bh) * P.hb + x / P.bw;
}

// ---- pass 1: histograms -------------------------------------------------------------------------------------------------
// grid (blocks, n); a block walks groups of 4 pixels (12 bytes = 3 dwords) of its frame.  hist: [n][tiles][3][256]
template <bool TILED>
__global__ __launch_bounds__(256) void k_cb_hist(const uint8_t* __restrict__ src, cb_params P, u32* __restrict__ hist, u32* __restrict__ neq)
{
    // natural images concentrate on few bins: 16 interleaved copies (bin * 16 + lane % 16) cut same-address atomics of a wave
    // from 64 lanes to 4 and keep the copies of one bin in different banks
    __shared__ u32 lh[3][256 * CB_COPIES];
    const int f = blockIdx.y, tid = threadIdx.x, cp = tid & (CB_COPIES - 1);
    const size_t npx = (size_t)P.w * P.h;
    const uint8_t* fs = src + (size_t)f * npx * 3;
    u32* fh = hist + (size_t)f * P.hb * P.vb * 768;
    if (!TILED) {
        for (int i = tid; i < 768 * CB_COPIES; i += 256) (&lh[0][0])[i] = 0;
        __syncthreads();
    }
    // neq[tile][0..2] become non-zero when some pixel of the tile has B != G, G != R, B != R: channels that agree everywhere
    // have the same sequence of values, hence bitwise equal running means (k_cb_plan1 needs to know)
    u32* fneq = neq + (size_t)f * P.hb * P.vb * 4;
    u32 nq = 0;
    const size_t ngroups = npx / 4;
    const u32* s32 = reinterpret_cast<const u32*>(fs);
    for (size_t gidx = (size_t)blockIdx.x * 256 + tid; gidx < ngroups; gidx += (size_t)gridDim.x * 256) {
        const u32 a = s32[3 * gidx], b = s32[3 * gidx + 1], c = s32[3 * gidx + 2];
        const u32 px[4][3] = {{a & 255, (a >> 8) & 255, (a >> 16) & 255}, {a >> 24, b & 255, (b >> 8) & 255},
                              {(b >> 16) & 255, b >> 24, c & 255}, {(c >> 8) & 255, (c >> 16) & 255, c >> 24}};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 q = (px[k][0] != px[k][1] ? 1u : 0u) | (px[k][1] != px[k][2] ? 2u : 0u) | (px[k][0] != px[k][2] ? 4u : 0u);
            if (TILED) {
                const size_t tl = (size_t)cb_tile_of(P, 4 * gidx + k);
                u32* th = fh + tl * 768;
                atomicAdd(th + px[k][0], 1u); atomicAdd(th + 256 + px[k][1], 1u); atomicAdd(th + 512 + px[k][2], 1u);
                if (q & 1u) fneq[tl * 4] = 1u;
                if (q & 2u) fneq[tl * 4 + 1] = 1u;
                if (q & 4u) fneq[tl * 4 + 2] = 1u;
            } else {
                nq |= q;
                atomicAdd(&lh[0][px[k][0] * CB_COPIES + cp], 1u); atomicAdd(&lh[1][px[k][1] * CB_COPIES + cp], 1u);
                atomicAdd(&lh[2][px[k][2] * CB_COPIES + cp], 1u);
            }
        }
    }
    if (blockIdx.x == 0 && tid < (int)(npx & 3)) {   // tail pixels of the frame
        const size_t p = (npx & ~(size_t)3) + tid;
        const u32 q = (fs[3 * p] != fs[3 * p + 1] ? 1u : 0u) | (fs[3 * p + 1] != fs[3 * p + 2] ? 2u : 0u) | (fs[3 * p] != fs[3 * p + 2] ? 4u : 0u);
        if (TILED) {
            const size_t tl = (size_t)cb_tile_of(P, p);
            u32* th = fh + tl * 768;
            atomicAdd(th + fs[3 * p], 1u); atomicAdd(th + 256 + fs[3 * p + 1], 1u); atomicAdd(th + 512 + fs[3 * p + 2], 1u);
            if (q & 1u) fneq[tl * 4] = 1u;
            if (q & 2u) fneq[tl * 4 + 1] = 1u;
            if (q & 4u) fneq[tl * 4 + 2] = 1u;
        } else {
            nq |= q;
            atomicAdd(&lh[0][fs[3 * p] * CB_COPIES + cp], 1u); atomicAdd(&lh[1][fs[3 * p + 1] * CB_COPIES + cp], 1u);
            atomicAdd(&lh[2][fs[3 * p + 2] * CB_COPIES + cp], 1u);
        }
    }
    if (!TILED) {
        __syncthreads();
        for (int i = tid; i < 768; i += 256) {
            u32 v = 0;
#pragma unroll
            for (int k = 0; k < CB_COPIES; k++) v += (&lh[0][0])[i * CB_COPIES + k];
            if (v) atomicAdd(fh + i, v);
        }
        if (nq & 1u) fneq[0] = 1u;
        if (nq & 2u) fneq[1] = 1u;
        if (nq & 4u) fneq[2] = 1u;
    }
}

// ---- plan 1 -----------------------------------------------------------------------------------------------------------------
// cpp:111-139 percentile_min_max on a histogram
__device__ void cb_percentile(const u32* __restrict__ counts, size_t n, int& mn, int& mx)
{
    int low_bound = (int)(0.002f * (float)n);
    int high_bound = (int)n - (int)(0.998f * (float)n);
    mn = 0; mx = 255;
    for (int i = 0; i < 256; i++) {
        if (low_bound < (int)counts[i]) { mn = i; break; }
        low_bound -= (int)counts[i];
    }
    for (int i = 255; i >= 0; i--) {
        if (high_bound < (int)counts[i]) { mx = i; break; }
        high_bound -= (int)counts[i];
    }
}
__device__ __forceinline__ int cb_cast_u8(double v)   // (unsigned char)double as gcc/x86-64 does it: cvttsd2si (0x80000000 when out of range / NaN), low byte
{
    int i;
    if (!(v > -2147483649.0 && v < 2147483648.0)) i = (int)0x80000000;
    else i = (int)v;
    return i & 0xff;
}
__device__ __forceinline__ int cb_constrain(double v) { return v < 0 ? 0 : (v > 255 ? 255 : cb_cast_u8(v)); }

struct cb_plan {      // per frame, device
    int lo[3], hi[3];             // B, G, R clip bounds (cpp:398-425)
    double avg[3];                // channel means after clipping (cpp:427-429)
    int s_lo, s_hi, v_lo, v_hi;   // cpp:619-629
};

// |fold - exact mean| after n steps of avg += (x - avg) / count over values in [0, 255] (cpp:466-468), every operation rounded to
// nearest (u = 2^-53): one step adds at most d_k = u (2 * 255 / k + 255) (1 + 2u) of error (the subtraction and the division each
// within u of (x - avg) / k <= 255 / k, the addition within u of a value <= 255), and e_n = e_{n-1} (1 - 1/n) + d_n gives
// n e_n = sum k d_k <= u (510 n + 255 n (n + 1) / 2).  A further factor 1.001 swallows the (1 + 2u) terms.
__device__ __forceinline__ double cb_fold_bound(double n) { return (510.0 + 127.5 * (n + 1.0)) * 0x1p-53 * 1.001; }

// The statements of cpp:474-541 for table index v = tid: the three table entries (B, G, R) the tile means l[] (B, G, R) lead to.
__device__ __forceinline__ void cb_tile_entries(const cb_params& P, const cb_plan& pl, const double* l, const double* ratio, const int* cmin,
                                                const double* __restrict__ powtab, int tid, int* out)
{
    const double b_avg = pl.avg[0], g_avg = pl.avg[1], r_avg = pl.avg[2];
    double lb = l[0], lg = l[1], lr = l[2];
    double gain[3] = {1, 1, 1};
    if (P.flags & VP_CB_EQUALIZE_RGB) {
        if (fabs(lr - r_avg) > r_avg / 6 || fabs(lb - b_avg) > b_avg / 6 || fabs(lg - g_avg) > g_avg / 6) { lr = r_avg; lb = b_avg; lg = g_avg; }
        if (lr > lg && lr > lb) { gain[1] = lr / lg; gain[0] = lr / lb; }           // red cast: lift G and B
        else if (lg > lr && lg > lb) { gain[2] = lg / lr; gain[0] = lg / lb; }      // green cast: lift R and B
        else { gain[2] = lb / lr; gain[1] = lb / lg; }                              // blue cast (and ties): lift R and G
    }
    const bool red = lr > lg && lr > lb, green = !red && (lg > lr && lg > lb);
    for (int c = 0; c < 3; c++) {
        int v = min(max(tid, pl.lo[c]), pl.hi[c]);   // clip (no-op without extrema clipping: the value lies inside)
        if (P.flags & VP_CB_EQUALIZE_RGB) {
            const bool lifted = red ? (c != 2) : (green ? (c != 1) : (c != 0));
            if (lifted) {
                if (P.flags & VP_CB_ADAPTIVE_CAST) v = cb_constrain(v * (powtab[v] * (gain[c] - 1.) + 1.));   // powtab[v] = pow((255. - v) / 255., 0.25) from the host's libm
                else v = cb_constrain(v * gain[c]);
            }
        }
        if (P.flags & VP_CB_RGB_CONTRAST) v = cb_cast_u8((v - cmin[c]) * ratio[c]);
        out[c] = v;
    }
}

// one block per frame.  lut: [n][tiles][3][256] u8 (B, G, R); src: the frames (read only when a tile's fold has to be run);
// neq: [n][tiles][4] flags of k_cb_hist; folds (nullable): number of tiles whose fold was run, for the tests
__global__ __launch_bounds__(256) void k_cb_plan1(cb_params P, const u32* __restrict__ hist, const u32* __restrict__ neq,
                                                  const uint8_t* __restrict__ src, cb_plan* __restrict__ plans, uint8_t* __restrict__ lut,
                                                  u32* __restrict__ folds, int force_fold, const double* __restrict__ powtab)
{
    __shared__ u32 gh[3][256];
    __shared__ cb_plan pl;
    __shared__ double tsum[3];
    __shared__ double tcnt;
    __shared__ int certain_s;
    __shared__ double pert[2][3];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int tiles = P.hb * P.vb;
    const size_t npx = (size_t)P.w * P.h;
    const u32* fh = hist + (size_t)f * tiles * 768;
    for (int i = tid; i < 768; i += 256) {
        u32 s = 0;
        for (int t = 0; t < tiles; t++) s += fh[(size_t)t * 768 + i];
        (&gh[0][0])[i] = s;
    }
    __syncthreads();
    if (tid < 3) {
        int mn = 255, mx = 0;
        if (P.flags & VP_CB_EXTREMA_CLIPPING) cb_percentile(gh[tid], npx, mn, mx);
        else {
            for (int i = 0; i < 256; i++) if (gh[tid][i]) { mn = i; break; }
            for (int i = 255; i >= 0; i--) if (gh[tid][i]) { mx = i; break; }
        }
        unsigned long long s = 0;
        for (int i = 0; i < 256; i++) s += (unsigned long long)min(max(i, mn), mx) * gh[tid][i];
        pl.lo[tid] = mn; pl.hi[tid] = mx;
        pl.avg[tid] = (double)s / (double)npx;
    }
    __syncthreads();
    // reference order of the triples is (r, g, b); channel index here is 0 B, 1 G, 2 R
    const double b_avg = pl.avg[0], g_avg = pl.avg[1], r_avg = pl.avg[2];
    // RGB contrast stretch (cpp:545-597): which channel is min / mid / max by mean, then three linear maps
    double ratio[3] = {1, 1, 1};
    int cmin[3] = {0, 0, 0};
    if (P.flags & VP_CB_RGB_CONTRAST) {
        int mxc, mdc, mnc;
        if (r_avg > g_avg) {
            if (r_avg > b_avg) { mxc = 2; if (g_avg > b_avg) { mdc = 1; mnc = 0; } else { mdc = 0; mnc = 1; } }
            else { mxc = 0; mdc = 2; mnc = 1; }
        } else {
            if (g_avg > b_avg) { mxc = 1; if (r_avg > b_avg) { mdc = 2; mnc = 0; } else { mdc = 0; mnc = 2; } }
            else { mxc = 0; mdc = 1; mnc = 2; }
        }
        const double desired_max = (double)((pl.hi[mnc] + pl.hi[mdc] + pl.hi[mxc]) / 3);
        ratio[mnc] = (desired_max - pl.lo[mnc]) / (double)(pl.hi[mnc] - pl.lo[mnc]);
        ratio[mdc] = (desired_max - 0.0) / (double)(pl.hi[mdc] - pl.lo[mdc]);
        ratio[mxc] = (pl.hi[mxc] - 0.0) / (double)(pl.hi[mxc] - pl.lo[mxc]);
        cmin[0] = pl.lo[0]; cmin[1] = pl.lo[1]; cmin[2] = pl.lo[2];
    }
    for (int t = 0; t < tiles; t++) {
        // exact tile means of the clipped channels
        __syncthreads();
        if (tid < 3) {
            const u32* th = fh + (size_t)t * 768 + tid * 256;
            unsigned long long s = 0, c = 0;
            for (int i = 0; i < 256; i++) { s += (unsigned long long)min(max(i, pl.lo[tid]), pl.hi[tid]) * th[i]; c += th[i]; }
            tsum[tid] = c ? (double)s / (double)c : 0.0;
            if (tid == 0) tcnt = (double)c;
        }
        __syncthreads();
        uint8_t* tl = lut + ((size_t)f * tiles + t) * 768;
        int v[3];
        if (!(P.flags & VP_CB_EQUALIZE_RGB)) {       // the means are not read at all
            const double l[3] = {tsum[0], tsum[1], tsum[2]};
            cb_tile_entries(P, pl, l, ratio, cmin, powtab, tid, v);
            for (int c = 0; c < 3; c++) tl[c * 256 + tid] = (uint8_t)v[c];
            continue;
        }
        // ---- can the fold's value matter? ----
        if (tid == 0) {
            const double e = cb_fold_bound(tcnt), M[3] = {tsum[0], tsum[1], tsum[2]};
            const u32* q = neq + ((size_t)f * tiles + t) * 4;
            const bool same01 = q[0] == 0u && pl.lo[0] == pl.lo[1] && pl.hi[0] == pl.hi[1];   // B and G: one sequence of values, one fold
            const bool same12 = q[1] == 0u && pl.lo[1] == pl.lo[2] && pl.hi[1] == pl.hi[2];
            const bool same02 = q[2] == 0u && pl.lo[0] == pl.lo[2] && pl.hi[0] == pl.hi[2];
            const int grp[3] = {0, same01 ? 0 : 1, same02 ? 0 : (same12 ? (same01 ? 0 : 1) : 2)};
            bool certain = !force_fold && tcnt > 0;
            // cpp:474: certainly true for some channel, or certainly false for all (the compared values carry their own rounding:
            // 2^-40 is far above it)
            bool replaced = false, unsure = false;
            for (int c = 0; c < 3; c++) {
                const double d = fabs(M[c] - pl.avg[c]), thr = pl.avg[c] / 6, m = e + 0x1p-40;
                if (d > thr + m) replaced = true;
                else if (d >= thr - m) unsure = true;
            }
            if (!replaced && unsure) certain = false;
            for (int c = 0; c < 3; c++) { pert[0][c] = 0; pert[1][c] = 0; }
            if (certain && !replaced) {
                // which group of channels has the largest mean must be beyond doubt: the two largest DIFFERENT folds at least 2e apart
                int top = 0;
                for (int c = 1; c < 3; c++) if (M[c] > M[top]) top = c;
                for (int c = 0; c < 3; c++)
                    if (grp[c] != grp[top] && M[top] - M[c] <= 2 * e + 0x1p-40) certain = false;
                // the two extreme cases for every gain (largest mean over another): top group down and the rest up, and the reverse
                for (int c = 0; c < 3; c++) { const double sg = grp[c] == grp[top] ? -1.0 : 1.0; pert[0][c] = sg * e; pert[1][c] = -sg * e; }
            }
            certain_s = certain ? 1 : 0;
        }
        __syncthreads();
        bool differ = false;
        if (certain_s) {
            const double la[3] = {tsum[0] + pert[0][0], tsum[1] + pert[0][1], tsum[2] + pert[0][2]};
            const double lb2[3] = {tsum[0] + pert[1][0], tsum[1] + pert[1][1], tsum[2] + pert[1][2]};
            int vb[3];
            cb_tile_entries(P, pl, la, ratio, cmin, powtab, tid, v);
            cb_tile_entries(P, pl, lb2, ratio, cmin, powtab, tid, vb);
            differ = v[0] != vb[0] || v[1] != vb[1] || v[2] != vb[2];
        }
        const bool ok = !__syncthreads_or(differ || !certain_s);
        if (!ok) {
            // ---- the fold itself, in the reference's order (cpp:459-469): one thread per channel ----
            if (tid < 3) {
                const int by = t / P.hb, bx = t - by * P.hb;
                const uint8_t* fs = src + (size_t)f * npx * 3;
                const int lo = pl.lo[tid], hi = pl.hi[tid];
                double avg = 0;
                int count = 0;
                for (int j = 0; j < P.bh; j++) {
                    const uint8_t* row = fs + ((size_t)(by * P.bh + j) * P.w + (size_t)bx * P.bw) * 3 + tid;
                    for (int i = 0; i < P.bw; i++) {
                        const int x = min(max((int)row[3 * (size_t)i], lo), hi);
                        ++count;
                        avg += (x - avg) / count;
                    }
                }
                tsum[tid] = avg;
                if (tid == 0 && folds) atomicAdd(folds, 1u);
            }
            __syncthreads();
            const double l[3] = {tsum[0], tsum[1], tsum[2]};
            cb_tile_entries(P, pl, l, ratio, cmin, powtab, tid, v);
        }
        for (int c = 0; c < 3; c++) tl[c * 256 + tid] = (uint8_t)v[c];
    }
    if (tid == 0) { pl.s_lo = 0; pl.s_hi = 255; pl.v_lo = 0; pl.v_hi = 255; plans[f] = pl; }
}

// ---- pass 2: S / V histograms of the table-mapped frame ---------------------------------------------------------------------
template <bool TILED>
__global__ __launch_bounds__(256) void k_cb_hsvhist(const uint8_t* __restrict__ src, cb_params P, vp_tables tab, const uint8_t* __restrict__ lut,
                                                    u32* __restrict__ svhist)
{
    __shared__ HsvTab ht;
    __shared__ uint8_t ll[768];
    __shared__ u32 lh[2][256 * CB_COPIES];
    const int f = blockIdx.y, tid = threadIdx.x, cp = tid & (CB_COPIES - 1);
    const int tiles = P.hb * P.vb;
    const size_t npx = (size_t)P.w * P.h;
    const uint8_t* fs = src + (size_t)f * npx * 3;
    const uint8_t* fl = lut + (size_t)f * tiles * 768;
    ht.sdiv[tid] = tab.sdiv[tid]; ht.hdiv[tid] = tab.hdiv[tid];
    for (int i = tid; i < 768; i += 256) ll[i] = fl[i];
    for (int i = tid; i < 512 * CB_COPIES; i += 256) (&lh[0][0])[i] = 0;
    __syncthreads();
    auto one = [&](size_t p, int b, int g, int r) {
        if (TILED) { const uint8_t* tl = fl + (size_t)cb_tile_of(P, p) * 768; b = tl[b]; g = tl[256 + g]; r = tl[512 + r]; }
        else { b = ll[b]; g = ll[256 + g]; r = ll[512 + r]; }
        int H, S, V;
        cb_bgr2hsv(ht, b, g, r, H, S, V);
        atomicAdd(&lh[0][S * CB_COPIES + cp], 1u);
        atomicAdd(&lh[1][V * CB_COPIES + cp], 1u);
    };
    const size_t ngroups = npx / 4;
    const u32* s32 = reinterpret_cast<const u32*>(fs);
    for (size_t gidx = (size_t)blockIdx.x * 256 + tid; gidx < ngroups; gidx += (size_t)gridDim.x * 256) {
        const u32 a = s32[3 * gidx], bb = s32[3 * gidx + 1], c = s32[3 * gidx + 2];
        one(4 * gidx, (int)(a & 255), (int)((a >> 8) & 255), (int)((a >> 16) & 255));
        one(4 * gidx + 1, (int)(a >> 24), (int)(bb & 255), (int)((bb >> 8) & 255));
        one(4 * gidx + 2, (int)((bb >> 16) & 255), (int)(bb >> 24), (int)(c & 255));
        one(4 * gidx + 3, (int)((c >> 8) & 255), (int)((c >> 16) & 255), (int)(c >> 24));
    }
    if (blockIdx.x == 0 && tid < (int)(npx & 3)) {
        const size_t p = (npx & ~(size_t)3) + tid;
        one(p, fs[3 * p], fs[3 * p + 1], fs[3 * p + 2]);
    }
    __syncthreads();
    for (int i = tid; i < 512; i += 256) {
        u32 v = 0;
#pragma unroll
        for (int k = 0; k < CB_COPIES; k++) v += (&lh[0][0])[i * CB_COPIES + k];
        if (v) atomicAdd(svhist + (size_t)f * 512 + i, v);
    }
}

// S / V clip bounds and stretch tables (cpp:619-660).  svlut: [n][2][256]
__global__ __launch_bounds__(256) void k_cb_plan2(cb_params P, const u32* __restrict__ svhist, cb_plan* __restrict__ plans, uint8_t* __restrict__ svlut)
{
    __shared__ int bounds[4];
    const int f = blockIdx.x, tid = threadIdx.x;
    const size_t npx = (size_t)P.w * P.h;
    if (tid < 2) cb_percentile(svhist + (size_t)f * 512 + tid * 256, npx, bounds[2 * tid], bounds[2 * tid + 1]);
    __syncthreads();
    for (int c = 0; c < 2; c++) {
        const int lo = bounds[2 * c], hi = bounds[2 * c + 1];
        int v = min(max(tid, lo), hi);
        if (hi != lo) v = ((v - lo) * 255) / (hi - lo);   // the reference divides by zero here (SIGFPE); the channel stays as clipped
        svlut[(size_t)f * 512 + c * 256 + tid] = (uint8_t)v;
    }
    if (tid == 0) { plans[f].s_lo = bounds[0]; plans[f].s_hi = bounds[1]; plans[f].v_lo = bounds[2]; plans[f].v_hi = bounds[3]; }
}

// ---- pass 3: apply --------------------------------------------------------------------------------------------------------
template <bool TILED, bool HSV>
__global__ __launch_bounds__(256) void k_cb_apply(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, cb_params P, vp_tables tab,
                                                  const uint8_t* __restrict__ lut, const uint8_t* __restrict__ svlut)
{
    __shared__ HsvTab ht;
    __shared__ uint8_t ll[768];
    __shared__ uint8_t sl[512];
    const int f = blockIdx.y, tid = threadIdx.x;
    const int tiles = P.hb * P.vb;
    const size_t npx = (size_t)P.w * P.h;
    const uint8_t* fs = src + (size_t)f * npx * 3;
    uint8_t* fd = dst + (size_t)f * npx * 3;
    const uint8_t* fl = lut + (size_t)f * tiles * 768;
    if (HSV) {
        ht.sdiv[tid] = tab.sdiv[tid]; ht.hdiv[tid] = tab.hdiv[tid];
        for (int i = tid; i < 512; i += 256) sl[i] = svlut[(size_t)f * 512 + i];
    }
    for (int i = tid; i < 768; i += 256) ll[i] = fl[i];
    __syncthreads();
    const size_t ngroups = npx / 4;
    const u32* s32 = reinterpret_cast<const u32*>(fs);
    u32* d32 = reinterpret_cast<u32*>(fd);
    auto one = [&](size_t p, int& b, int& g, int& r) {
        if (TILED) { const uint8_t* tl = fl + (size_t)cb_tile_of(P, p) * 768; b = tl[b]; g = tl[256 + g]; r = tl[512 + r]; }
        else { b = ll[b]; g = ll[256 + g]; r = ll[512 + r]; }
        if (HSV) {
            int H, S, V;
            cb_bgr2hsv(ht, b, g, r, H, S, V);
            cb_hsv2bgr(H, sl[S], sl[256 + V], b, g, r);
        }
    };
    for (size_t gidx = (size_t)blockIdx.x * 256 + tid; gidx < ngroups; gidx += (size_t)gridDim.x * 256) {
        const u32 a = s32[3 * gidx], bb = s32[3 * gidx + 1], c = s32[3 * gidx + 2];
        int px[4][3] = {{(int)(a & 255), (int)((a >> 8) & 255), (int)((a >> 16) & 255)}, {(int)(a >> 24), (int)(bb & 255), (int)((bb >> 8) & 255)},
                        {(int)((bb >> 16) & 255), (int)(bb >> 24), (int)(c & 255)}, {(int)((c >> 8) & 255), (int)((c >> 16) & 255), (int)(c >> 24)}};
#pragma unroll
        for (int k = 0; k < 4; k++) one(4 * gidx + k, px[k][0], px[k][1], px[k][2]);
        d32[3 * gidx] = (u32)px[0][0] | ((u32)px[0][1] << 8) | ((u32)px[0][2] << 16) | ((u32)px[1][0] << 24);
        d32[3 * gidx + 1] = (u32)px[1][1] | ((u32)px[1][2] << 8) | ((u32)px[2][0] << 16) | ((u32)px[2][1] << 24);
        d32[3 * gidx + 2] = (u32)px[2][2] | ((u32)px[3][0] << 8) | ((u32)px[3][1] << 16) | ((u32)px[3][2] << 24);
    }
    if (blockIdx.x == 0 && tid < (int)(npx & 3)) {
        const size_t p = (npx & ~(size_t)3) + tid;
        int b = fs[3 * p], g = fs[3 * p + 1], r = fs[3 * p + 2];
        one(p, b, g, r);
        fd[3 * p] = (uint8_t)b; fd[3 * p + 1] = (uint8_t)g; fd[3 * p + 2] = (uint8_t)r;
    }
}

// ---- HSI contrast stage (cpp:141-341, 678-775) -------------------------------------------------------------------------------
// Works on the BGR image the earlier stages produced (in place).  The reference converts to float H, S, I planes, takes the
// 0.2 % / 99.8 % order statistics of S and I by quickselect, clips, stretches and converts back.  S and I need no
// trigonometry, so the order statistics come from four 8-bit radix passes over the image that recompute (S, I) per pixel
// (3 B/px per pass, no float planes), driven entirely from the device (k_hsi_pick chooses the bucket of each pass); the
// final pass recomputes H as well and writes the result.  float / double mixing follows the C++ expressions operand by
// operand; acos / cos come from the device libm, hence the tolerance of 1 stated in the tests for this stage.
#define CB_PI 3.14159265358979323846
struct hsi_state { u32 prefix[4], mask[4], rank[4]; float val[4]; };   // 0 S-low, 1 S-high, 2 I-low, 3 I-high

__device__ __forceinline__ float cb_clipf(float c, float mn, float mx)   // cpp:47-69 clip_channel_f, one element
{
    if (c < mn) return mn;
    if (c > mx) return mx;
    if (isnan(c)) return mn;
    if (isinf(c)) return mx;
    return c;
}
__device__ __forceinline__ void cb_rgb2si(int r, int g, int b, float& S, float& I)   // cpp:186-201 + the clips of cpp:253-255
{
    I = (float)((double)((float)r + (float)g + (float)b) / 3.);
    const int mn = min(min(r, g), b);
    S = I > 0 ? (float)(1. - (double)((float)mn / I)) : 0.f;
    S = cb_clipf(S, 0.f, 1.f);
    I = cb_clipf(I, 0.f, 255.f);
}
__device__ __forceinline__ float cb_rgb2h(int r, int g, int b)   // cpp:202-207
{
    const float rad = (float)r * r + (float)g * g + (float)b * b - (float)(r * g) - (float)(r * b) - (float)(g * b);
    float H = (float)acos(((double)(float)r - (0.5 * g) - (0.5 * b)) / sqrt((double)rad));
    if (b > g) H = (float)((CB_PI * 2) - (double)H);
    return cb_clipf(H, 0.f, (float)(2. * CB_PI));
}
__device__ __forceinline__ int cb_uchar_clip(float f)   // cpp:155-164; (int) of NaN / huge values as x86 cvttss2si: INT_MIN
{
    int n = (f > -2147483904.f && f < 2147483648.f) ? (int)f : (int)0x80000000;
    return n < 0 ? 0 : (n > 255 ? 255 : n);
}
__device__ __forceinline__ bool cb_feq(float a, float b) { return fabs((double)(a - b)) < 0.000001; }
__device__ __forceinline__ void cb_hsi2rgb(float h, float s, float i, int& r, int& g, int& b)   // cpp:261-306
{
    const float lo = i - i * s;
    if (cb_feq(h, 0.f)) { r = cb_uchar_clip(i + 2 * i * s); g = cb_uchar_clip(lo); b = cb_uchar_clip(lo); }
    else if (0. < h && h < 2. * CB_PI / 3.) {
        const double q = cos((double)h) / cos(CB_PI / 3. - h);
        r = cb_uchar_clip((float)(i + i * s * q));
        g = cb_uchar_clip((float)(i + i * s * (1 - q)));
        b = cb_uchar_clip(lo);
    } else if (cb_feq(h, (float)(2. * CB_PI / 3.))) { r = cb_uchar_clip(lo); g = cb_uchar_clip(i + 2 * i * s); b = cb_uchar_clip(lo); }
    else if (2. * CB_PI / 3. < h && h < 4. * CB_PI / 3.) {
        const double q = cos(h - 2. * CB_PI / 3.) / cos(CB_PI - h);
        r = cb_uchar_clip(lo);
        g = cb_uchar_clip((float)(i + i * s * q));
        b = cb_uchar_clip((float)(i + i * s * (1 - q)));
    } else if (cb_feq(h, (float)(4. * CB_PI / 3.))) { r = cb_uchar_clip(lo); g = cb_uchar_clip(lo); b = cb_uchar_clip(i + 2 * i * s); }
    else {
        const double q = cos(h - 4. * CB_PI / 3.) / cos(5. * CB_PI / 3. - h);
        r = cb_uchar_clip((float)(i + i * s * (1 - q)));
        g = cb_uchar_clip(lo);
        b = cb_uchar_clip((float)(i + i * s * q));
    }
}

__global__ void k_hsi_init(cb_params P, hsi_state* __restrict__ st)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= P.n) return;
    const size_t npx = (size_t)P.w * P.h;
    const u32 lowb = (u32)(int)(0.002f * (float)npx);          // cpp:142-143
    u32 highb = (u32)(int)(0.998f * (float)npx);
    if (highb >= npx) highb = (u32)npx - 1;
    hsi_state s;
    for (int q = 0; q < 4; q++) { s.prefix[q] = 0; s.mask[q] = 0; s.rank[q] = (q & 1) ? highb : lowb; s.val[q] = 0.f; }
    st[f] = s;
}

// one radix pass: digit histograms of the S and I keys that still match each of the four prefixes.  hist: [n][4][256]
__global__ __launch_bounds__(256) void k_hsi_hist(const uint8_t* __restrict__ img, cb_params P, const hsi_state* __restrict__ st, int shift,
                                                  u32* __restrict__ hist)
{
    __shared__ u32 lh[4][256 * 8];
    const int f = blockIdx.y, tid = threadIdx.x, cp = tid & 7;
    const size_t npx = (size_t)P.w * P.h;
    const uint8_t* fs = img + (size_t)f * npx * 3;
    const hsi_state s = st[f];
    for (int i = tid; i < 4 * 256 * 8; i += 256) (&lh[0][0])[i] = 0;
    __syncthreads();
    auto one = [&](int b, int g, int r) {
        float S, I;
        cb_rgb2si(r, g, b, S, I);
        const u32 ks = __float_as_uint(S), ki = __float_as_uint(I);
        if ((ks & s.mask[0]) == s.prefix[0]) atomicAdd(&lh[0][((ks >> shift) & 255u) * 8 + cp], 1u);
        if ((ks & s.mask[1]) == s.prefix[1]) atomicAdd(&lh[1][((ks >> shift) & 255u) * 8 + cp], 1u);
        if ((ki & s.mask[2]) == s.prefix[2]) atomicAdd(&lh[2][((ki >> shift) & 255u) * 8 + cp], 1u);
        if ((ki & s.mask[3]) == s.prefix[3]) atomicAdd(&lh[3][((ki >> shift) & 255u) * 8 + cp], 1u);
    };
    const size_t ngroups = npx / 4;
    const u32* s32 = reinterpret_cast<const u32*>(fs);
    for (size_t gidx = (size_t)blockIdx.x * 256 + tid; gidx < ngroups; gidx += (size_t)gridDim.x * 256) {
        const u32 a = s32[3 * gidx], bb = s32[3 * gidx + 1], c = s32[3 * gidx + 2];
        one((int)(a & 255), (int)((a >> 8) & 255), (int)((a >> 16) & 255));
        one((int)(a >> 24), (int)(bb & 255), (int)((bb >> 8) & 255));
        one((int)((bb >> 16) & 255), (int)(bb >> 24), (int)(c & 255));
        one((int)((c >> 8) & 255), (int)((c >> 16) & 255), (int)(c >> 24));
    }
    if (blockIdx.x == 0 && tid < (int)(npx & 3)) {
        const size_t p = (npx & ~(size_t)3) + tid;
        one(fs[3 * p], fs[3 * p + 1], fs[3 * p + 2]);
    }
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) {
        u32 v = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) v += (&lh[0][0])[i * 8 + k];
        if (v) atomicAdd(hist + (size_t)f * 1024 + i, v);
    }
}

// one block per frame: walk each of the four histograms to the bucket holding the wanted rank, clear the histograms
__global__ __launch_bounds__(256) void k_hsi_pick(hsi_state* __restrict__ st, u32* __restrict__ hist, int shift)
{
    const int f = blockIdx.x, tid = threadIdx.x;
    u32* h = hist + (size_t)f * 1024;
    if (tid < 4) {
        hsi_state* s = st + f;
        u32 rank = s->rank[tid];
        int b = 0;
        for (; b < 255; b++) {
            const u32 c = h[tid * 256 + b];
            if (rank < c) break;
            rank -= c;
        }
        s->rank[tid] = rank;
        s->prefix[tid] |= (u32)b << shift;
        s->mask[tid] |= 0xffu << shift;
        if (shift == 0) s->val[tid] = __uint_as_float(s->prefix[tid]);
    }
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) h[i] = 0;
}

__global__ __launch_bounds__(256) void k_hsi_apply(uint8_t* __restrict__ img, cb_params P, const hsi_state* __restrict__ st)
{
    const int f = blockIdx.y, tid = threadIdx.x;
    const size_t npx = (size_t)P.w * P.h;
    uint8_t* fs = img + (size_t)f * npx * 3;
    const float s_min = st[f].val[0], s_max = st[f].val[1], i_min = st[f].val[2], i_max = st[f].val[3];
    const float s_mult = (float)(1. / (double)(s_max - s_min)), i_mult = (float)(255. / (double)(i_max - i_min));   // cpp:747-748
    for (size_t p = (size_t)blockIdx.x * 256 + tid; p < npx; p += (size_t)gridDim.x * 256) {
        int b = fs[3 * p], g = fs[3 * p + 1], r = fs[3 * p + 2];
        float S, I;
        cb_rgb2si(r, g, b, S, I);
        const float H = cb_rgb2h(r, g, b);
        S = cb_clipf(S, s_min, s_max);
        I = cb_clipf(I, i_min, i_max);
        S = (S - s_min) * s_mult;
        I = (I - i_min) * i_mult;
        S = cb_clipf(S, 0.f, 1.f);
        I = cb_clipf(I, 0.f, 255.f);
        cb_hsi2rgb(H, S, I, r, g, b);
        fs[3 * p] = (uint8_t)b; fs[3 * p + 1] = (uint8_t)g; fs[3 * p + 2] = (uint8_t)r;
    }
}

// HSV -> BGR as an operator (cv2.cvtColor(COLOR_HSV2BGR), 8-bit): packed rows
__global__ __launch_bounds__(256) void k_hsv2bgr(const uint8_t* __restrict__ src, size_t npx, uint8_t* __restrict__ dst)
{
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npx; p += (size_t)gridDim.x * 256) {
        int b, g, r;
        cb_hsv2bgr(src[3 * p], src[3 * p + 1], src[3 * p + 2], b, g, r);
        dst[3 * p] = (uint8_t)b; dst[3 * p + 1] = (uint8_t)g; dst[3 * p + 2] = (uint8_t)r;
    }
}

int vpk_hsv2bgr(vp_ctx* ctx, const uint8_t* d_src, size_t npx, uint8_t* d_dst)
{
    vp_prof_scope ps(ctx, VPK_OTHER);
    const unsigned blocks = (unsigned)std::min<size_t>((npx + 255) / 256, (size_t)ctx->num_cu * 16);
    hipLaunchKernelGGL(k_hsv2bgr, dim3(blocks), dim3(256), 0, ctx->stream, d_src, npx, d_dst);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

size_t vp_balance_ws_bytes(int n, int tiles) { return vp_align((size_t)n * tiles * 772 * 4 + 64) + vp_align((size_t)n * 1024 * 4) + vp_align((size_t)n * tiles * 768) +
                                                      vp_align((size_t)n * 512) + vp_align(sizeof(cb_plan) * (size_t)n) + vp_align(sizeof(hsi_state) * (size_t)n) + 4096 + 2048; }

// d_src / d_dst: (n, h, w, 3) packed; d_dst may equal d_src.  Frames must start 4-byte aligned (w*h*3 % 4 == 0 or n == 1).
int vpk_color_balance(vp_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int w, int h, int n, int flags, int hblocks, int vblocks)
{
    if (hblocks <= 0 || vblocks <= 0) return vp_fail(ctx, VP_ERR_INVALID, "colour balance: tiles");
    if ((flags & VP_CB_EQUALIZE_RGB) && (w % hblocks || h % vblocks))
        return vp_fail(ctx, VP_ERR_UNSUPPORTED, "colour balance: tiles must divide the frame (the reference wraps rows otherwise)");
    if (!(flags & VP_CB_EQUALIZE_RGB)) { hblocks = 1; vblocks = 1; }
    const int tiles = hblocks * vblocks;
    if (tiles > CB_MAX_TILES) return vp_fail(ctx, VP_ERR_UNSUPPORTED, "colour balance: too many tiles");
    const size_t npx = (size_t)w * h;
    if (n > 1 && (npx * 3) % 4) return vp_fail(ctx, VP_ERR_UNSUPPORTED, "colour balance: batched frames must be 4-byte aligned");
    cb_params P = {w, h, n, flags, hblocks, vblocks, w / hblocks, h / vblocks};
    u32* hist = (u32*)vp_ws_take(ctx, (size_t)n * tiles * 772 * 4 + 64);   // histograms | per-tile channel-equality flags | fold counter
    u32* neq = hist ? hist + (size_t)n * tiles * 768 : nullptr;
    u32* folds = neq ? neq + (size_t)n * tiles * 4 : nullptr;
    u32* svhist = (u32*)vp_ws_take(ctx, (size_t)n * 1024 * 4);   // S/V histograms; reused as the four radix histograms of the HSI stage
    uint8_t* lut = (uint8_t*)vp_ws_take(ctx, (size_t)n * tiles * 768);
    uint8_t* svlut = (uint8_t*)vp_ws_take(ctx, (size_t)n * 512);
    cb_plan* plans = (cb_plan*)vp_ws_take(ctx, sizeof(cb_plan) * (size_t)n);
    hsi_state* hst = (hsi_state*)vp_ws_take(ctx, sizeof(hsi_state) * (size_t)n);
    if (!hist || !svhist || !lut || !svlut || !plans || !hst) return vp_fail(ctx, VP_ERR_NOMEM, "colour balance workspace");
    hipStream_t s = ctx->stream;
    const bool tiled = tiles > 1, hsv = (flags & VP_CB_HSV_CONTRAST) != 0;
    // enough blocks to fill the chip, few enough that the per-block table loads and histogram flushes stay small
    const unsigned bx = (unsigned)std::max<size_t>(1, std::min<size_t>((npx / 4 + 255) / 256, std::max<size_t>(8, (size_t)ctx->num_cu * 8 / (size_t)n)));
    const dim3 grid(bx, (unsigned)n);
    vp_prof_scope ps(ctx, VPK_OTHER);
    VP_HIP(ctx, hipMemsetAsync(hist, 0, (size_t)n * tiles * 772 * 4 + 64, s));
    if (tiled) hipLaunchKernelGGL((k_cb_hist<true>), grid, dim3(256), 0, s, d_src, P, hist, neq);
    else hipLaunchKernelGGL((k_cb_hist<false>), grid, dim3(256), 0, s, d_src, P, hist, neq);
    static const int force_fold = getenv("VP_CB_FORCE_FOLD") ? atoi(getenv("VP_CB_FORCE_FOLD")) : 0;   // tests: run the fold for every tile
    // adaptive_cast_correction (cpp:489-490) calls pow((255. - v) / 255., 0.25): 256 possible arguments.  The table comes from the
    // host's libm - the library the reference itself would call on this machine - so that the device's own pow (which may round the
    // last bit differently) never enters the result.
    double* powtab = nullptr;
    if (flags & VP_CB_ADAPTIVE_CAST) {
        static double host_tab[256];
        static std::once_flag made;                  // contexts are per thread: two of them may get here together
        std::call_once(made, [] { for (int v = 0; v < 256; v++) host_tab[v] = pow((255. - v) / 255., 0.25); });
        powtab = (double*)vp_ws_take(ctx, sizeof host_tab);
        if (!powtab) return vp_fail(ctx, VP_ERR_NOMEM, "colour balance workspace");
        VP_HIP(ctx, hipMemcpyAsync(powtab, host_tab, sizeof host_tab, hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL(k_cb_plan1, dim3((unsigned)n), dim3(256), 0, s, P, hist, neq, d_src, plans, lut, folds, force_fold, (const double*)powtab);
    VP_HIP(ctx, hipMemcpyAsync(ctx->cb_folds_own, folds, 4, hipMemcpyDeviceToDevice, s));
    ctx->cb_folds_dev = ctx->cb_folds_own;
    if (hsv) {
        VP_HIP(ctx, hipMemsetAsync(svhist, 0, (size_t)n * 512 * 4, s));
        if (tiled) hipLaunchKernelGGL((k_cb_hsvhist<true>), grid, dim3(256), 0, s, d_src, P, ctx->tab, lut, svhist);
        else hipLaunchKernelGGL((k_cb_hsvhist<false>), grid, dim3(256), 0, s, d_src, P, ctx->tab, lut, svhist);
        hipLaunchKernelGGL(k_cb_plan2, dim3((unsigned)n), dim3(256), 0, s, P, svhist, plans, svlut);
    }
    if (tiled) {
        if (hsv) hipLaunchKernelGGL((k_cb_apply<true, true>), grid, dim3(256), 0, s, d_src, d_dst, P, ctx->tab, lut, svlut);
        else hipLaunchKernelGGL((k_cb_apply<true, false>), grid, dim3(256), 0, s, d_src, d_dst, P, ctx->tab, lut, svlut);
    } else {
        if (hsv) hipLaunchKernelGGL((k_cb_apply<false, true>), grid, dim3(256), 0, s, d_src, d_dst, P, ctx->tab, lut, svlut);
        else hipLaunchKernelGGL((k_cb_apply<false, false>), grid, dim3(256), 0, s, d_src, d_dst, P, ctx->tab, lut, svlut);
    }
    if (flags & VP_CB_HSI_CONTRAST) {
        VP_HIP(ctx, hipMemsetAsync(svhist, 0, (size_t)n * 1024 * 4, s));
        hipLaunchKernelGGL(k_hsi_init, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, P, hst);
        for (int shift = 24; shift >= 0; shift -= 8) {
            hipLaunchKernelGGL(k_hsi_hist, grid, dim3(256), 0, s, d_dst, P, hst, shift, svhist);
            hipLaunchKernelGGL(k_hsi_pick, dim3((unsigned)n), dim3(256), 0, s, hst, svhist, shift);
        }
        hipLaunchKernelGGL(k_hsi_apply, grid, dim3(256), 0, s, d_dst, P, hst);
    }
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}
