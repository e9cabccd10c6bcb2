// Contour extraction on the GPU (included by vp_ccl.hip; shares its union-find device code).
//
// Replaces utils/feature.py:5-40 `outer_contours` / `all_contours` = cv2.findContours(RETR_EXTERNAL | RETR_LIST,
// CHAIN_APPROX_SIMPLE | NONE).  OpenCV (imgproc/src/contours.cpp) finds borders with a sequential raster scan that
// marks pixels as it goes; what it returns can be stated without the scan:
//   * one outer border per 8-connected foreground component, starting at the component's first pixel in raster order;
//   * one hole border per 4-connected background region that does not reach the image frame, starting at the
//     foreground pixel left of the region's first pixel;
//   * RETR_LIST returns all of them, RETR_EXTERNAL the outer borders of components that are not inside a hole;
//   * order: by start pixel, raster order, newest (= last) first;
//   * each border is the Suzuki-Abe trace from its start pixel, which depends on the binary image only.
// So: two union-find passes (foreground 8-conn, background 4-conn, both with first-pixel ids = VP_CCL_PIXEL), a
// bitmap of start pixels + popcount prefix for the order, and one wave per border for the sequential trace
// (counting pass, exclusive scan of the counts, writing pass).
//
// Known divergence (documented in DESIGN.md): OpenCV's RETR_EXTERNAL decides "inside a hole" from the sign of the
// last mark left of the start pixel, which differs from the topological rule when a one-pixel-thick wall pixel was
// negatively marked by its outer trace; tests/test_gpu_contours.py counts such cases on random masks.

struct ct_frame_out {      // per frame, device
    int32_t n_contours;
    int32_t n_points;
};

// outside[] bit per background root: the region touches the image frame
__global__ __launch_bounds__(256) void k_ct_outside(const u64* __restrict__ bits, ccl_geom G, const u32* __restrict__ parent, u32* __restrict__ outside)
{
    // one thread per frame-border word: rows 0 and h-1 fully, columns 0 and ww-1 of the other rows
    const int f = blockIdx.y;
    const int nborder = 2 * G.ww + 2 * G.h;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nborder) return;
    int y, j, which;   // which: 0 = every segment of the word, 1 = the one touching bit 0, 2 = the one touching the last valid bit
    if (t < G.ww) { y = 0; j = t; which = 0; }
    else if (t < 2 * G.ww) { y = G.h - 1; j = t - G.ww; which = 0; }
    else if (t < 2 * G.ww + G.h) { y = t - 2 * G.ww; j = 0; which = 1; }
    else { y = t - 2 * G.ww - G.h; j = G.ww - 1; which = 2; }
    const u64* fb = bits + (size_t)f * G.h * G.ww;
    const u64 w = ccl_word(G, fb, y * G.ww + j, j);
    if (!w) return;
    const u32* p = parent + (size_t)f * G.nids;
    u32* o = outside + (size_t)f * G.nw32;
    u64 rem = w;
    if (which == 1) rem = (w & 1ull) ? bit_range(0, run_end(w, 0)) : 0ull;
    if (which == 2) { const int last = ((G.w - 1) & 63); rem = ((w >> last) & 1ull) ? bit_range(run_start(w, last), last) : 0ull; }
    while (rem) {
        const int s = __ffsll((long long)rem) - 1;
        const int e = run_end(rem, s);
        rem &= ~bit_range(s, e);
        const int st = run_start(w, s);   // `rem` may have been cut: the segment id comes from the real start
        u32 r = seg_id(G, y, 64 * j + st);
        for (u32 q = p[r]; q != r; q = p[r]) r = q;
        atomicOr(o + (r >> 5), 1u << (r & 31));
    }
}

// root id (VP_CCL_PIXEL: y*wb + x/2) -> first pixel of the region
__device__ __forceinline__ void ct_root_pixel(const ccl_geom& G, const u64* __restrict__ fb, u32 id, int& y, int& x)
{
    y = (int)(id / (u32)G.wb);
    const int x2 = (int)(id - (u32)y * (u32)G.wb);
    const int j = (2 * x2) >> 6;
    const u64 w = ccl_word(G, fb, y * G.ww + j, j);
    x = ((w >> ((2 * x2) & 63)) & 1ull) ? 2 * x2 : 2 * x2 + 1;
}

// seeds: start-pixel bitmap (same layout as a bit image) + hole bitmap.
// grid (ceil(nw32/256), n): thread = one 32-bit word of the root bitmaps.
__global__ __launch_bounds__(256) void k_ct_seeds(const u64* __restrict__ bits, ccl_geom Gf, ccl_geom Gb, const u32* __restrict__ fg_flags,
                                                  const u32* __restrict__ bg_flags, const u32* __restrict__ bg_parent,
                                                  const u32* __restrict__ outside, int mode, u64* __restrict__ startmap, u64* __restrict__ holemap)
{
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    if (t >= Gf.nw32) return;
    const int f = blockIdx.y;
    const u64* fb = bits + (size_t)f * Gf.h * Gf.ww;
    u64* sm = startmap + (size_t)f * Gf.h * Gf.ww;
    u64* hm = holemap + (size_t)f * Gf.h * Gf.ww;
    const u32* bp = bg_parent + (size_t)f * Gb.nids;
    const u32* out = outside + (size_t)f * Gb.nw32;
    u32 m = fg_flags[(size_t)f * Gf.nw32 + t];
    while (m) {
        const int b = __ffs((int)m) - 1;
        m &= m - 1;
        int y, x;
        ct_root_pixel(Gf, fb, t * 32 + b, y, x);
        bool keep = true;
        if (mode == 0 && x > 0) {   // RETR_EXTERNAL: the region left of the first pixel must reach the frame
            const int xl = x - 1, j = xl >> 6;
            const u64 wb = ccl_word(Gb, fb, y * Gb.ww + j, j);
            u32 r = seg_id(Gb, y, 64 * j + run_start(wb, xl & 63));
            for (u32 q = bp[r]; q != r; q = bp[r]) r = q;
            keep = (out[r >> 5] >> (r & 31)) & 1u;
        }
        if (keep) atomicOr((unsigned long long*)&sm[y * Gf.ww + (x >> 6)], 1ull << (x & 63));
    }
    if (mode == 1) {   // RETR_LIST: hole borders
        u32 hb = bg_flags[(size_t)f * Gb.nw32 + t] & ~out[t];
        while (hb) {
            const int b = __ffs((int)hb) - 1;
            hb &= hb - 1;
            int y, x;
            ct_root_pixel(Gb, fb, t * 32 + b, y, x);
            const int xs = x - 1;   // a hole never touches column 0
            atomicOr((unsigned long long*)&sm[y * Gf.ww + (xs >> 6)], 1ull << (xs & 63));
            atomicOr((unsigned long long*)&hm[y * Gf.ww + (xs >> 6)], 1ull << (xs & 63));
        }
    }
}

// per frame: popcount prefix over the start bitmap -> list of border starts in raster order
// starts[f][rank] = pixel index (y*w + x) | hole << 31      (one block per frame; the bitmap is small)
__global__ __launch_bounds__(1024) void k_ct_rank(const u64* __restrict__ startmap, const u64* __restrict__ holemap, int nwords, int ww, int w,
                                                  u32* __restrict__ starts, int max_contours, ct_frame_out* __restrict__ out)
{
    __shared__ u32 wsum[16];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u64* sm = startmap + (size_t)f * nwords;
    const u64* hm = holemap + (size_t)f * nwords;
    u32* st = starts + (size_t)f * max_contours;
    const int per = (nwords + 1023) / 1024;
    const int lo = min(tid * per, nwords), hi = min(lo + per, nwords);
    u32 cnt = 0;
    for (int i = lo; i < hi; i++) cnt += (u32)__popcll(sm[i]);
    u32 inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    u32 woff = 0, total = 0;
    for (int k = 0; k < 16; k++) { if (k < wv) woff += wsum[k]; total += wsum[k]; }
    u32 run = woff + inc - cnt;
    if (cnt) {
        for (int i = lo; i < hi; i++) {
            u64 m = sm[i];
            if (!m) continue;
            const u64 hb = hm[i];
            const int y = i / ww, j = i - y * ww;
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                if (run < (u32)max_contours) st[run] = (u32)(y * w + 64 * j + b) | (((hb >> b) & 1ull) ? 0x80000000u : 0u);
                run++;
            }
        }
    }
    if (tid == 0) out[f].n_contours = (int32_t)total;
}

// ---- sequential trace of one border by one wave ----------------------------------------------------------------------
// All 64 lanes run the same trace (uniform control flow); lane 0 writes.  What the other lanes buy is the window: a
// CTW_ROWS x 64*CTW_WORDS pixel piece of the mask around the current pixel, loaded by the whole wave in one round trip to
// memory and kept in LDS, so that a trace step never waits on HBM/L2 - only every >= 64 steps, when the border leaves
// the window.  Inside the window the 3x3 neighbourhoods come from an 8x8 tile held in one register.
#define CTW_ROWS 128
#define CTW_WORDS 4

struct ct_win { int y0, j0; };   // origin: row y0, word column j0 (either may lie outside the image: zero-filled)

__device__ __forceinline__ void ct_win_load(const ccl_geom& G, const u64* __restrict__ fb, int y, int x, ct_win& W, u64* lds)
{
    W.y0 = y - CTW_ROWS / 2;
    W.j0 = (x - 96) >> 6;          // x - 64*j0 in [96, 160): at least 96 pixels either side
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CTW_ROWS / 64; k++) {
        const int r = (int)threadIdx.x + 64 * k, yy = W.y0 + r;
        const bool yin = yy >= 0 && yy < G.h;
        const u64* row = fb + (size_t)min(max(yy, 0), G.h - 1) * G.ww;
        u64 v[CTW_WORDS];
#pragma unroll
        for (int c = 0; c < CTW_WORDS; c++) v[c] = row[min(max(W.j0 + c, 0), G.ww - 1)];
#pragma unroll
        for (int c = 0; c < CTW_WORDS; c++) {
            const int j = W.j0 + c;
            lds[r * CTW_WORDS + c] = (yin && j >= 0 && j < G.ww) ? v[c] : 0ull;
        }
    }
    __syncthreads();
}

// 8x8-pixel tile in one register: bit (8*r + c) = pixel (ty + r, tx + c).  The tile must lie inside the window.
struct ct_tile { u64 bits; int ty, tx; };

__device__ __forceinline__ bool ct_win_holds(const ct_win& W, int ty, int tx)
{
    return (unsigned)(ty - W.y0) <= (unsigned)(CTW_ROWS - 8) && (unsigned)(tx - 64 * W.j0) <= (unsigned)(64 * CTW_WORDS - 8);
}
__device__ __forceinline__ void ct_tile_fetch(const u64* lds, const ct_win& W, int ty, int tx, ct_tile& T)
{
    T.ty = ty; T.tx = tx;
    const int ry = ty - W.y0, rx = tx - 64 * W.j0;
    const int j = rx >> 6, sh = rx & 63, j1 = min(j + 1, CTW_WORDS - 1);
    u64 lo[8], hi[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { lo[r] = lds[(ry + r) * CTW_WORDS + j]; hi[r] = lds[(ry + r) * CTW_WORDS + j1]; }
    u64 acc = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        u64 v = lo[r] >> sh;
        v |= sh ? (hi[r] << (64 - sh)) : 0ull;
        acc |= (v & 0xffull) << (8 * r);
    }
    T.bits = acc;
}
// (re)place the tile around (y, x): the pixel sits one step from the trailing edge of the direction it is moving in, so a
// straight run gets ~5 steps out of one fetch
__device__ __forceinline__ void ct_tile_place(const ccl_geom& G, const u64* __restrict__ fb, u64* lds, ct_win& W, int y, int x, int dy, int dx, ct_tile& T)
{
    const int ty = y - (dy > 0 ? 1 : (dy < 0 ? 6 : 3));
    const int tx = x - (dx > 0 ? 1 : (dx < 0 ? 6 : 3));
    if (!ct_win_holds(W, ty, tx)) ct_win_load(G, fb, y, x, W, lds);
    ct_tile_fetch(lds, W, ty, tx, T);
}
__device__ __forceinline__ bool ct_tile_covers(const ct_tile& T, int y, int x)   // 3x3 neighbourhood inside the tile
{
    const unsigned ry = (unsigned)(y - T.ty - 1), rx = (unsigned)(x - T.tx - 1);
    return ry <= 5u && rx <= 5u;
}
// the 8 neighbours of (y, x) as a ring: bit d = neighbour in direction d of {E, NE, N, NW, W, SW, S, SE}
__device__ __forceinline__ u32 ct_ring(const ct_tile& T, int y, int x)
{
    const u64 t = T.bits >> (8 * (y - T.ty - 1) + (x - T.tx - 1));
    const u32 a = (u32)t & 7u, m = (u32)(t >> 8) & 7u, b = (u32)(t >> 16) & 7u;
    // row above: bit0 -> NW(3), bit1 -> N(2), bit2 -> NE(1): a 3-bit reversal, looked up in a nibble table
    return ((0xE6A2C480u >> (4 * a)) & 0xFu) | (m >> 2) | ((m & 1u) << 4) | (b << 5);
}

// Suzuki-Abe trace of one border (imgproc/src/contours.cpp icvFetchContour), counting or writing points.
template <bool WRITE>
__device__ int ct_trace(const ccl_geom& G, const u64* __restrict__ fb, u64* lds, int y0, int x0, bool is_hole, int method, int32_t* __restrict__ pts)
{
    // 8-neighbourhood deltas {E, NE, N, NW, W, SW, S, SE} packed 2 bits each (value + 1): a table indexed at run time would live
    // in memory and cost a load per probe
#define dx8(s) ((int)((0x901Au >> (2 * (s))) & 3u) - 1)
#define dy8(s) ((int)((0xA901u >> (2 * (s))) & 3u) - 1)
    const bool writer = threadIdx.x == 0;
    ct_win W;
    ct_tile T;
    ct_win_load(G, fb, y0, x0, W, lds);
    ct_tile_fetch(lds, W, y0 - 3, x0 - 3, T);
    u32 R = ct_ring(T, y0, x0);
    if (!R) {   // single pixel
        if (WRITE && writer) { pts[0] = x0; pts[1] = y0; }
        return 1;
    }
    // first neighbour clockwise from W (outer border) or from E (hole border): the pixel the border "comes from"
    int s = is_hole ? 0 : 4;
    do { s = (s - 1) & 7; } while (!((R >> s) & 1u));
    const int x1 = x0 + dx8(s), y1 = y0 + dy8(s);   // i1
    int x3 = x0, y3 = y0;
    int prev_s = s ^ 4;
    int n = 0;
    // a border visits a pixel at most once per incoming direction: bound the walk so that a corrupted image cannot
    // keep the wave alive forever
    long long guard = 8ll * G.w * G.h + 16;
    for (; guard > 0; guard--) {
        // first neighbour counter-clockwise after direction s
        const u32 q = (R | (R << 8)) >> (s + 1);
        s = (s + __ffs((int)q)) & 7;
        const int x4 = x3 + dx8(s), y4 = y3 + dy8(s);
        if (s != prev_s || method == 1) {
            if (WRITE && writer) { pts[2 * n] = x3; pts[2 * n + 1] = y3; }
            n++;
            prev_s = s;
        }
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;
        x3 = x4; y3 = y4;
        if (!ct_tile_covers(T, y3, x3)) ct_tile_place(G, fb, lds, W, y3, x3, dy8(s), dx8(s), T);
        R = ct_ring(T, y3, x3);
        s = (s + 4) & 7;
    }
    return n;
#undef dx8
#undef dy8
}

// grid (CT_TRACE_BLOCKS, n), one wave per block: a wave traces borders blockIdx.x, blockIdx.x + gridDim.x, ...
#define CT_TRACE_BLOCKS 256
template <bool WRITE>
__global__ __launch_bounds__(64) void k_ct_trace(const u64* __restrict__ bits, ccl_geom G, const u32* __restrict__ starts,
                                                 const ct_frame_out* __restrict__ info, int method,
                                                 int32_t* __restrict__ counts, uint8_t* __restrict__ is_hole_out,
                                                 const int32_t* __restrict__ offsets, int32_t* __restrict__ points, int max_contours,
                                                 long long max_points)
{
    __shared__ u64 win[CTW_ROWS * CTW_WORDS];
    const int f = blockIdx.y;
    const int K = min(info[f].n_contours, max_contours);
    const u64* fb = bits + (size_t)f * G.h * G.ww;
    for (int rank = blockIdx.x; rank < K; rank += gridDim.x) {
        const u32 st = starts[(size_t)f * max_contours + rank];
        const bool hole = st >> 31;
        const int pix = (int)(st & 0x7fffffffu);
        const int y = pix / G.w, x = pix - y * G.w;
        if (!WRITE) {
            const int cnt = ct_trace<false>(G, fb, win, y, x, hole, method, nullptr);
            if (threadIdx.x == 0) {
                counts[(size_t)f * max_contours + rank] = cnt;
                is_hole_out[(size_t)f * max_contours + rank] = hole ? 1 : 0;
            }
        } else {
            const long long off = offsets[(size_t)f * max_contours + rank];
            const int cnt = counts[(size_t)f * max_contours + rank];
            if (off + cnt <= max_points) ct_trace<true>(G, fb, win, y, x, hole, method, points + 2 * ((size_t)f * max_points + off));
        }
    }
}

// per frame: exclusive scan of counts[0..K) -> offsets, total -> out[f].n_points (one block per frame)
__global__ __launch_bounds__(256) void k_ct_offsets(const int32_t* __restrict__ counts, int32_t* __restrict__ offsets, ct_frame_out* __restrict__ out, int max_contours)
{
    __shared__ u32 wsum[4];
    __shared__ u32 carry;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K = min(out[f].n_contours, max_contours);
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < K; base += 256) {
        const int i = base + tid;
        const u32 v = i < K ? (u32)counts[(size_t)f * max_contours + i] : 0u;
        u32 inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        u32 woff = 0;
        for (int k = 0; k < wv; k++) woff += wsum[k];
        if (i < K) offsets[(size_t)f * max_contours + i] = (int32_t)(carry + woff + inc - v);
        __syncthreads();
        if (tid == 255) carry += woff + inc;
        __syncthreads();
    }
    if (tid == 0) out[f].n_points = (int32_t)carry;
}

size_t vp_contours_ws_bytes(int w, int h, int n)
{
    const size_t nids = vp_ccl_nids(w, h);
    const size_t words = (size_t)n * h * vp_ww(w);
    return 2 * vp_align(nids * 4 * n) + 3 * vp_align(nids / 8 * n) + 2 * vp_align(words * 8) + vp_align(words * 4) + vp_align(sizeof(ct_frame_out) * n) + 4096;
}

// d_counts / d_is_hole: [n][max_contours]; d_points: [n][max_points][2]; d_info: [n] {n_contours, n_points}.  Contours are stored
// in discovery order (raster order of the start pixel); cv2 returns them reversed — the caller reverses.
int vpk_find_contours(vp_ctx* ctx, const u64* d_bits, int w, int h, int n, int mode, int method, int32_t* d_counts, uint8_t* d_is_hole,
                      int32_t* d_offsets, int32_t* d_points, int max_contours, long long max_points, int32_t* d_info)
{
    if (mode != 0 && mode != 1) return vp_fail(ctx, VP_ERR_INVALID, "contour mode");
    if (method != 1 && method != 2) return vp_fail(ctx, VP_ERR_INVALID, "contour approximation");
    ccl_geom Gf, Gb;
    ccl_make_geom(Gf, w, h, VP_CCL_PIXEL, 0, 0);
    ccl_make_geom(Gb, w, h, VP_CCL_PIXEL, 1, 1);
    const size_t nids = Gf.nids;
    const size_t words = (size_t)n * h * Gf.ww;
    u32* fg_parent = (u32*)vp_ws_take(ctx, nids * 4 * n);
    u32* bg_parent = (u32*)vp_ws_take(ctx, nids * 4 * n);
    u32* fg_flags = (u32*)vp_ws_take(ctx, nids / 8 * n);
    u32* bg_flags = (u32*)vp_ws_take(ctx, nids / 8 * n);
    u32* outside = (u32*)vp_ws_take(ctx, nids / 8 * n);
    u64* startmap = (u64*)vp_ws_take(ctx, words * 8);
    u64* holemap = (u64*)vp_ws_take(ctx, words * 8);
    u32* starts = (u32*)vp_ws_take(ctx, (size_t)n * max_contours * 4);
    if (!fg_parent || !bg_parent || !fg_flags || !bg_flags || !outside || !startmap || !holemap || !starts)
        return vp_fail(ctx, VP_ERR_NOMEM, "contour workspace");
    hipStream_t s = ctx->stream;
    ct_frame_out* info = reinterpret_cast<ct_frame_out*>(d_info);
    int rc = ccl_roots(ctx, d_bits, Gf, n, fg_parent, fg_flags);
    if (rc != VP_OK) return rc;
    rc = ccl_roots(ctx, d_bits, Gb, n, bg_parent, bg_flags);
    if (rc != VP_OK) return rc;
    VP_HIP(ctx, hipMemsetAsync(outside, 0, nids / 8 * n, s));
    VP_HIP(ctx, hipMemsetAsync(startmap, 0, words * 8, s));
    VP_HIP(ctx, hipMemsetAsync(holemap, 0, words * 8, s));
    vp_prof_scope ps(ctx, VPK_OTHER);
    hipLaunchKernelGGL(k_ct_outside, dim3((unsigned)((2 * Gb.ww + 2 * h + 255) / 256), (unsigned)n), dim3(256), 0, s, d_bits, Gb, bg_parent, outside);
    hipLaunchKernelGGL(k_ct_seeds, dim3((unsigned)((Gf.nw32 + 255) / 256), (unsigned)n), dim3(256), 0, s, d_bits, Gf, Gb, fg_flags, bg_flags, bg_parent,
                       outside, mode, startmap, holemap);
    hipLaunchKernelGGL(k_ct_rank, dim3((unsigned)n), dim3(1024), 0, s, startmap, holemap, h * Gf.ww, Gf.ww, w, starts, max_contours, info);
    const dim3 tgrid((unsigned)min(max_contours, CT_TRACE_BLOCKS), (unsigned)n);
    hipLaunchKernelGGL((k_ct_trace<false>), tgrid, dim3(64), 0, s, d_bits, Gf, starts, info, method, d_counts, d_is_hole, d_offsets, d_points,
                       max_contours, max_points);
    hipLaunchKernelGGL(k_ct_offsets, dim3((unsigned)n), dim3(256), 0, s, d_counts, d_offsets, info, max_contours);
    hipLaunchKernelGGL((k_ct_trace<true>), tgrid, dim3(64), 0, s, d_bits, Gf, starts, info, method, d_counts, d_is_hole, d_offsets, d_points,
                       max_contours, max_points);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}
