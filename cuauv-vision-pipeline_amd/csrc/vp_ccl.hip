// 8-connected component labelling with statistics on bit-packed masks, for gfx950.
//
// North-star replacement for the cv2.findContours stage of utils/feature.py:5-40 /
// modules/red_buoy.py:38: output equals cv2.connectedComponentsWithStats(mask, 8, CV_32S)
// (labels, stats rows [left, top, width, height, area], centroids; row 0 = background).
//
// The union-find runs on *word segments* — maximal runs of 1-bits inside one 64-bit word of a
// row — not on pixels.  A segment's id encodes where it starts, in an order that matches cv2's
// numbering: for VP_CCL_BLOCK2X2, id = 2*((y/2)*Wb + x/2) + (y&1), i.e. the raster index of the
// aligned 2x2 block holding the segment's first pixel.  Linking always points the larger id at
// the smaller, so a component's root is its smallest id and cv2's label for the component is
// simply the rank of that root among all roots.  Ranks come from a bitmap of roots + popcount
// prefix — no sort.  The label image (4 B/px, the dominant HBM traffic of the whole chain) is
// written exactly once, by a streaming kernel that looks up one label per segment.
//
//   k_ccl_init     parent[id] = id for every segment            (sparse)
//   k_ccl_link     unions: across word boundaries in a row, and with the row above (8-conn)
//   k_ccl_flatten  parent[id] = root(id); mark roots in the bitmap
//   k_ccl_rank     per frame: exclusive popcount prefix over the bitmap, nlabels, zero accumulators
//   k_ccl_stats    seglabel[id] = rank+1; wave-aggregated atomics into per-label accumulators
//   k_ccl_final    accumulators -> stats (i32 x5) + centroids (f64 x2)
//   k_ccl_write    bits + seglabel -> int32 label image (coalesced 16-B stores)
#include "vp_internal.h"
#include <limits.h>

struct ccl_geom {
    int w, h, ww, wb, numbering;
    u32 nids;   // multiple of 32
    u32 nw32;   // nids / 32
};

struct ccl_acc {   // 48 B
    u32 area;
    int minx, miny, maxx, maxy;
    u32 pad;
    u64 sx, sy;
};

size_t vp_ccl_nids(int w, int h)
{
    const size_t wb = (size_t)(w + 1) / 2, hb = (size_t)(h + 1) / 2;
    return (2 * hb * wb + 31) / 32 * 32;
}

size_t vp_ccl_ws_bytes(int w, int h, int n, int max_labels)
{
    const size_t nids = vp_ccl_nids(w, h);
    return vp_align(nids * 4 * n) * 2 + vp_align(nids / 8 * n) * 2 + vp_align(sizeof(ccl_acc) * (size_t)max_labels * n) + 1024;
}

void vp_ccl_ws_carve(vp_ctx* ctx, int w, int h, int n, int max_labels, vp_ccl_ws* out)
{
    const size_t nids = vp_ccl_nids(w, h);
    out->parent = (u32*)vp_ws_take(ctx, nids * 4 * n);
    out->seglabel = (u32*)vp_ws_take(ctx, nids * 4 * n);
    out->flags = (u32*)vp_ws_take(ctx, nids / 8 * n);
    out->prefix = (u32*)vp_ws_take(ctx, nids / 8 * n);
    out->acc = vp_ws_take(ctx, sizeof(ccl_acc) * (size_t)max_labels * n);
}

__device__ __forceinline__ u32 seg_id(const ccl_geom& G, int y, int x)
{
    if (G.numbering == VP_CCL_BLOCK2X2) return (((u32)(y >> 1) * (u32)G.wb + (u32)(x >> 1)) << 1) | (u32)(y & 1);
    return (u32)y * (u32)G.wb + (u32)(x >> 1);
}

__device__ __forceinline__ u64 bit_range(int s, int e)  // bits s..e inclusive
{
    const int len = e - s + 1;
    return (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << s;
}
// start / end (inclusive) of the run of 1s of `w` that contains set bit b
__device__ __forceinline__ int run_start(u64 w, int b)
{
    const u64 t = ~w & ((1ull << b) - 1ull);
    return t ? 64 - __clzll(t) : 0;
}
__device__ __forceinline__ int run_end(u64 w, int b)
{
    const u64 t = ~(w >> b);  // bit 0 is clear
    return t ? b + (__ffsll((long long)t) - 1) - 1 : 63;
}

__device__ __forceinline__ u32 ld_rlx(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rlx(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// find with path halving.  Parent ids strictly decrease towards the root, links are only ever
// added at roots (CAS below), so a stale or half-compressed pointer still names an ancestor.
__device__ __forceinline__ u32 uf_find_halve(u32* p, u32 x)
{
    for (;;) {
        const u32 q = ld_rlx(p + x);
        if (q == x) return x;
        const u32 g = ld_rlx(p + q);
        if (g == q) return q;
        st_rlx(p + x, g);
        x = g;
    }
}
__device__ __forceinline__ u32 uf_find_ro(const u32* p, u32 x)
{
    for (;;) {
        const u32 q = ld_rlx(p + x);
        if (q == x) return x;
        x = q;
    }
}
__device__ __forceinline__ void uf_unite(u32* p, u32 a, u32 b)
{
    for (;;) {
        a = uf_find_halve(p, a);
        b = uf_find_halve(p, b);
        if (a == b) return;
        if (a < b) { const u32 t = a; a = b; b = t; }
        const u32 old = atomicCAS(p + a, a, b);   // link the larger root under the smaller
        if (old == a) return;
        a = old;
    }
}

// grid: (ceil(h*ww/256), n)
__global__ __launch_bounds__(256) void k_ccl_init(const u64* __restrict__ bits, ccl_geom G, u32* __restrict__ parent)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= G.h * G.ww) return;
    const u64 w = bits[(size_t)blockIdx.y * G.h * G.ww + idx];
    if (!w) return;
    const int y = idx / G.ww, j = idx - y * G.ww;
    u32* p = parent + (size_t)blockIdx.y * G.nids;
    u64 starts = w & ~(w << 1);
    while (starts) {
        const int s = __ffsll((long long)starts) - 1;
        starts &= starts - 1;
        const u32 id = seg_id(G, y, 64 * j + s);
        p[id] = id;
    }
}

__global__ __launch_bounds__(256) void k_ccl_link(const u64* __restrict__ bits, ccl_geom G, u32* __restrict__ parent)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= G.h * G.ww) return;
    const u64* fb = bits + (size_t)blockIdx.y * G.h * G.ww;
    const u64 w = fb[idx];
    if (!w) return;
    const int y = idx / G.ww, j = idx - y * G.ww;
    u32* p = parent + (size_t)blockIdx.y * G.nids;
    if ((w & 1ull) && j > 0) {
        const u64 prev = fb[idx - 1];
        if (prev >> 63) uf_unite(p, seg_id(G, y, 64 * j), seg_id(G, y, 64 * (j - 1) + run_start(prev, 63)));
    }
    if (y == 0) return;
    const u64 um = fb[idx - G.ww];
    const u64 ul = j > 0 ? fb[idx - G.ww - 1] : 0ull;
    const u64 ur = j + 1 < G.ww ? fb[idx - G.ww + 1] : 0ull;
    if (!(um | (ul >> 63) | (ur & 1ull))) return;
    u64 rem = w;
    while (rem) {
        const int s = __ffsll((long long)rem) - 1;
        const int e = run_end(rem, s);
        const u64 S = bit_range(s, e);
        rem &= ~S;
        const u32 me = seg_id(G, y, 64 * j + s);
        u64 c = um & (S | (S << 1) | (S >> 1));
        while (c) {
            const int b = __ffsll((long long)c) - 1;
            const int st = run_start(um, b), en = run_end(um, b);
            uf_unite(p, me, seg_id(G, y - 1, 64 * j + st));
            c &= ~bit_range(st, en);
        }
        if ((S & 1ull) && (ul >> 63)) uf_unite(p, me, seg_id(G, y - 1, 64 * (j - 1) + run_start(ul, 63)));
        if ((S >> 63) && (ur & 1ull)) uf_unite(p, me, seg_id(G, y - 1, 64 * (j + 1)));
    }
}

__global__ __launch_bounds__(256) void k_ccl_flatten(const u64* __restrict__ bits, ccl_geom G, u32* __restrict__ parent,
                                                     u32* __restrict__ flags)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= G.h * G.ww) return;
    const u64 w = bits[(size_t)blockIdx.y * G.h * G.ww + idx];
    if (!w) return;
    const int y = idx / G.ww, j = idx - y * G.ww;
    u32* p = parent + (size_t)blockIdx.y * G.nids;
    u32* f = flags + (size_t)blockIdx.y * G.nw32;
    u64 starts = w & ~(w << 1);
    while (starts) {
        const int s = __ffsll((long long)starts) - 1;
        starts &= starts - 1;
        const u32 id = seg_id(G, y, 64 * j + s);
        const u32 r = uf_find_ro(p, id);   // read-only walk: nobody else writes p[id] in this kernel
        if (r == id) atomicOr(f + (r >> 5), 1u << (r & 31));
        else st_rlx(p + id, r);
    }
}

// one block of 1024 threads per frame
__global__ __launch_bounds__(1024) void k_ccl_rank(ccl_geom G, const u32* __restrict__ flags, u32* __restrict__ prefix,
                                                   int32_t* __restrict__ nlabels, ccl_acc* __restrict__ acc, int max_labels)
{
    __shared__ u32 wsum[16];
    __shared__ u32 total;
    const int f = blockIdx.x;
    const u32* fl = flags + (size_t)f * G.nw32;
    u32* pf = prefix + (size_t)f * G.nw32;
    const u32 chunk = (G.nw32 + 1023u) / 1024u;
    const u32 lo = threadIdx.x * chunk;
    const u32 hi = min(lo + chunk, G.nw32);
    u32 s = 0;
    for (u32 i = lo; i < hi; i++) s += __popc(fl[i]);
    // block exclusive scan of s
    u32 inc = s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (int k = 0; k < 16; k++) { const u32 t = wsum[k]; wsum[k] = run; run += t; }
        total = run;
    }
    __syncthreads();
    u32 run = wsum[wv] + inc - s;
    for (u32 i = lo; i < hi; i++) { pf[i] = run; run += __popc(fl[i]); }
    const int nl = (int)total + 1;
    if (threadIdx.x == 0 && nlabels) nlabels[f] = nl;
    ccl_acc* a = acc + (size_t)f * max_labels;
    const int nz = min(nl, max_labels);
    for (int i = threadIdx.x; i < nz; i += 1024) {
        ccl_acc z;
        z.area = 0; z.minx = INT_MAX; z.miny = INT_MAX; z.maxx = INT_MIN; z.maxy = INT_MIN; z.pad = 0; z.sx = 0; z.sy = 0;
        a[i] = z;
    }
}

__device__ __forceinline__ u32 sum_bitpos(u64 z)  // sum of the positions of the set bits
{
    return (u32)__popcll(z & 0xAAAAAAAAAAAAAAAAull) + ((u32)__popcll(z & 0xCCCCCCCCCCCCCCCCull) << 1) +
           ((u32)__popcll(z & 0xF0F0F0F0F0F0F0F0ull) << 2) + ((u32)__popcll(z & 0xFF00FF00FF00FF00ull) << 3) +
           ((u32)__popcll(z & 0xFFFF0000FFFF0000ull) << 4) + ((u32)__popcll(z & 0xFFFFFFFF00000000ull) << 5);
}

struct contrib { u32 area; u64 sx, sy; int minx, maxx, miny, maxy; };

__device__ __forceinline__ void wave_combine(contrib& c)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        c.area += __shfl_xor(c.area, d);
        c.sx += __shfl_xor(c.sx, d);
        c.sy += __shfl_xor(c.sy, d);
        c.minx = min(c.minx, __shfl_xor(c.minx, d));
        c.maxx = max(c.maxx, __shfl_xor(c.maxx, d));
        c.miny = min(c.miny, __shfl_xor(c.miny, d));
        c.maxy = max(c.maxy, __shfl_xor(c.maxy, d));
    }
}

__device__ __forceinline__ void acc_commit(ccl_acc* a, const contrib& c)
{
    atomicAdd(&a->area, c.area);
    atomicAdd((unsigned long long*)&a->sx, (unsigned long long)c.sx);
    atomicAdd((unsigned long long*)&a->sy, (unsigned long long)c.sy);
    atomicMin(&a->minx, c.minx);
    atomicMax(&a->maxx, c.maxx);
    atomicMin(&a->miny, c.miny);
    atomicMax(&a->maxy, c.maxy);
}

// grid: (ceil(h*ww/256), n).  No early return: the whole wave takes part in the reductions.
__global__ __launch_bounds__(256) void k_ccl_stats(const u64* __restrict__ bits, ccl_geom G, const u32* __restrict__ parent,
                                                   const u32* __restrict__ flags, const u32* __restrict__ prefix,
                                                   u32* __restrict__ seglabel, ccl_acc* __restrict__ acc, int max_labels)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < G.h * G.ww;
    const int f = blockIdx.y;
    const u64 w = live ? bits[(size_t)f * G.h * G.ww + idx] : 0ull;
    const int y = live ? idx / G.ww : 0, j = live ? idx - y * G.ww : 0;
    const u32* p = parent + (size_t)f * G.nids;
    const u32* fl = flags + (size_t)f * G.nw32;
    const u32* pf = prefix + (size_t)f * G.nw32;
    u32* sl = seglabel + (size_t)f * G.nids;
    ccl_acc* a = acc + (size_t)f * max_labels;
    const int lane = threadIdx.x & 63;

    // background contribution of this word
    u64 valid = ~0ull;
    if (j == G.ww - 1 && (G.w & 63)) valid = (1ull << (G.w & 63)) - 1ull;
    u64 z = live ? (~w & valid) : 0ull;
    u64 rem = w;
    bool first = true;
    for (;;) {
        // next contribution of this lane: background first, then one segment per round
        bool has = false;
        u32 label = 0;
        contrib c;
        c.area = 0; c.sx = 0; c.sy = 0; c.minx = INT_MAX; c.maxx = INT_MIN; c.miny = INT_MAX; c.maxy = INT_MIN;
        if (first) {
            if (z) {
                has = true;
                const u32 cnt = (u32)__popcll(z);
                c.area = cnt;
                c.sx = (u64)cnt * (u64)(64 * j) + sum_bitpos(z);
                c.sy = (u64)cnt * (u64)y;
                c.minx = 64 * j + (__ffsll((long long)z) - 1);
                c.maxx = 64 * j + 63 - __clzll(z);
                c.miny = c.maxy = y;
            }
        } else if (rem) {
            has = true;
            const int s = __ffsll((long long)rem) - 1;
            const int e = run_end(rem, s);
            rem &= ~bit_range(s, e);
            const u32 id = seg_id(G, y, 64 * j + s);
            u32 r = ld_rlx(p + id);
            label = pf[r >> 5] + (u32)__popc(fl[r >> 5] & ((1u << (r & 31)) - 1u)) + 1u;
            sl[id] = label;
            const u32 len = (u32)(e - s + 1);
            const u32 xs = (u32)(64 * j + s), xe = (u32)(64 * j + e);
            c.area = len;
            c.sx = (u64)len * (u64)(xs + xe) / 2ull;
            c.sy = (u64)len * (u64)y;
            c.minx = (int)xs; c.maxx = (int)xe; c.miny = c.maxy = y;
        }
        first = false;
        u64 active = __ballot(has);
        if (!active && !__any(rem != 0)) break;
        while (active) {
            const int leader = __ffsll((long long)active) - 1;
            const u32 lab = __shfl(label, leader);
            const bool mine = has && label == lab;
            const u64 grp = __ballot(mine);
            contrib g = c;
            if (!mine) { g.area = 0; g.sx = 0; g.sy = 0; g.minx = INT_MAX; g.maxx = INT_MIN; g.miny = INT_MAX; g.maxy = INT_MIN; }
            if (__popcll(grp) > 1) wave_combine(g);
            if (lane == leader && lab < (u32)max_labels) acc_commit(a + lab, g);
            active &= ~grp;
        }
    }
}

// grid: (ceil(max_labels/256), n)
__global__ __launch_bounds__(256) void k_ccl_final(const ccl_acc* __restrict__ acc, const int32_t* __restrict__ nlabels,
                                                   int max_labels, int32_t* __restrict__ stats, double* __restrict__ cent)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= max_labels) return;
    const int f = blockIdx.y;
    const size_t o = (size_t)f * max_labels + l;
    if (l < nlabels[f]) {
        const ccl_acc a = acc[o];
        if (stats) {
            int32_t* s = stats + o * 5;
            s[0] = a.minx;
            s[1] = a.miny;
            s[2] = (int32_t)((u32)a.maxx - (u32)a.minx + 1u);
            s[3] = (int32_t)((u32)a.maxy - (u32)a.miny + 1u);
            s[4] = (int32_t)a.area;
        }
        if (cent) {
            const double area = (double)a.area;
            cent[o * 2] = (double)a.sx / area;
            cent[o * 2 + 1] = (double)a.sy / area;
        }
    } else {
        if (stats) { int32_t* s = stats + o * 5; s[0] = s[1] = s[2] = s[3] = s[4] = 0; }
        if (cent) { cent[o * 2] = 0.0; cent[o * 2 + 1] = 0.0; }
    }
}

// grid: (n*h, ceil(ceil(w/4)/256)).  One lane = 4 px = one 16-B store.
__global__ __launch_bounds__(256) void k_ccl_write(const u64* __restrict__ bits, ccl_geom G, const u32* __restrict__ seglabel,
                                                   int32_t* __restrict__ labels)
{
    const int g = blockIdx.y * 256 + threadIdx.x;
    const int x0 = g * 4;
    if (x0 >= G.w) return;
    const u32 row = blockIdx.x;            // frame*h + y
    const u32 f = row / (u32)G.h;
    const int y = (int)(row - f * (u32)G.h);
    const int j = x0 >> 6, sub = x0 & 63;
    const u64 w = bits[(size_t)row * G.ww + j];
    const u32 nib = (u32)(w >> sub) & 0xfu;
    int v[4] = {0, 0, 0, 0};
    if (nib) {
        const u32* sl = seglabel + (size_t)f * G.nids;
        int prev = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((nib >> k) & 1u) {
                if (k == 0 || !((nib >> (k - 1)) & 1u)) prev = (int)sl[seg_id(G, y, 64 * j + run_start(w, sub + k))];
                v[k] = prev;
            }
        }
    }
    int32_t* drow = labels + (size_t)row * G.w;
    if (x0 + 4 <= G.w && ((((uintptr_t)(drow + x0)) & 15) == 0)) {
        *reinterpret_cast<int4*>(drow + x0) = make_int4(v[0], v[1], v[2], v[3]);
    } else {
        for (int k = 0; k < 4; k++)
            if (x0 + k < G.w) drow[x0 + k] = v[k];
    }
}

int vpk_ccl(vp_ctx* ctx, const u64* d_bits, int w, int h, int n, int numbering, const vp_ccl_ws& ws, int32_t* d_labels,
            int32_t* d_stats, double* d_centroids, int max_labels, int32_t* d_nlabels)
{
    if (numbering != VP_CCL_BLOCK2X2 && numbering != VP_CCL_PIXEL) return vp_fail(ctx, VP_ERR_INVALID, "numbering");
    if (max_labels < 1) return vp_fail(ctx, VP_ERR_INVALID, "max_labels");
    ccl_geom G;
    G.w = w; G.h = h; G.ww = vp_ww(w); G.wb = (w + 1) / 2; G.numbering = numbering;
    G.nids = (u32)vp_ccl_nids(w, h);
    G.nw32 = G.nids / 32;
    const dim3 wgrid((unsigned)((h * G.ww + 255) / 256), (unsigned)n);
    hipStream_t s = ctx->stream;
    { vp_prof_scope ps(ctx, VPK_OTHER); VP_HIP(ctx, hipMemsetAsync(ws.flags, 0, (size_t)G.nw32 * 4 * n, s)); }
    { vp_prof_scope ps(ctx, VPK_CCL_INIT); hipLaunchKernelGGL(k_ccl_init, wgrid, dim3(256), 0, s, d_bits, G, ws.parent); }
    { vp_prof_scope ps(ctx, VPK_CCL_LINK); hipLaunchKernelGGL(k_ccl_link, wgrid, dim3(256), 0, s, d_bits, G, ws.parent); }
    { vp_prof_scope ps(ctx, VPK_CCL_FLATTEN); hipLaunchKernelGGL(k_ccl_flatten, wgrid, dim3(256), 0, s, d_bits, G, ws.parent, ws.flags); }
    { vp_prof_scope ps(ctx, VPK_CCL_RANK); hipLaunchKernelGGL(k_ccl_rank, dim3((unsigned)n), dim3(1024), 0, s, G, ws.flags, ws.prefix, d_nlabels, (ccl_acc*)ws.acc, max_labels); }
    { vp_prof_scope ps(ctx, VPK_CCL_STATS); hipLaunchKernelGGL(k_ccl_stats, wgrid, dim3(256), 0, s, d_bits, G, ws.parent, ws.flags, ws.prefix, ws.seglabel, (ccl_acc*)ws.acc, max_labels); }
    if (d_stats || d_centroids) {
        vp_prof_scope ps(ctx, VPK_CCL_FINAL);
        hipLaunchKernelGGL(k_ccl_final, dim3((unsigned)((max_labels + 255) / 256), (unsigned)n), dim3(256), 0, s, (const ccl_acc*)ws.acc,
                           d_nlabels, max_labels, d_stats, d_centroids);
    }
    if (d_labels) {
        vp_prof_scope ps(ctx, VPK_CCL_WRITE);
        const dim3 lgrid((unsigned)((size_t)n * h), (unsigned)(((w + 3) / 4 + 255) / 256));
        hipLaunchKernelGGL(k_ccl_write, lgrid, dim3(256), 0, s, d_bits, G, ws.seglabel, d_labels);
    }
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}
