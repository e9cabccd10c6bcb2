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
// written exactly once, by a streaming kernel.
//
//   k_ccl_local    per 32-row strip, in LDS: segments -> compact ids -> union-find (across word boundaries in a
//                  row and with the row above, 8-conn) -> parent[id] = smallest id of the strip-local component;
//                  clears the strip's slice of the root bitmap and marks the strip-local representatives
//   k_ccl_boundary unions between the first row of a strip and the row above it (global CAS union-find);
//                  a root that gets linked loses its bit in the bitmap, so the bitmap ends up = exact roots
//   (k_ccl_init + k_ccl_link: the same in global memory only, fallback for very wide images)
//   k_ccl_rank     per frame: exclusive popcount prefix over the bitmap, nlabels, zero accumulators; extra blocks
//                  of the same launch reduce the background row (label 0) without atomics
//   k_ccl_stats    label = rank(root)+1 per segment -> seglabel[id], wordlabel[word]; per-block LDS aggregation,
//                  then one set of global atomics per (block, label)
//   k_ccl_final    accumulators -> stats (i32 x5) + centroids (f64 x2)
//   k_ccl_write    bits + wordlabel (+ seglabel for words holding several segments) -> int32 label image
//
// Measured and dropped (git history has them; DESIGN.md section 5 lists the numbers): one 1024-thread block per frame running
// boundary -> rank -> stats -> final with block barriers; writing the stats rows from the last block of k_ccl_stats; 2-4 internal
// streams; 64-row strips; a per-strip rank fed by root counters; path compression after the boundary unions; replicated
// accumulators; label write of one half of the batch under the bookkeeping of the other.
#include "vp_internal.h"
#include "vp_ccl_dev.h"
#include <limits.h>
#include <cstdio>
#include <string.h>
#include <cstdlib>
#include <algorithm>
#include <vector>

struct ccl_acc {   // 48 B
    u32 area;
    int minx, miny, maxx, maxy;
    u32 pad;
    u64 sx, sy;
};

struct contrib { u32 area; int minx, maxx, miny, maxy; u32 pad; u64 sx, sy; };

// every entry "empty": what acc_commit expects to add to
__global__ __launch_bounds__(256) void k_ccl3_acc_init(ccl_acc* __restrict__ acc, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        ccl_acc z;
        z.area = 0; z.minx = INT_MAX; z.miny = INT_MAX; z.maxx = INT_MIN; z.maxy = INT_MIN; z.pad = 0; z.sx = 0; z.sy = 0;
        acc[i] = z;
    }
}

#define RK_PARTS 8
#define BG_PARTS 8
#define C2_RC 128              // two-level labelling (vp_ccl2.inl): stride of the per-strip component tables
struct c2_box { int minx, maxx, miny, maxy; };

size_t vp_ccl_nids(int w, int h)
{
    const size_t wb = (size_t)(w + 1) / 2, hb = (size_t)(h + 1) / 2;
    return (2 * hb * wb + 127) / 128 * 128;
}

// strips of the strip-local pass never exceed h / 8 + 1 (ccl_make_geom picks 8, 16 or 32 rows)
static size_t c2_strips_max(int h) { return (size_t)(h + 7) / 8; }
static size_t c3_strips_cap(int h) { return (size_t)(h + 1) / 2 + 2; }   // vp_ccl3.inl: strips of at least 2 rows, + 1 boundary slot
#define C3_STATE_BYTES 48
// which of the two labelling launches takes a strip (vp_ccl3.inl): one byte per (frame, strip)
static size_t c3_items_bytes(int n, int h) { return (size_t)n * c3_strips_cap(h); }

size_t vp_ccl_ws_bytes(int w, int h, int n, int max_labels)
{
    const size_t nids = vp_ccl_nids(w, h);
    const size_t ns = c2_strips_max(h) * (size_t)n;
    return vp_align(nids * 4 * n) * 2 + vp_align(nids / 8 * n) * 2 + vp_align(sizeof(ccl_acc) * (size_t)max_labels * n) +
           vp_align((size_t)n * h * vp_ww(w) * 4) + vp_align(sizeof(contrib) * BG_PARTS * (size_t)n) +
           vp_align(ns * 4) + vp_align(ns * C2_RC * sizeof(contrib)) + vp_align(ns * sizeof(c2_box)) + vp_align(ns * C2_RC * 4) +
           vp_align((size_t)n * 4) + 2 * vp_align(nids / 8 * n) + vp_align((size_t)n * 4) + 256 + vp_align((size_t)n * C3_STATE_BYTES) +
           vp_align((size_t)n * c3_strips_cap(h) * 12) + vp_align((size_t)n * c3_strips_cap(h) * 40) + vp_align(c3_items_bytes(n, h)) + 4096 + 1024;
}

void vp_ccl_ws_carve(vp_ctx* ctx, int w, int h, int n, int max_labels, vp_ccl_ws* out)
{
    const size_t nids = vp_ccl_nids(w, h);
    out->parent = (u32*)vp_ws_take(ctx, nids * 4 * n);
    out->seglabel = (u32*)vp_ws_take(ctx, nids * 4 * n);
    out->flags = (u32*)vp_ws_take(ctx, nids / 8 * n);
    out->prefix = (u32*)vp_ws_take(ctx, nids / 8 * n);
    out->acc = vp_ws_take(ctx, sizeof(ccl_acc) * (size_t)max_labels * n);
    out->wordlabel = (u32*)vp_ws_take(ctx, (size_t)n * h * vp_ww(w) * 4);
    out->bgpart = vp_ws_take(ctx, sizeof(contrib) * BG_PARTS * (size_t)n);
    const size_t ns = c2_strips_max(h) * (size_t)n;
    out->c2_ncomp = (u32*)vp_ws_take(ctx, ns * 4);
    out->c2_recs = vp_ws_take(ctx, ns * C2_RC * sizeof(contrib));
    out->c2_bgbox = vp_ws_take(ctx, ns * sizeof(c2_box));
    out->c2_label = (u32*)vp_ws_take(ctx, ns * C2_RC * 4);
    out->c2_crowded = (u32*)vp_ws_take(ctx, (size_t)n * 4);
    out->c3_child = (u32*)vp_ws_take(ctx, nids / 8 * n);
    out->c3_lroot = (u32*)vp_ws_take(ctx, nids / 8 * n);
    out->c3_clist = (u32*)vp_ws_take(ctx, (size_t)n * 4);
    out->c3_ncrowded = (u32*)vp_ws_take(ctx, 4);
    out->c3_state = vp_ws_take(ctx, (size_t)n * C3_STATE_BYTES);
    out->c3_tot = vp_ws_take(ctx, (size_t)n * c3_strips_cap(h) * 40);              // sizeof(contrib)
    out->c3_barr = (u32*)vp_ws_take(ctx, (size_t)n * c3_strips_cap(h) * 12);   // per frame: boundary arrivals | boundaries done per strip | roots per strip
    out->c3_items = (unsigned char*)vp_ws_take(ctx, c3_items_bytes(n, h));     // per (handed-over frame, strip): 1 = light labelling launch, 2 = heavy
}

bool vp_ccl_ws_ok(const vp_ccl_ws& ws)
{
    return ws.parent && ws.seglabel && ws.flags && ws.prefix && ws.acc && ws.wordlabel && ws.bgpart && ws.c2_ncomp && ws.c2_recs &&
           ws.c2_bgbox && ws.c2_label && ws.c2_crowded && ws.c3_child && ws.c3_lroot && ws.c3_clist && ws.c3_ncrowded && ws.c3_state && ws.c3_barr && ws.c3_tot && ws.c3_items;
}

// ---- whole-image global-memory path (fallback for images too wide for the LDS strip kernel) ---------------
// grid: (ceil(h*ww/256), n); flags zeroed by a memset before k_ccl_init
__global__ __launch_bounds__(256) void k_ccl_init(const u64* __restrict__ bits, ccl_geom G, u32* __restrict__ parent, u32* __restrict__ flags)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= G.h * G.ww) return;
    const int y = idx / G.ww, j = idx - y * G.ww;
    const u64 w = ccl_word(G, bits + (size_t)blockIdx.y * G.h * G.ww, idx, j);
    if (!w) return;
    u32* p = parent + (size_t)blockIdx.y * G.nids;
    u32* f = flags + (size_t)blockIdx.y * G.nw32;
    u64 starts = w & ~(w << 1);
    while (starts) {
        const int s = __ffsll((long long)starts) - 1;
        starts &= starts - 1;
        const u32 id = seg_id(G, y, 64 * j + s);
        p[id] = id;
        atomicOr(f + (id >> 5), 1u << (id & 31));
    }
}

__global__ __launch_bounds__(256) void k_ccl_link(const u64* __restrict__ bits, ccl_geom G, u32* __restrict__ parent, u32* __restrict__ flags)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= G.h * G.ww) return;
    const u64* fb = bits + (size_t)blockIdx.y * G.h * G.ww;
    const int y = idx / G.ww, j = idx - y * G.ww;
    const u64 w = ccl_word(G, fb, idx, j);
    if (!w) return;
    global_link_word(fb, G, parent + (size_t)blockIdx.y * G.nids, flags + (size_t)blockIdx.y * G.nw32, y, j, idx, w, true, true);
}

// One block per (frame, strip of CL_ROWS rows).
// dynamic LDS: lbits[nw] u64 | wbase[nw + 2] u32 | lparent[cap] | lgid[cap] | lmin[cap] (cap <= nw + 2: lmin reuses wbase)
__global__ __launch_bounds__(256) void k_ccl_local(const u64* __restrict__ bits, ccl_geom G, u32* __restrict__ parent,
                                                   u32* __restrict__ flags, int strips, int cap, const u32* __restrict__ only,
                                                   u32* __restrict__ zero_too)   // nullable: bitmap with the layout of `flags`, cleared here
{
    extern __shared__ __attribute__((aligned(16))) u64 cl_lds[];
    __shared__ u32 wsum[4];
    __shared__ u32 total_s;
    const int ww = G.ww;
    const int frame = blockIdx.x / strips, strip = blockIdx.x - frame * strips;
    if (only && !only[frame]) return;   // fallback launch of the two-level path: this frame was resolved there
    const int y0 = strip * G.rows;
    const int nrows = min(G.rows, G.h - y0);
    const int nwmax = G.rows * ww;
    u64* lbits = cl_lds;
    u32* wbase = reinterpret_cast<u32*>(cl_lds + nwmax);
    u32* lparent = wbase + (nwmax + 2);
    const u64* fb = bits + (size_t)frame * G.h * ww;
    CL_FOR_WORDS(r, j, i) lbits[i] = ccl_word(G, fb, (y0 + r) * ww + j, j);
    __syncthreads();
    ccl_local_strip(G, lbits, wbase, lparent, lparent + cap, cap <= nwmax + 2 ? wbase : lparent + 2 * cap, wsum, &total_s, y0, nrows, strip, strips, fb,
                    parent + (size_t)frame * G.nids, flags + (size_t)frame * G.nw32, (u32)cap, zero_too ? zero_too + (size_t)frame * G.nw32 : nullptr);
}

// vertical unions across strip boundaries: grid (strips - 1, n), block = 64 threads over the words of the row
__global__ __launch_bounds__(64) void k_ccl_boundary(const u64* __restrict__ bits, ccl_geom G, u32* __restrict__ parent, u32* __restrict__ flags,
                                                     const u32* __restrict__ only)
{
    if (only && !only[blockIdx.y]) return;
    const int y = (blockIdx.x + 1) * G.rows;
    const u64* fb = bits + (size_t)blockIdx.y * G.h * G.ww;
    u32* p = parent + (size_t)blockIdx.y * G.nids;
    u32* f = flags + (size_t)blockIdx.y * G.nw32;
    for (int j = threadIdx.x; j < G.ww; j += 64) {
        const int idx = y * G.ww + j;
        const u64 w = ccl_word(G, fb, idx, j);
        if (w) global_link_word(fb, G, p, f, y, j, idx, w, false, true);
    }
}

// ---- ranks + background -----------------------------------------------------------------------------------

__device__ __forceinline__ u32 sum_bitpos(u64 z)  // sum of the positions of the set bits
{
    return (u32)__popcll(z & 0xAAAAAAAAAAAAAAAAull) + ((u32)__popcll(z & 0xCCCCCCCCCCCCCCCCull) << 1) +
           ((u32)__popcll(z & 0xF0F0F0F0F0F0F0F0ull) << 2) + ((u32)__popcll(z & 0xFF00FF00FF00FF00ull) << 3) +
           ((u32)__popcll(z & 0xFFFF0000FFFF0000ull) << 4) + ((u32)__popcll(z & 0xFFFFFFFF00000000ull) << 5);
}

__device__ __forceinline__ void contrib_zero(contrib& c)
{
    c.area = 0; c.sx = 0; c.sy = 0; c.minx = INT_MAX; c.maxx = INT_MIN; c.miny = INT_MAX; c.maxy = INT_MIN; c.pad = 0;
}
__device__ __forceinline__ void contrib_merge(contrib& c, const contrib& o)
{
    c.area += o.area; c.sx += o.sx; c.sy += o.sy;
    c.minx = min(c.minx, o.minx); c.maxx = max(c.maxx, o.maxx); c.miny = min(c.miny, o.miny); c.maxy = max(c.maxy, o.maxy);
}
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

// grid (RK_PARTS + BG_PARTS, n), 256 threads.
// blocks [0, RK_PARTS): exclusive popcount prefix over the root bitmap of one frame — part p first counts the bits
//   of parts < p (the bitmap is 130 KB per 1080p frame, L2-resident), then scans its own slice; the last part also
//   publishes nlabels and zeroes the accumulators.
// blocks [RK_PARTS, RK_PARTS + BG_PARTS): background pixels of a slice of the frame reduced to one record.
__global__ __launch_bounds__(256) void k_ccl_rank(ccl_geom G, const u32* __restrict__ flags, u32* __restrict__ prefix,
                                                  int32_t* __restrict__ nlabels, ccl_acc* __restrict__ acc, int max_labels,
                                                  const u64* __restrict__ bits, contrib* __restrict__ bgpart, const u32* __restrict__ only)
{
    if (only && !only[blockIdx.y]) return;
    __shared__ u32 wsum[4];
    __shared__ u32 wsum2[4];
    __shared__ u32 bcast;
    __shared__ contrib part[4];
    const int f = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    if (blockIdx.x >= RK_PARTS) {
        const int bp = blockIdx.x - RK_PARTS;
        const int rper = (G.h + BG_PARTS - 1) / BG_PARTS;
        const int r0 = bp * rper, r1 = min(r0 + rper, G.h);
        const u64* fb = bits + (size_t)f * G.h * G.ww;
        const u64 lastmask = (G.w & 63) ? ((1ull << (G.w & 63)) - 1ull) : ~0ull;
        contrib c;
        contrib_zero(c);
        for (int y = r0 + (tid >> 5); y < r1; y += 8)
            for (int j = tid & 31; j < G.ww; j += 32) {
                u64 z = ~fb[(size_t)y * G.ww + j];
                if (j == G.ww - 1) z &= lastmask;
                if (!z) continue;
                const u32 cnt = (u32)__popcll(z);
                c.area += cnt;
                c.sx += (u64)cnt * (u64)(64 * j) + sum_bitpos(z);
                c.sy += (u64)cnt * (u64)y;
                c.minx = min(c.minx, 64 * j + (__ffsll((long long)z) - 1));
                c.maxx = max(c.maxx, 64 * j + 63 - __clzll(z));
                c.miny = min(c.miny, y);
                c.maxy = max(c.maxy, y);
            }
        wave_combine(c);
        if (lane == 0) part[wv] = c;
        __syncthreads();
        if (tid == 0) {
            for (int k = 1; k < 4; k++) contrib_merge(c, part[k]);
            bgpart[(size_t)f * BG_PARTS + bp] = c;
        }
        return;
    }
    const int partno = blockIdx.x;
    const u32 nq = G.nw32 / 4;                               // uint4 words per frame
    const u32 qper = (nq + RK_PARTS - 1) / RK_PARTS;
    const u32 q0 = min((u32)partno * qper, nq), q1 = min(q0 + qper, nq);
    const uint4* fl = reinterpret_cast<const uint4*>(flags + (size_t)f * G.nw32);
    uint4* pf = reinterpret_cast<uint4*>(prefix + (size_t)f * G.nw32);
    // own slice: thread = contiguous run of uint4s, loaded once and kept in registers (RK_OWN covers 1080p..4K)
    constexpr int RK_OWN = 16;
    const u32 per = (q1 - q0 + 255) / 256;
    const u32 lo = min(q0 + (u32)tid * per, q1), hi = min(lo + per, q1);
    uint4 own[RK_OWN];
    u32 cnt = 0;
#pragma unroll
    for (int k = 0; k < RK_OWN; k++) {
        own[k] = make_uint4(0, 0, 0, 0);
        if (lo + k < hi) own[k] = fl[lo + k];
    }
    // bits before this part: independent loads, issued in batches
    u32 before = 0;
#pragma unroll 8
    for (u32 q = tid; q < q0; q += 256) { const uint4 v = fl[q]; before += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }
#pragma unroll
    for (int k = 0; k < RK_OWN; k++) cnt += __popc(own[k].x) + __popc(own[k].y) + __popc(own[k].z) + __popc(own[k].w);
    for (u32 q = lo + RK_OWN; q < hi; q++) { const uint4 v = fl[q]; cnt += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }   // slices beyond RK_OWN (huge images)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) before += __shfl_xor(before, d);
    u32 inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    if (lane == 0) wsum[wv] = before;
    if (lane == 63) wsum2[wv] = inc;
    __syncthreads();
    const u32 base = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    u32 woff = 0;
    for (int k = 0; k < wv; k++) woff += wsum2[k];
    u32 run = base + woff + inc - cnt;
#pragma unroll
    for (int k = 0; k < RK_OWN; k++) {
        if (lo + k < hi) {
            const uint4 v = own[k];
            uint4 o;
            o.x = run; run += __popc(v.x);
            o.y = run; run += __popc(v.y);
            o.z = run; run += __popc(v.z);
            o.w = run; run += __popc(v.w);
            pf[lo + k] = o;
        }
    }
    for (u32 q = lo + RK_OWN; q < hi; q++) {
        const uint4 v = fl[q];
        uint4 o;
        o.x = run; run += __popc(v.x);
        o.y = run; run += __popc(v.y);
        o.z = run; run += __popc(v.z);
        o.w = run; run += __popc(v.w);
        pf[q] = o;
    }
    if (partno == RK_PARTS - 1) {
        if (tid == 255) bcast = run;   // last thread's running count = all bits of the frame
        __syncthreads();
        const int nl = (int)bcast + 1;
        if (tid == 0 && nlabels) nlabels[f] = nl;
        ccl_acc* a = acc + (size_t)f * max_labels;
        const int nz = min(nl, max_labels);
        for (int i = tid; i < nz; i += 256) {
            ccl_acc z;
            z.area = 0; z.minx = INT_MAX; z.miny = INT_MAX; z.maxx = INT_MIN; z.maxy = INT_MIN; z.pad = 0; z.sx = 0; z.sy = 0;
            a[i] = z;
        }
    }
}

// ---- per-label statistics ----------------------------------------------------------------------------------

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

// grid: (ceil(h*ww/256), n).  Per-segment contributions are first combined per label in a small LDS hash table
// (a block covers ~8 rows, so it sees a handful of labels), then each occupied slot is flushed with one set of
// global atomics.
#define ST_SLOTS 64
struct st_table {
    u32 label[ST_SLOTS];
    u32 area[ST_SLOTS];
    int minx[ST_SLOTS], maxx[ST_SLOTS], miny[ST_SLOTS], maxy[ST_SLOTS];
    u64 sx[ST_SLOTS], sy[ST_SLOTS];
};
__global__ __launch_bounds__(256) void k_ccl_stats(const u64* __restrict__ bits, ccl_geom G, const u32* __restrict__ parent,
                                                   const u32* __restrict__ flags, const u32* __restrict__ prefix,
                                                   u32* __restrict__ seglabel, u32* __restrict__ wordlabel,
                                                   ccl_acc* __restrict__ acc, int max_labels, const u32* __restrict__ only)
{
    if (only && !only[blockIdx.y]) return;
    __shared__ st_table T;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < G.h * G.ww;
    const int f = blockIdx.y;
    const u64 w = live ? bits[(size_t)f * G.h * G.ww + idx] : 0ull;
    const bool any_fg = __syncthreads_or(w != 0ull);   // block-uniform
    ccl_acc* a = acc + (size_t)f * max_labels;
    if (any_fg) {
        if (threadIdx.x < ST_SLOTS) {
            const int i = threadIdx.x;
            T.label[i] = 0; T.area[i] = 0; T.sx[i] = 0; T.sy[i] = 0;
            T.minx[i] = INT_MAX; T.miny[i] = INT_MAX; T.maxx[i] = INT_MIN; T.maxy[i] = INT_MIN;
        }
        __syncthreads();
        // label of a segment + its contribution; the first segment of every word goes through a wave-level shortcut: when all
        // the first segments of a wave carry the same label (the inside of a blob, a full mask) their records are combined with
        // shuffles and added once, instead of 64 lanes queueing on the same LDS slot
        const int y = idx / G.ww, j = idx - y * G.ww;
        const u32* p = parent + (size_t)f * G.nids;
        const u32* fl = flags + (size_t)f * G.nw32;
        const u32* pf = prefix + (size_t)f * G.nw32;
        u32* sl = seglabel + (size_t)f * G.nids;
        auto table_add = [&](u32 label, const contrib& c) {
            u32 slot = (label * 2654435761u) >> 26;   // 6 bits
            for (int probe = 0; probe < 8; probe++, slot = (slot + 1) & (ST_SLOTS - 1)) {
                const u32 cur = atomicCAS(&T.label[slot], 0u, label);
                if (cur == 0u || cur == label) {
                    atomicAdd(&T.area[slot], c.area);
                    atomicAdd((unsigned long long*)&T.sx[slot], (unsigned long long)c.sx);
                    atomicAdd((unsigned long long*)&T.sy[slot], (unsigned long long)c.sy);
                    atomicMin(&T.minx[slot], c.minx);
                    atomicMax(&T.maxx[slot], c.maxx);
                    atomicMin(&T.miny[slot], c.miny);
                    atomicMax(&T.maxy[slot], c.maxy);
                    return;
                }
            }
            acc_commit(a + label, c);   // crowded block (noise): straight to global memory
        };
        const bool multi = nstarts(w) > 1u;   // the label-write kernel reads seglabel only for words holding several segments
        auto lookup = [&](int s) -> u32 {     // label of the segment starting at bit s: four dependent, uncoalesced reads
            const u32 id = seg_id(G, y, 64 * j + s);
            u32 r = id;
            for (u32 q = p[r]; q != r; q = p[r]) r = q;   // read-only walk: strip root, then across strips
            const u32 label = pf[r >> 5] + (u32)__popc(fl[r >> 5] & ((1u << (r & 31)) - 1u)) + 1u;
            if (multi) sl[id] = label;
            return label;
        };
        auto record = [&](int s, int e, contrib& c) {
            const u32 len = (u32)(e - s + 1);
            const int xs = 64 * j + s, xe = 64 * j + e;
            c.area = len; c.sx = (u64)len * (u64)(xs + xe) / 2ull; c.sy = (u64)len * (u64)y;
            c.minx = xs; c.maxx = xe; c.miny = c.maxy = y;
        };
        u64 rem = w;
        contrib c0;
        contrib_zero(c0);
        u32 lab0 = 0;
        {
            // A first segment that continues the only segment of the word to its left (same row, same wave) has that segment's
            // label: only the first word of such a chain looks its label up, the others take it through a wave scan.  Inside
            // blobs that removes most of the uncoalesced reads this kernel waits for.
            const int lane = threadIdx.x & 63;
            const u64 wprev = __shfl_up(w, 1);
            bool take = (w & 1ull) && lane > 0 && j > 0 && (wprev >> 63) && nstarts(wprev) == 1u;
            int s0 = 0, e0 = 0;
            u32 lab = 0;
            if (w) {
                s0 = __ffsll((long long)rem) - 1;
                e0 = run_end(rem, s0);
                rem &= ~bit_range(s0, e0);
                if (!take) lab = lookup(s0);
            }
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u32 pv = __shfl_up(lab, d);
                const int pt = __shfl_up((int)take, d);
                if (lane >= d && take) { lab = pv; take = pt != 0; }
            }
            if (w) {
                if (multi) sl[seg_id(G, y, 64 * j + s0)] = lab;
                wordlabel[(size_t)f * G.h * G.ww + idx] = lab;
                if (lab < (u32)max_labels) { lab0 = lab; record(s0, e0, c0); }
            }
        }
        const unsigned long long act = __ballot(lab0 != 0u);
        if (act) {
            const int lead = __ffsll((long long)act) - 1;
            const u32 ref = __shfl(lab0, lead);
            if (__popcll(act) >= 8 && __all(lab0 == 0u || lab0 == ref)) {
                wave_combine(c0);
                if ((int)(threadIdx.x & 63) == lead) table_add(ref, c0);
            } else if (lab0) {
                table_add(lab0, c0);
            }
        }
        while (rem) {
            const int s = __ffsll((long long)rem) - 1;
            const int e = run_end(rem, s);
            rem &= ~bit_range(s, e);
            contrib c;
            const u32 label = lookup(s);
            if (label < (u32)max_labels) { record(s, e, c); table_add(label, c); }
        }
        __syncthreads();
        if (threadIdx.x < ST_SLOTS && T.label[threadIdx.x]) {
            const int i = threadIdx.x;
            contrib c;
            c.area = T.area[i]; c.sx = T.sx[i]; c.sy = T.sy[i]; c.minx = T.minx[i]; c.maxx = T.maxx[i]; c.miny = T.miny[i]; c.maxy = T.maxy[i];
            acc_commit(a + T.label[i], c);
        }
    }
}

// grid: (ceil(max_labels/256), n)
__global__ __launch_bounds__(256) void k_ccl_final(const ccl_acc* __restrict__ acc, const contrib* __restrict__ bgpart,
                                                   const int32_t* __restrict__ nlabels, int max_labels, int32_t* __restrict__ stats,
                                                   double* __restrict__ cent, const u32* __restrict__ only)
{
    if (only && !only[blockIdx.y]) return;
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= max_labels) return;
    const int f = blockIdx.y;
    const size_t o = (size_t)f * max_labels + l;
    if (l < nlabels[f]) {
        ccl_acc a = acc[o];
        if (l == 0) {   // background: merge the per-slice records
            contrib c;
            contrib_zero(c);
            for (int k = 0; k < BG_PARTS; k++) contrib_merge(c, bgpart[(size_t)f * BG_PARTS + k]);
            a.area = c.area; a.minx = c.minx; a.maxx = c.maxx; a.miny = c.miny; a.maxy = c.maxy; a.sx = c.sx; a.sy = c.sy;
        }
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

// ---- label image -------------------------------------------------------------------------------------------
// One lane = 4 px = one 16-B store.  A block streams WR_ROWS consecutive rows; a thread owns WR_K groups and runs
// in phases — all bit-word + word-label loads, then the stores — so that no load waits behind a store (vmcnt
// counts stores too on CDNA4).  A word with a single segment (the common case inside blobs) takes its label from
// the dense wordlabel array; only words holding several segments go to the sparse seglabel array.
// launch bound 8 waves per SIMD: the compiler otherwise takes 74 VGPRs (6 waves); at 62 VGPRs and full occupancy the kernel streams at
// 5.8 instead of 5.2 TB/s (188 instead of 208 us per 128 frames)
#define WR_ROWS 4
#define WR_K 8
__global__ __launch_bounds__(256, 8) void k_ccl_write(const u64* __restrict__ bits, ccl_geom G, const u32* __restrict__ seglabel,
                                                   const u32* __restrict__ wordlabel, int32_t* __restrict__ labels, u32 total_rows,
                                                   u32 gpr, u32 gpr_magic)
{
    const u32 row0 = blockIdx.x * WR_ROWS;
    const u32 nrows = min((u32)WR_ROWS, total_rows - row0);
    const u32 ngroups = nrows * gpr;
    const bool vec = (G.w & 3) == 0 && ((((uintptr_t)labels) & 15) == 0);
    const u64* brow0 = bits + (size_t)row0 * G.ww;
    const u32* wrow0 = wordlabel + (size_t)row0 * G.ww;
    int32_t* lrow0 = labels + (size_t)row0 * G.w;
    const u32 f0 = row0 / (u32)G.h;
    const u32 y0 = row0 - f0 * (u32)G.h;
    for (u32 qb = 0; qb < ngroups; qb += 256 * WR_K) {
        u64 w[WR_K];
        u32 wl[WR_K], rl[WR_K], g[WR_K];
        bool live[WR_K];
#pragma unroll
        for (int k = 0; k < WR_K; k++) {
            const u32 q = qb + (u32)k * 256 + threadIdx.x;
            live[k] = q < ngroups;
            const u32 qq = live[k] ? q : 0;
            rl[k] = gpr == 1 ? qq : __umulhi(qq, gpr_magic);   // qq / gpr (exact for qq < 2^16 * gpr)
            g[k] = qq - rl[k] * gpr;
            const u32 wi = rl[k] * (u32)G.ww + (g[k] >> 4);
            w[k] = brow0[wi];
            wl[k] = wrow0[wi];
        }
        u32 la[WR_K], lb[WR_K], m1[WR_K], m2[WR_K];
        bool sparse = false;
#pragma unroll
        for (int k = 0; k < WR_K; k++) {
            const u32 nib = (u32)(w[k] >> ((g[k] * 4) & 63)) & 0xfu;
            m1[k] = nib;
            m2[k] = 0;
            la[k] = wl[k];
            lb[k] = 0;
            sparse |= nib && (nstarts(w[k]) > 1);
        }
        if (__any(sparse)) {   // wave-uniform, rare: some word here holds more than one segment
#pragma unroll
            for (int k = 0; k < WR_K; k++) {
                const u32 nib = m1[k];
                if (!nib || nstarts(w[k]) <= 1) continue;
                const int x0 = (int)g[k] * 4, sub = x0 & 63, j = x0 >> 6;
                u32 y = y0 + rl[k], f = f0;
                while (y >= (u32)G.h) { y -= (u32)G.h; f++; }
                const u32* sl = seglabel + (size_t)f * G.nids;
                // first run of the nibble, and what is left after it (at most one more run)
                const int tz = __ffs((int)nib) - 1;
                const u32 t = nib >> tz;
                const int runlen = __ffs((int)~t) - 1;
                m1[k] = ((1u << runlen) - 1u) << tz;
                m2[k] = nib & ~m1[k];
                la[k] = sl[seg_id(G, (int)y, 64 * j + run_start(w[k], sub + tz))];
                if (m2[k]) lb[k] = sl[seg_id(G, (int)y, 64 * j + sub + (__ffs((int)m2[k]) - 1))];
            }
        }
#pragma unroll
        for (int k = 0; k < WR_K; k++) {
            if (!live[k]) continue;
            const int x0 = (int)g[k] * 4;
            int vv[4];
#pragma unroll
            for (int b = 0; b < 4; b++) vv[b] = ((m1[k] >> b) & 1u) ? (int)la[k] : (((m2[k] >> b) & 1u) ? (int)lb[k] : 0);
            int32_t* d = lrow0 + (size_t)rl[k] * G.w + x0;
            if (vec && x0 + 4 <= G.w) {
                vp_store16(d, (u32)vv[0], (u32)vv[1], (u32)vv[2], (u32)vv[3]);
            } else {
                for (int b = 0; b < 4; b++)
                    if (x0 + b < G.w) d[b] = vv[b];
            }
        }
    }
}

static void ccl_make_geom(ccl_geom& G, int w, int h, int numbering, int invert, int conn4)
{
    G.w = w; G.h = h; G.ww = vp_ww(w); G.wb = (w + 1) / 2; G.numbering = numbering;
    G.nids = (u32)vp_ccl_nids(w, h);
    G.nw32 = G.nids / 32;
    G.invert = invert; G.conn4 = conn4;
    // Strip height of the strip-local pass.  A block's time grows faster than its strip (64 rows: 75 us, 32: 53 us at 1080p), and
    // wide rows make strips heavy: at 4K (60 words per row) 16-row strips take k_ccl_local from 116 to 44 us for +4 us of
    // boundary unions; at 1080p the two cancel.  16 rows need ceil(w/2) even (bitmap slices must not share a word).
    G.rows = (G.ww > 32 && (G.wb % 2) == 0) ? 16 : 32;
    if (const char* e = getenv("VP_CL_ROWS")) { const int r = atoi(e); if ((r == 8 || r == 16 || r == 32) && ((u32)r * (u32)G.wb) % 32u == 0) G.rows = r; }
}

#include "vp_ccl3.inl"
#include "vp_ccl2.inl"

// LDS of the strip-local kernels: lbits | wbase | lparent | lgid (| lmin when it cannot share wbase's words)
static size_t ccl_local_lds(const ccl_geom& G, size_t& cap)
{
    const size_t nwmax = (size_t)G.rows * G.ww;
    // Foreground: room for one segment per word of the strip (a full mask) + 2, the size of the wbase array whose LDS lmin then
    // reuses; denser strips (speckle) take the global fallback.  Background pass of the contour code: every empty word is a
    // segment and every foreground edge adds one, so it gets its own lmin array and 1024 more entries.
    static const char* cap_env = getenv("VP_CL_CAP");
    cap = G.invert ? nwmax + 1024 : nwmax + 2;
    if (cap_env && (size_t)atoi(cap_env) >= 64 && (size_t)atoi(cap_env) < cap) cap = (size_t)atoi(cap_env);
    size_t lds_local = nwmax * 8 + (nwmax + 2) * 4 + (cap <= nwmax + 2 ? 2 : 3) * cap * 4;
    if (lds_local > 64 * 1024 && G.invert) { cap = nwmax + 2; lds_local = nwmax * 8 + (nwmax + 2) * 4 + 2 * cap * 4; }
    return lds_local;
}

// union-find phase only: parent[] (every segment points at a smaller id of its component, roots at themselves)
// and the exact root bitmap in flags[]; `only` (nullable): per-frame switch, frames whose entry is 0 are left alone
// zero_too (nullable): a bitmap laid out like `flags` that the caller wants cleared (the strip kernel clears its slice of the root
// bitmap anyway; the contour code's "reaches the frame" bitmap rides along instead of costing a memset launch)
static int ccl_roots(vp_ctx* ctx, const u64* d_bits, const ccl_geom& G, int n, u32* parent, u32* flags, const u32* only = nullptr,
                     u32* zero_too = nullptr)
{
    const int h = G.h;
    const dim3 wgrid((unsigned)((h * G.ww + 255) / 256), (unsigned)n);
    hipStream_t s = ctx->stream;
    const int strips = (h + G.rows - 1) / G.rows;
    size_t cap;
    const size_t lds_local = ccl_local_lds(G, cap);
    if (lds_local <= 64 * 1024) {
        { vp_prof_scope ps(ctx, VPK_CCL_LOCAL); hipLaunchKernelGGL(k_ccl_local, dim3((unsigned)((size_t)n * strips)), dim3(256), lds_local, s, d_bits, G, parent, flags, strips, (int)cap, only, zero_too); }
        if (strips > 1) { vp_prof_scope ps(ctx, VPK_CCL_BOUNDARY); hipLaunchKernelGGL(k_ccl_boundary, dim3((unsigned)(strips - 1), (unsigned)n), dim3(64), 0, s, d_bits, G, parent, flags, only); }
    } else {
        if (zero_too) { vp_prof_scope ps(ctx, VPK_MEMSET); VP_HIP(ctx, hipMemsetAsync(zero_too, 0, (size_t)G.nw32 * 4 * n, s)); }
        { vp_prof_scope ps(ctx, VPK_MEMSET); VP_HIP(ctx, hipMemsetAsync(flags, 0, (size_t)G.nw32 * 4 * n, s)); }
        { vp_prof_scope ps(ctx, VPK_CCL_BOUNDARY); hipLaunchKernelGGL(k_ccl_init, wgrid, dim3(256), 0, s, d_bits, G, parent, flags); }
        { vp_prof_scope ps(ctx, VPK_CCL_LOCAL); hipLaunchKernelGGL(k_ccl_link, wgrid, dim3(256), 0, s, d_bits, G, parent, flags); }
    }
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// the one-level kernels after the union-find: ranks, per-label statistics, stats rows; on ctx->stream
static int ccl_one_level_tail(vp_ctx* ctx, const u64* d_bits, const ccl_geom& G, int n, const vp_ccl_ws& ws, int32_t* d_stats,
                              double* d_centroids, int max_labels, int32_t* d_nlabels, const u32* only)
{
    const dim3 wgrid((unsigned)((G.h * G.ww + 255) / 256), (unsigned)n);
    hipStream_t s = ctx->stream;
    { vp_prof_scope ps(ctx, VPK_CCL_RANK); hipLaunchKernelGGL(k_ccl_rank, dim3(RK_PARTS + BG_PARTS, (unsigned)n), dim3(256), 0, s, G, ws.flags, ws.prefix, d_nlabels, (ccl_acc*)ws.acc, max_labels, d_bits, (contrib*)ws.bgpart, only); }
    { vp_prof_scope ps(ctx, VPK_CCL_STATS); hipLaunchKernelGGL(k_ccl_stats, wgrid, dim3(256), 0, s, d_bits, G, ws.parent, ws.flags, ws.prefix, ws.seglabel, ws.wordlabel, (ccl_acc*)ws.acc, max_labels, only); }
    if (d_stats || d_centroids) {
        vp_prof_scope ps(ctx, VPK_CCL_FINAL);
        hipLaunchKernelGGL(k_ccl_final, dim3((unsigned)((max_labels + 255) / 256), (unsigned)n), dim3(256), 0, s, (const ccl_acc*)ws.acc,
                           (const contrib*)ws.bgpart, d_nlabels, max_labels, d_stats, d_centroids, only);
    }
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

int vpk_ccl(vp_ctx* ctx, const u64* d_bits, int w, int h, int n, int numbering, const vp_ccl_ws& ws, int32_t* d_labels,
            int32_t* d_stats, double* d_centroids, int max_labels, int32_t* d_nlabels)
{
    if (numbering != VP_CCL_BLOCK2X2 && numbering != VP_CCL_PIXEL) return vp_fail(ctx, VP_ERR_INVALID, "numbering");
    if (max_labels < 1) return vp_fail(ctx, VP_ERR_INVALID, "max_labels");
    ccl_geom G;
    ccl_make_geom(G, w, h, numbering, 0, 0);
    hipStream_t s = ctx->stream;
    const int strips = (h + G.rows - 1) / G.rows;
    const u32 gpr = (u32)((w + 3) / 4);
    const u32 magic = (u32)((0x100000000ull + gpr - 1) / gpr);

    // ---- two-level path (vp_ccl2.inl) ----------------------------------------------------------------------------------
    const size_t nwmax = (size_t)G.rows * G.ww;
    const size_t cap2 = nwmax + 2;
    const size_t rc = 96;   // components per strip with LDS accumulators (44 B each)
    size_t tail_words = std::max(rc * 11, cap2);    // union queue, then root -> list place (cap words), then the accumulators
    tail_words += tail_words & 1;
    const size_t list_words = nwmax + (nwmax + 1) / 2 + ((nwmax + (nwmax + 1) / 2) & 1);   // wbase + word list, padded to 8 bytes
    const size_t lds2 = nwmax * 8 + (list_words + cap2 + tail_words) * 4;                   // 1080p: 21.8 KB, seven blocks per CU
    const bool two_level = ctx->ccl_levels == 2 && lds2 <= 64 * 1024 && G.ww <= 64 && G.rows <= CL_ROWS && strips <= C2_MAXSTRIPS &&
                           (G.rows % WR_ROWS) == 0 && (G.rows % 8) == 0 && (size_t)strips <= c2_strips_max(h);
    if (two_level) {
        const int mcap = (ctx->ccl_mcap >= 0 && ctx->ccl_mcap < C2_MCAP) ? ctx->ccl_mcap : C2_MCAP;
        // frames the merge hands over go to the crowded-frame kernels of vp_ccl3.inl when the geometry suits them (it does for every
        // frame up to 8192 px wide), otherwise to the one-level kernels
        static const u32 c3_max_ids = getenv("VP_C3_IDS") ? (u32)atoi(getenv("VP_C3_IDS")) : (u32)C3_IDS;
        c3_plan P3 = c3_make_plan(G, std::min<u32>(c3_max_ids, C3_IDS));
        const bool c3_tall = P3.ids > 8192;           // taller strips: twice the threads per block (one word per thread still)
        static const bool c3_off = getenv("VP_CCL3") && atoi(getenv("VP_CCL3")) == 0;
        size_t lds3a = 0, lds3b = 0, lds3c = 0;
        if (P3.ok && !c3_off && (size_t)P3.strips + 1 <= c3_strips_cap(h) && sizeof(c3_state) == C3_STATE_BYTES) {
            lds3a = c3_link_lds(G, P3);
            lds3b = c3_label_lds(G, P3, 0, C3_ACC);
            lds3c = c3_label_lds(G, P3, C3_LIGHT_ROOTS, C3_LIGHT_ACC);
            // (grow-only; kernels accept more dynamic LDS than the 64 KB default once told so - per device, hence kept in the context)
            const void* fa_ = c3_tall ? (const void*)k_ccl3_link<2 * C3_LINK_THREADS> : (const void*)k_ccl3_link<C3_LINK_THREADS>;
            const void* fb_ = c3_tall ? (const void*)k_ccl3_label<2 * C3_LABEL_THREADS, 0, C3_ACC> : (const void*)k_ccl3_label<C3_LABEL_THREADS, 0, C3_ACC>;
            const void* fc_ = (const void*)k_ccl3_label<C3_LABEL_THREADS, C3_LIGHT_ROOTS, C3_LIGHT_ACC>;
            size_t& ra = ctx->c3_lds_set[c3_tall ? 2 : 0]; size_t& rb = ctx->c3_lds_set[c3_tall ? 3 : 1]; size_t& rc_ = ctx->c3_lds_set[4];
            if (lds3a > ra) { if (hipFuncSetAttribute(fa_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3a) == hipSuccess) ra = lds3a; else { (void)hipGetLastError(); P3.ok = 0; } }
            if (P3.ok && lds3b > rb) { if (hipFuncSetAttribute(fb_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3b) == hipSuccess) rb = lds3b; else { (void)hipGetLastError(); P3.ok = 0; } }
            if (P3.ok && lds3c > rc_) { if (hipFuncSetAttribute(fc_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3c) == hipSuccess) rc_ = lds3c; else { (void)hipGetLastError(); P3.ok = 0; } }
        } else {
            P3.ok = 0;
        }
        const int c3_strips = P3.ok ? P3.strips : 0;
        if (P3.ok && getenv("VP_CCL3_OCC")) {   // diagnosis: blocks per CU the runtime grants the crowded-frame kernels
            int oa = 0, ob = 0, oc = 0;
            if (c3_tall) {
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&oa, k_ccl3_link<2 * C3_LINK_THREADS>, 2 * C3_LINK_THREADS, lds3a);
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&ob, k_ccl3_label<2 * C3_LABEL_THREADS, 0, C3_ACC>, 2 * C3_LABEL_THREADS, lds3b);
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&oc, k_ccl3_label<C3_LABEL_THREADS, C3_LIGHT_ROOTS, C3_LIGHT_ACC>, C3_LABEL_THREADS, lds3c);
            } else {
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&oa, k_ccl3_link<C3_LINK_THREADS>, C3_LINK_THREADS, lds3a);
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&ob, k_ccl3_label<C3_LABEL_THREADS, 0, C3_ACC>, C3_LABEL_THREADS, lds3b);
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&oc, k_ccl3_label<C3_LABEL_THREADS, C3_LIGHT_ROOTS, C3_LIGHT_ACC>, C3_LABEL_THREADS, lds3c);
            }
            hipFuncAttributes fb2, fc2;
            hipFuncGetAttributes(&fb2, c3_tall ? (const void*)k_ccl3_label<2 * C3_LABEL_THREADS, 0, C3_ACC> : (const void*)k_ccl3_label<C3_LABEL_THREADS, 0, C3_ACC>);
            hipFuncGetAttributes(&fc2, (const void*)k_ccl3_label<C3_LABEL_THREADS, C3_LIGHT_ROOTS, C3_LIGHT_ACC>);
            fprintf(stderr, "ccl3 occupancy: link %d blocks/CU (LDS %zu), label heavy %d blocks/CU (LDS %zu + %zu static, %d regs), light %d blocks/CU (LDS %zu + %zu static, %d regs), CUs %d\n",
                    oa, lds3a, ob, lds3b, fb2.sharedSizeBytes, fb2.numRegs, oc, lds3c, fc2.sharedSizeBytes, fc2.numRegs, ctx->num_cu);
        }
        { vp_prof_scope ps(ctx, VPK_CCL2_LOCAL);
          hipLaunchKernelGGL(k_ccl2_local, dim3((unsigned)((size_t)n * strips)), dim3(256), lds2, s, d_bits, G, strips, (int)cap2, (int)rc, (int)tail_words, ws.c2_ncomp,
                             (contrib*)ws.c2_recs, (c2_box*)ws.c2_bgbox, ws.wordlabel, ws.seglabel, ws.c3_ncrowded); }
        { vp_prof_scope ps(ctx, VPK_CCL2_MERGE);
          hipLaunchKernelGGL(k_ccl2_merge, dim3((unsigned)n), dim3(C2_THREADS), 0, s, d_bits, G, strips, mcap, ws.c2_ncomp, (const contrib*)ws.c2_recs,
                             (const c2_box*)ws.c2_bgbox, ws.wordlabel, ws.seglabel, ws.c2_label, ws.c2_crowded, d_nlabels, d_stats, d_centroids, max_labels,
                             ws.c3_ncrowded, ws.c3_clist, (c3_state*)ws.c3_state, ws.c3_barr, c3_strips); }
        VP_HIP(ctx, hipGetLastError());
        static const int c3_dbg = getenv("VP_CCL3_DBG") ? atoi(getenv("VP_CCL3_DBG")) : 0;   // timing experiments only: parts of the kernels skipped
        static const int c3_dry = getenv("VP_CCL3_DRY") ? atoi(getenv("VP_CCL3_DRY")) : 0;   // experiments: 1 no launches, 2 no link, 3 no label
        // Accumulators of the components that span strips, indexed by label: a set of the context's own that is all "empty" between
        // calls - k_ccl3_rows resets exactly the entries it reads - so that no launch writes max_labels entries per frame (268 MB per 128
        // frames with 65,536 labels allowed: a quarter of what k_ccl3_link stored).  Sub-batches on several streams would share it: they
        // keep the per-call workspace and its clearing.
        ccl_acc* acc3 = (ccl_acc*)ws.acc;
        bool acc3_own = false;
        if (P3.ok && c3_dry != 1 && ctx->chain_streams == 1) {
            const size_t need = sizeof(ccl_acc) * (size_t)max_labels * (size_t)n;
            if (need > ctx->c3_acc_bytes) {
                VP_HIP(ctx, hipStreamSynchronize(s));
                if (ctx->c3_acc) { (void)hipFree(ctx->c3_acc); ctx->c3_acc = nullptr; ctx->c3_acc_bytes = 0; }
                if (hipMalloc(&ctx->c3_acc, need) == hipSuccess) { ctx->c3_acc_bytes = need; ctx->c3_acc_dirty = 1; }
                else (void)hipGetLastError();                  // no memory for it: the workspace's set, cleared per call
            }
            if (ctx->c3_acc && need <= ctx->c3_acc_bytes) {
                if (ctx->c3_acc_dirty) {
                    const size_t entries = ctx->c3_acc_bytes / sizeof(ccl_acc);
                    hipLaunchKernelGGL(k_ccl3_acc_init, dim3((unsigned)std::min<size_t>((entries + 255) / 256, 65535)), dim3(256), 0, s, (ccl_acc*)ctx->c3_acc, entries);
                }
                ctx->c3_acc_dirty = 1;                         // until k_ccl3_rows of this call has been queued
                acc3 = (ccl_acc*)ctx->c3_acc;
                acc3_own = true;
            }
        }
        if (P3.ok && c3_dry != 1) {
            // two launches that read the list of handed-over frames and leave at once when it is empty (the usual case)
            if (c3_dry != 2) { vp_prof_scope ps(ctx, VPK_CCL_LOCAL);
              static const int lgrid = getenv("VP_C3_LGRID") ? atoi(getenv("VP_C3_LGRID")) : 16;
              if (c3_tall) hipLaunchKernelGGL(k_ccl3_link<2 * C3_LINK_THREADS>, dim3((unsigned)(ctx->num_cu * lgrid)), dim3(2 * C3_LINK_THREADS), lds3a, s, d_bits, G, P3, ws.c3_ncrowded, ws.c3_clist, ws.parent,
                                 ws.flags, ws.c3_child, ws.c3_lroot, ws.seglabel, acc3, max_labels, acc3_own ? 0 : 1, c3_dbg);
              else hipLaunchKernelGGL(k_ccl3_link<C3_LINK_THREADS>, dim3((unsigned)(ctx->num_cu * lgrid)), dim3(C3_LINK_THREADS), lds3a, s, d_bits, G, P3, ws.c3_ncrowded, ws.c3_clist, ws.parent,
                                 ws.flags, ws.c3_child, ws.c3_lroot, ws.seglabel, acc3, max_labels, acc3_own ? 0 : 1, c3_dbg); }
            { vp_prof_scope ps(ctx, VPK_CCL_BOUNDARY);
              const size_t span = (size_t)2 * G.wb + 4;     // u16 roots of the two rows (row pairs) that meet: span entries each
              static const int bgrid = getenv("VP_C3_BGRID") ? atoi(getenv("VP_C3_BGRID")) : 64;   // blocks per CU in the grid: items differ a lot in cost, the dispatcher balances
              hipLaunchKernelGGL(k_ccl3_bound, dim3((unsigned)(ctx->num_cu * bgrid)), dim3(256), span * 4 + 16, s, d_bits, G, P3, ws.c3_ncrowded, ws.c3_clist, ws.parent, ws.flags,
                                 ws.c3_child, ws.seglabel, c3_dbg); }
            { vp_prof_scope ps(ctx, VPK_CCL_RANK);
              hipLaunchKernelGGL(k_ccl3_rank, dim3((unsigned)(ctx->num_cu * 4)), dim3(256), 0, s, G, P3, ws.c3_ncrowded, ws.c3_clist, ws.flags, ws.prefix, ws.c3_barr, ws.c3_lroot, ws.parent,
                                 ws.c3_items); }
            if (c3_dry != 3) { vp_prof_scope ps(ctx, VPK_CCL_STATS);
              // two instantiations, each taking the strips k_ccl3_rank marked for it: LIGHT first - two blocks per CU - then HEAVY
              static const int agrid = getenv("VP_C3_AGRID") ? atoi(getenv("VP_C3_AGRID")) : 2;
              static const int agrid_light = getenv("VP_C3_AGRID_LIGHT") ? atoi(getenv("VP_C3_AGRID_LIGHT")) : 4;
#define C3_LABEL_ARGS d_bits, G, P3, ws.c3_ncrowded, ws.c3_items, ws.c3_clist, ws.parent, ws.flags, ws.c3_child, ws.prefix, ws.c3_lroot, ws.seglabel, ws.c3_barr, \
                             (c3_state*)ws.c3_state, (contrib*)ws.c3_tot, (int)c3_strips_cap(h), d_nlabels, acc3, max_labels, d_labels, d_stats, d_centroids, c3_dbg
              // (the light instantiation always with 512 threads: at ~120 registers a CU holds 16 waves, i.e. two blocks of eight)
              hipLaunchKernelGGL((k_ccl3_label<C3_LABEL_THREADS, C3_LIGHT_ROOTS, C3_LIGHT_ACC>), dim3((unsigned)(ctx->num_cu * agrid_light)), dim3(C3_LABEL_THREADS), lds3c, s, C3_LABEL_ARGS);
              if (c3_tall) hipLaunchKernelGGL((k_ccl3_label<2 * C3_LABEL_THREADS, 0, C3_ACC>), dim3((unsigned)(ctx->num_cu * agrid)), dim3(2 * C3_LABEL_THREADS), lds3b, s, C3_LABEL_ARGS);
              else hipLaunchKernelGGL((k_ccl3_label<C3_LABEL_THREADS, 0, C3_ACC>), dim3((unsigned)(ctx->num_cu * agrid)), dim3(C3_LABEL_THREADS), lds3b, s, C3_LABEL_ARGS);
#undef C3_LABEL_ARGS
            }
            if (d_stats || d_centroids || acc3_own) {
                vp_prof_scope ps(ctx, VPK_CCL_FINAL);
                hipLaunchKernelGGL(k_ccl3_rows, dim3((unsigned)(ctx->num_cu * 4)), dim3(256), 0, s, G, P3, ws.c3_ncrowded, ws.c3_clist, ws.flags, ws.c3_child, ws.prefix,
                                   ws.c3_barr, (const c3_state*)ws.c3_state, (const contrib*)ws.c3_tot, (int)c3_strips_cap(h), acc3, max_labels, d_stats, d_centroids,
                                   acc3_own ? 1 : 0);
                if (acc3_own && c3_dry == 0 && c3_dbg == 0) ctx->c3_acc_dirty = 0;   // every entry the labelling launch touched has been handed back clean
            }
            VP_HIP(ctx, hipGetLastError());
        } else if (!P3.ok) {
            // the one-level kernels, launched whatever the frames hold (the flags live on the device); they leave at once for frames the merge resolved
            int rc2 = ccl_roots(ctx, d_bits, G, n, ws.parent, ws.flags, ws.c2_crowded);
            if (rc2 == VP_OK) rc2 = ccl_one_level_tail(ctx, d_bits, G, n, ws, d_stats, d_centroids, max_labels, d_nlabels, ws.c2_crowded);
            if (rc2 != VP_OK) return rc2;
        }
        if (c3_dry == 4) {   // diagnosis: which frames were handed over, and why
            std::vector<u32> cr(n), nc((size_t)n * strips);
            hipStreamSynchronize(s);
            hipMemcpy(cr.data(), ws.c2_crowded, (size_t)n * 4, hipMemcpyDeviceToHost);
            hipMemcpy(nc.data(), ws.c2_ncomp, (size_t)n * strips * 4, hipMemcpyDeviceToHost);
            for (int f = 0; f < n; f++)
                if (cr[f]) { fprintf(stderr, "crowded frame %d of %d:", f, n); for (int k = 0; k < strips; k++) fprintf(stderr, " %u", nc[(size_t)f * strips + k]); fprintf(stderr, "\n"); }
        }
        if (d_labels) {
            vp_prof_scope ps(ctx, VPK_CCL2_WRITE);
            const dim3 wr_grid((unsigned)((h + WR_ROWS - 1) / WR_ROWS), (unsigned)n);
            hipLaunchKernelGGL(k_ccl2_write<0>, wr_grid, dim3(256), 0, s, d_bits, G, strips, (int)rc, ws.seglabel, ws.wordlabel, ws.c2_label, ws.c2_crowded, d_labels, gpr, magic,
                               P3.ok ? 1 : 0);
        }
        VP_HIP(ctx, hipGetLastError());
        return VP_OK;
    }

    // ---- one-level path -------------------------------------------------------------------------------------------------
    int rc1 = ccl_roots(ctx, d_bits, G, n, ws.parent, ws.flags);
    if (rc1 != VP_OK) return rc1;
    rc1 = ccl_one_level_tail(ctx, d_bits, G, n, ws, d_stats, d_centroids, max_labels, d_nlabels, nullptr);
    if (rc1 != VP_OK) return rc1;
    if (d_labels) {
        vp_prof_scope ps(ctx, VPK_CCL_WRITE);
        const u32 total_rows = (u32)((size_t)n * h);
        hipLaunchKernelGGL(k_ccl_write, dim3((total_rows + WR_ROWS - 1) / WR_ROWS), dim3(256), 0, s, d_bits, G, ws.seglabel, ws.wordlabel, d_labels,
                           total_rows, gpr, magic);
    }
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

#ifdef VP_PROBE
// measurement builds: per probe point, the ticks summed over the blocks that reached point 15 (ran to the end) in out[k*16 + i],
// their number in out[k*16 + 15]; slots are cleared afterwards
extern "C" int vp_debug_probe(double* out32)
{
    static std::vector<unsigned int> h((size_t)2 * C2_PROBE_BLOCKS * 16);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_c2_probe), h.size() * 4) != hipSuccess) return -1;
    for (int i = 0; i < 32; i++) out32[i] = 0;
    for (int k = 0; k < 2; k++)
        for (int b = 0; b < C2_PROBE_BLOCKS; b++) {
            const unsigned int* r = &h[((size_t)k * C2_PROBE_BLOCKS + b) * 16];
            if (r[15] == 0xffffffffu || r[14] != 0x600dc0deu) continue;
            for (int i = 0; i < 14; i++) out32[k * 16 + i] += r[i];
            out32[k * 16 + 15] += 1;
        }
    std::fill(h.begin(), h.end(), 0u);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_c2_probe), h.data(), h.size() * 4) == hipSuccess ? 0 : -1;
}
#endif

#ifdef VP_PROBE
// crowded-frame kernels (link, bound, label): out[k * 16 + i] = ticks (10 ns) of phase i summed over blocks and items, out[k * 16 + 15] = blocks that ran
extern "C" int vp_debug_probe3(double* out32 /* 48 */)
{
    static std::vector<unsigned long long> h((size_t)3 * 2048 * 16);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_c3_probe), h.size() * 8) != hipSuccess) return -1;
    for (int i = 0; i < 48; i++) out32[i] = 0;
    for (int k = 0; k < 3; k++)
        for (int b = 0; b < 2048; b++) {
            const unsigned long long* r = &h[((size_t)k * 2048 + b) * 16];
            unsigned long long t = 0;
            for (int i = 0; i < 15; i++) { out32[k * 16 + i] += (double)r[i]; t += r[i]; }
            if (t) out32[k * 16 + 15] += 1;
        }
    std::fill(h.begin(), h.end(), 0ull);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_c3_probe), h.data(), h.size() * 8) == hipSuccess ? 0 : -1;
}
#endif

#include "vp_contours.inl"
