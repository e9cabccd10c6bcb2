// Device-side building blocks of the labelling kernels, shared by vp_ccl.hip and the fused morphology + strip-local
// labelling kernel in vp_morph.hip.  See vp_ccl.hip for the algorithm.
#pragma once
#include "vp_internal.h"

struct ccl_geom {
    int w, h, ww, wb, numbering;
    u32 nids;   // multiple of 128
    u32 nw32;   // nids / 32
    int rows;   // rows per strip of the strip-local pass: 32, or 16 for wide frames (see ccl_make_geom)
    int invert; // label the zero pixels instead (background regions, for hole borders)
    int conn4;  // 4-connectivity (background of an 8-connected foreground)
};

// word j of a row as the labelling sees it
__device__ __forceinline__ u64 ccl_word(const ccl_geom& G, const u64* __restrict__ fb, int idx, int j)
{
    u64 w = fb[idx];
    if (G.invert) {
        w = ~w;
        if (j == G.ww - 1 && (G.w & 63)) w &= (1ull << (G.w & 63)) - 1ull;
    }
    return w;
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
__device__ __forceinline__ u32 nstarts(u64 w) { return (u32)__popcll(w & ~(w << 1)); }

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
// links the larger root under the smaller; the absorbed root loses its bit in the root bitmap.  The two walks advance side by side,
// halving as they go: every hop is a round trip to the memory side (agent-scope loads pass this die's L2) and the chains do not depend
// on each other.
__device__ __forceinline__ void uf_unite(u32* p, u32* flags, u32 a, u32 b)
{
    u32 qa = ld_rlx(p + a), qb = ld_rlx(p + b);
    for (;;) {
        while (qa != a || qb != b) {
            const bool ma = qa != a, mb = qb != b;
            const u32 ga = ma ? ld_rlx(p + qa) : qa, gb = mb ? ld_rlx(p + qb) : qb;
            if (ma) { if (ga != qa) st_rlx(p + a, ga); a = qa; qa = ga; }
            if (mb) { if (gb != qb) st_rlx(p + b, gb); b = qb; qb = gb; }
        }
        if (a == b) return;
        if (a < b) { const u32 t = a; a = b; b = t; const u32 tq = qa; qa = qb; qb = tq; }
        const u32 old = atomicCAS(p + a, a, b);
        if (old == a) { atomicAnd(flags + (a >> 5), ~(1u << (a & 31))); return; }
        qa = old;
        qb = ld_rlx(p + b);
    }
}

// Unions of the segments of word (y, j) in global memory.  horiz: with the segment ending the previous
// word of the row; vert: with the 8-connected segments of row y-1.  A contact through the left/right
// neighbour word of the row above is skipped when the word straight above already bridges it (that
// row's own horizontal union connects them).
__device__ __forceinline__ void global_link_word(const u64* __restrict__ fb, const ccl_geom& G, u32* p, u32* flags, int y, int j,
                                                 int idx, u64 w, bool horiz, bool vert)
{
    if (horiz && (w & 1ull) && j > 0) {
        const u64 prev = ccl_word(G, fb, idx - 1, j - 1);
        if (prev >> 63) uf_unite(p, flags, seg_id(G, y, 64 * j), seg_id(G, y, 64 * (j - 1) + run_start(prev, 63)));
    }
    if (!vert || y == 0) return;
    const u64 um = ccl_word(G, fb, idx - G.ww, j);
    const u64 ulw = j > 0 ? ccl_word(G, fb, idx - G.ww - 1, j - 1) : 0ull;
    const u64 ul = G.conn4 ? 0ull : ulw;
    const u64 ur = (j + 1 < G.ww && !G.conn4) ? ccl_word(G, fb, idx - G.ww + 1, j + 1) : 0ull;
    if (!(um | (ul >> 63) | (ur & 1ull))) return;
    // A run of this row that comes in from the word to the left, under a run of the row above that comes in from the left as well:
    // the word to the left meets the same two runs at its bit 63 (and the runs are one component each once their rows' horizontal
    // unions are made), so this word leaves the pair alone.  A row-wide run - the inverted mask of a contour pass is mostly that -
    // otherwise costs one global union per word on one and the same pair of components.
    const bool left_made_it = (w & 1ull) && (um & 1ull) && j > 0 && (ulw >> 63) && (ccl_word(G, fb, idx - 1, j - 1) >> 63);
    u64 rem = w;
    while (rem) {
        const int s = __ffsll((long long)rem) - 1;
        const int e = run_end(rem, s);
        const u64 S = bit_range(s, e);
        rem &= ~S;
        const u32 me = seg_id(G, y, 64 * j + s);
        u64 c = um & (G.conn4 ? S : (S | (S << 1) | (S >> 1)));
        if (s == 0 && left_made_it) c &= ~bit_range(0, run_end(um, 0));
        while (c) {
            const int b = __ffsll((long long)c) - 1;
            const int st = run_start(um, b), en = run_end(um, b);
            uf_unite(p, flags, me, seg_id(G, y - 1, 64 * j + st));
            c &= ~bit_range(st, en);
        }
        if ((S & 1ull) && (ul >> 63) && !(um & 1ull)) uf_unite(p, flags, me, seg_id(G, y - 1, 64 * (j - 1) + run_start(ul, 63)));
        if ((S >> 63) && (ur & 1ull) && !(um >> 63)) uf_unite(p, flags, me, seg_id(G, y - 1, 64 * (j + 1)));
    }
}

// ---- strip-local union-find in LDS ------------------------------------------------------------------
#define CL_ROWS 32    // largest strip height; G.rows is the one in use
#define CL_CAP 512    // default segments per strip handled in LDS (foreground); denser strips fall back to global memory

__device__ __forceinline__ u32 lds_find(volatile u32* p, u32 x)
{
    for (;;) {
        const u32 q = p[x];
        if (q == x) return x;
        const u32 g = p[q];
        if (g == q) return q;
        p[x] = g;
        x = g;
    }
}
// read-only walk to the root: safe beside other walkers and beside threads that overwrite their OWN entry with its root
// (a path-halving find is not: its delayed `p[x] = grandparent` can land after x's owner stored the root there)
__device__ __forceinline__ u32 lds_root(const volatile u32* p, u32 x)
{
    for (;;) {
        const u32 q = p[x];
        if (q == x) return x;
        x = q;
    }
}
__device__ __forceinline__ void lds_unite(u32* p, u32 a, u32 b)
{
    for (;;) {
        a = lds_find(p, a);
        b = lds_find(p, b);
        if (a == b) return;
        if (a < b) { const u32 t = a; a = b; b = t; }
        const u32 old = atomicCAS(p + a, a, b);
        if (old == a) return;
        a = old;
    }
}

// iterate the words of the strip with a division-free (row, column) mapping: 8 rows x 32 columns per pass
#define CL_FOR_WORDS(r, j, i)                                   \
    for (int r = threadIdx.x >> 5; r < nrows; r += 8)           \
        for (int j = threadIdx.x & 31, i = r * ww + j; j < ww; j += 32, i += 32)

// Strip-local union-find over the word segments of CL_ROWS (or fewer) rows whose (possibly inverted) bit words are already
// staged in LDS as lbits[nrows][ww]; called by all 256 threads of a block.  Returns S, the number of segments of the strip
// (block-uniform).  For 0 < S <= cap it leaves in LDS: wbase[word] = local index of the word's first segment (indices follow
// the words in row-major order, segments of a word by position), lgid[ci] = segment id (when WANT_GID), lparent[] = a forest whose trees are the
// strip's components (links point at smaller indices); *my_first = local index of the first segment of the calling thread's
// first word in CL_FOR_WORDS order (its segments are numbered consecutively in that order).  The last unions may still be in
// flight: callers synchronise before reading lparent.  wbase needs nrows*ww + 2 words, lparent / lgid `cap` words each, wsum 4
// words and total_s 1 word of LDS.
struct cl_noprobe { __device__ __forceinline__ void operator()(int) const {} };
template <bool WANT_GID = true, typename PROBE = cl_noprobe>
__device__ __forceinline__ u32 ccl_local_unions(const ccl_geom& G, const u64* lbits, u32* wbase, u32* lparent, u32* lgid,
                                                u32* wsum, u32* total_s, int y0, int nrows, u32 cap, u32* my_first, PROBE probe = PROBE())
{
    const int ww = G.ww;
    const int tid = threadIdx.x;
    u32 cnt = 0;
    CL_FOR_WORDS(r, j, i) cnt += nstarts(lbits[i]);
    // block exclusive scan of the per-thread segment counts
    u32 inc = cnt;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    if (tid == 0) { u32 run = 0; for (int k = 0; k < 4; k++) { const u32 t = wsum[k]; wsum[k] = run; run += t; } *total_s = run; }
    __syncthreads();
    const u32 S = *total_s;
    *my_first = wsum[wv] + inc - cnt;
    if (S == 0 || S > cap) return S;
    probe(1);
    {
        // Segments get their local index here, and the segments of one row that continue across word boundaries get their run's
        // first segment as parent straight away: a scan over the 32 words a half-wave holds (a word of all ones passes the
        // incoming leader on, any other word starts a new one), not 30 neighbour unions that would leave a chain 30 links deep
        // for every later find to walk (measured: the union phase was 24 of the 31 us a block took).
        u32 run = wsum[wv] + inc - cnt;
        const int l32 = tid & 31;
        const u32 NONE = 0xffffffffu;
        for (int r = tid >> 5; r < nrows; r += 8) {
            u32 carry = NONE;   // leader of the run that leaves the previous chunk of this row through bit 63
            for (int j0 = 0; j0 < ww; j0 += 32) {
                const int j = j0 + l32, i = r * ww + j;
                const bool valid = j < ww;
                const u64 w = valid ? lbits[i] : 0ull;
                const u32 first = run;
                if (valid) wbase[i] = run;
                u64 st = w & ~(w << 1);
                while (st) {
                    const int s = __ffsll((long long)st) - 1;
                    st &= st - 1;
                    lparent[run] = run;
                    if (WANT_GID) lgid[run] = seg_id(G, y0 + r, 64 * j + s);
                    run++;
                }
                // f_j(x) = (pass && x != NONE) ? x : val  -  leader of the run that leaves word j through bit 63, given the one entering
                bool pass = w == ~0ull;
                u32 val = (w >> 63) ? run - 1u : NONE;
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) {
                    const u32 pv = __shfl_up(val, d, 32);
                    const int pp = __shfl_up((int)pass, d, 32);
                    if (l32 >= d && pass) {          // (earlier pp, pv) then (pass = true, val)
                        if (pp) val = pv;            // both pass: the earlier default stands in
                        else { val = pv != NONE ? pv : val; pass = false; }
                    }
                }
                const u32 lout = (pass && carry != NONE) ? carry : val;
                u32 lin = __shfl_up(lout, 1, 32);
                if (l32 == 0) lin = carry;
                if (valid && (w & 1ull) && lin != NONE) lparent[first] = lin;   // continues the run: lin < first
                carry = __shfl(lout, 31, 32);
            }
        }
    }
    __syncthreads();
    probe(2);
    CL_FOR_WORDS(r, j, i) {
        const u64 w = lbits[i];
        if (!w) continue;
        const u32 base = wbase[i];
        if (r == 0) continue;
        const u64 um = lbits[i - ww];
        const u64 ul = (j > 0 && !G.conn4) ? lbits[i - ww - 1] : 0ull;
        const u64 ur = (j + 1 < ww && !G.conn4) ? lbits[i - ww + 1] : 0ull;
        if (!(um | (ul >> 63) | (ur & 1ull))) continue;
        const u32 ubase = wbase[i - ww];
        const u64 ustarts = um & ~(um << 1);
        // both this row's run and the run above come in from the left: the word to the left asks for the same union
        const bool left_has_it = (w & 1ull) && (um & 1ull) && j > 0 && (lbits[i - 1] >> 63) && (lbits[i - ww - 1] >> 63);
        u64 rem = w;
        u32 me = base;
        while (rem) {
            const int s = __ffsll((long long)rem) - 1;
            const int e = run_end(rem, s);
            const u64 Sg = bit_range(s, e);
            rem &= ~Sg;
            u64 c = um & (G.conn4 ? Sg : (Sg | (Sg << 1) | (Sg >> 1)));
            if (s == 0 && left_has_it) c &= ~bit_range(0, run_end(um, 0));
            while (c) {
                const int b = __ffsll((long long)c) - 1;
                const int st = run_start(um, b), en = run_end(um, b);
                lds_unite(lparent, me, ubase + (u32)__popcll(ustarts & ((1ull << st) - 1ull)));
                c &= ~bit_range(st, en);
            }
            if ((Sg & 1ull) && (ul >> 63) && !(um & 1ull)) lds_unite(lparent, me, wbase[i - ww - 1] + nstarts(ul) - 1u);
            if ((Sg >> 63) && (ur & 1ull) && !(um >> 63)) lds_unite(lparent, me, wbase[i - ww + 1]);
            me++;
        }
    }
    return S;
}

// Strip-local labelling for the one-level path: resolves the strip's components entirely in LDS and writes
// parent[id] = smallest id of the segment's strip-local component (init + link in one pass); clears the strip's slice
// of the root bitmap and marks the strip-local representatives.  LDS as for ccl_local_unions plus lmin (`cap` words; may alias
// wbase when cap <= nrows*ww + 2).  fb = the frame's bit image in global memory (only the dense-strip fallback reads it, for
// rows of this strip).
__device__ __forceinline__ void ccl_local_strip(const ccl_geom& G, const u64* lbits, u32* wbase, u32* lparent, u32* lgid, u32* lmin,
                                                u32* wsum, u32* total_s, int y0, int nrows, int strip, int strips,
                                                const u64* __restrict__ fb, u32* __restrict__ gp, u32* __restrict__ gf, u32 cap = CL_CAP,
                                                u32* __restrict__ gz = nullptr)   // gz: a second bitmap of the same layout, cleared along with gf
{
    const int ww = G.ww;
    const int tid = threadIdx.x;
    // this strip's slice of the root bitmap (ids of G.rows rows = a multiple of 32 ids, so slices never share a word:
    // 32 rows always are, 16 rows when ceil(w/2) is even - ccl_make_geom only picks 16 then)
    {
        const u32 rows_ids = (G.numbering == VP_CCL_BLOCK2X2) ? 2u * (u32)G.wb : (u32)G.wb;   // ids per row pair / per row
        const u32 lo = (G.numbering == VP_CCL_BLOCK2X2) ? (u32)(y0 >> 1) * rows_ids : (u32)y0 * rows_ids;
        const u32 w0 = lo >> 5;
        const u32 w1 = (strip == strips - 1) ? G.nw32 : ((G.numbering == VP_CCL_BLOCK2X2) ? ((u32)((y0 + G.rows) >> 1) * rows_ids) >> 5
                                                                                         : ((u32)(y0 + G.rows) * rows_ids) >> 5);
        for (u32 i = w0 + tid; i < w1; i += 256) { gf[i] = 0u; if (gz) gz[i] = 0u; }
    }
    u32 my_first;
    const u32 S = ccl_local_unions(G, lbits, wbase, lparent, lgid, wsum, total_s, y0, nrows, cap, &my_first);
    if (S == 0) return;
    if (S > cap) {
        // dense strip: same algorithm in global memory, restricted to this strip's rows
        CL_FOR_WORDS(r, j, i) {
            u64 st = lbits[i] & ~(lbits[i] << 1);
            while (st) {
                const int s = __ffsll((long long)st) - 1;
                st &= st - 1;
                const u32 id = seg_id(G, y0 + r, 64 * j + s);
                st_rlx(gp + id, id);
                atomicOr(gf + (id >> 5), 1u << (id & 31));
            }
        }
        __threadfence();
        __syncthreads();
        CL_FOR_WORDS(r, j, i) {
            const u64 w = lbits[i];
            if (w) global_link_word(fb, G, gp, gf, y0 + r, j, (y0 + r) * ww + j, w, true, r > 0);
        }
        return;
    }
    __syncthreads();
    // lmin may share its LDS with wbase, which the unions above were the last to read
    for (u32 ci = tid; ci < S; ci += 256) lmin[ci] = 0xffffffffu;
    __syncthreads();
    for (u32 ci = tid; ci < S; ci += 256) atomicMin(lmin + lds_find(lparent, ci), lgid[ci]);
    __syncthreads();
    for (u32 ci = tid; ci < S; ci += 256) {
        const u32 id = lgid[ci];
        const u32 m = lmin[lds_find(lparent, ci)];
        gp[id] = m;
        if (m == id) atomicOr(gf + (id >> 5), 1u << (id & 31));   // strip-local representative = root candidate
    }
}

