// Two-level labelling (included by vp_ccl.hip): the form of the component labelling the chain runs for ordinary frames.
//
// The one-level kernels of vp_ccl.hip keep everything in arrays indexed by segment id and resolve a frame in five dependent
// launches (local, boundary, rank, stats, final), each a few global round trips deep, with uncoalesced look-ups per segment.
// Here the unit above the segment is the strip-local COMPONENT:
//
//   k_ccl2_local  per 32-row strip, all in LDS: segments -> union-find -> the strip's components with their statistics and their
//                 numbering key (smallest segment id); writes a short component list per strip, a dense per-word component
//                 index (and a per-segment one for the few words that hold several segments)
//   k_ccl2_merge  one block per frame: reads the lists, unites components across the strip boundaries in LDS, ranks the surviving
//                 roots by key (= cv2's label), merges the statistics, writes the stats / centroid rows and a
//                 (strip, component) -> label table
//   k_ccl2_write  label image: label = table[strip][component index of the word], one cached look-up in front of the store
//
// Frames that do not fit (a strip with more segments than its LDS union-find holds or more components than C2_RC, a frame with
// more than C2_MCAP strip components) are flagged `crowded` by k_ccl2_merge and finished by the one-level kernels, which are
// launched unconditionally on a side stream and leave at once for every other frame (vpk_ccl).

// C2_RC (vp_ccl.hip): stride of the per-strip component tables; a kernel argument (rc <= C2_RC) bounds the count in use
#ifdef VP_PROBE   // measurement builds only (tools/build_probe.sh): ticks between probe points, one slot per block (plain stores)
#define C2_PROBE_BLOCKS 65536
__device__ unsigned int g_c2_probe[2][C2_PROBE_BLOCKS][16];
#define C2_PROBE_BEGIN unsigned long long pt_ = clock64()
#define C2_PROBE(k, i) do { if (threadIdx.x == 0 && blockIdx.x < C2_PROBE_BLOCKS) { const unsigned long long now_ = clock64(); g_c2_probe[k][blockIdx.x][i] = (unsigned int)(now_ - pt_); pt_ = now_; } } while (0)
#else
#define C2_PROBE_BEGIN do { } while (0)
#define C2_PROBE(k, i) do { } while (0)
#endif

#define C2_MCAP 2048           // strip components of one frame held in the merge block's LDS
#define C2_MAXSTRIPS 256
#define C2_DENSE 0xffffffffu   // ncomp value of a strip the strip-local pass could not resolve
#define C2_THREADS 1024
#define C2_PER (C2_MCAP / C2_THREADS)

// grid: n * strips blocks of 256 threads.
// dynamic LDS: lbits[nw] u64 | wbase[nw + 2] u32 (later: list index of each root) | lparent[cap] | accumulators of rc components (44 B each)
__global__ __launch_bounds__(256) void k_ccl2_local(const u64* __restrict__ bits, ccl_geom G, int strips, int cap, int rc,
                                                    u32* __restrict__ ncomp, contrib* __restrict__ recs, c2_box* __restrict__ bgbox,
                                                    u32* __restrict__ wordcomp, u32* __restrict__ segcomp)
{
    extern __shared__ __attribute__((aligned(16))) u64 cl_lds[];
    __shared__ u32 wsum[4];
    __shared__ u32 total_s;
    __shared__ c2_box bgp[4];
    __shared__ u32 nroots_s;
    const int ww = G.ww;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int frame = blockIdx.x / strips, strip = blockIdx.x - frame * strips;
    const int y0 = strip * G.rows;
    const int nrows = min(G.rows, G.h - y0);
    const int nwmax = G.rows * ww;
    u64* lbits = cl_lds;
    u32* wbase = reinterpret_cast<u32*>(cl_lds + nwmax);
    u32* lparent = wbase + (nwmax + 2);
    u32* lidx = wbase;                                 // list index of a root, under the root's local index (wbase is dead by then)
    u64* a_sx = reinterpret_cast<u64*>(lparent + cap); // cap even and the offset a multiple of 8 (host)
    u64* a_sy = a_sx + rc;
    u32* a_area = reinterpret_cast<u32*>(a_sy + rc);
    int* a_minx = reinterpret_cast<int*>(a_area + rc);
    int* a_maxx = a_minx + rc;
    int* a_miny = a_maxx + rc;
    int* a_maxy = a_miny + rc;
    u32* a_key = reinterpret_cast<u32*>(a_maxy + rc);
    const u64* fb = bits + (size_t)frame * G.h * ww;
    const size_t sidx = (size_t)frame * strips + strip;
    const u64 lastmask = (G.w & 63) ? ((1ull << (G.w & 63)) - 1ull) : ~0ull;
    // stage the strip; bounding box of its zero pixels on the way (the frame's background row needs it, its sums follow from the
    // foreground's)
    c2_box bb = {INT_MAX, INT_MIN, INT_MAX, INT_MIN};
    C2_PROBE_BEGIN;
    CL_FOR_WORDS(r, j, i) {
        const u64 w = fb[(size_t)(y0 + r) * ww + j];
        lbits[i] = w;
        u64 z = ~w;
        if (j == ww - 1) z &= lastmask;
        if (z) {
            bb.minx = min(bb.minx, 64 * j + (__ffsll((long long)z) - 1));
            bb.maxx = max(bb.maxx, 64 * j + 63 - __clzll(z));
            bb.miny = min(bb.miny, y0 + r);
            bb.maxy = max(bb.maxy, y0 + r);
        }
    }
    if (tid < rc) {
        a_sx[tid] = 0; a_sy[tid] = 0; a_area[tid] = 0; a_key[tid] = 0xffffffffu;
        a_minx[tid] = INT_MAX; a_maxx[tid] = INT_MIN; a_miny[tid] = INT_MAX; a_maxy[tid] = INT_MIN;
    }
    if (tid == 0) nroots_s = 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        bb.minx = min(bb.minx, __shfl_xor(bb.minx, d));
        bb.maxx = max(bb.maxx, __shfl_xor(bb.maxx, d));
        bb.miny = min(bb.miny, __shfl_xor(bb.miny, d));
        bb.maxy = max(bb.maxy, __shfl_xor(bb.maxy, d));
    }
    if (lane == 0) bgp[wv] = bb;
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < 4; k++) {
            bb.minx = min(bb.minx, bgp[k].minx); bb.maxx = max(bb.maxx, bgp[k].maxx);
            bb.miny = min(bb.miny, bgp[k].miny); bb.maxy = max(bb.maxy, bgp[k].maxy);
        }
        bgbox[sidx] = bb;
    }
    u32 my_first;
    C2_PROBE(0, 0);   // staged + background box
    auto probe = [&](int i) { C2_PROBE(0, i); };   // 1: counted + scanned, 2: indices + row leaders
    const u32 S = ccl_local_unions<false>(G, lbits, wbase, lparent, nullptr, wsum, &total_s, y0, nrows, (u32)cap, &my_first, probe);
    if (S == 0 || S > (u32)cap) {
        if (tid == 0) ncomp[sidx] = S ? C2_DENSE : 0u;
        return;
    }
    __syncthreads();
    C2_PROBE(0, 3);   // unions
    // every segment points at its root (read-only walks: each thread overwrites only its own entries); a root takes the next free
    // place of the strip's list - the list is unordered, the merge orders components by key
    for (u32 c0 = 0; c0 < S; c0 += 256) {
        const u32 ci = c0 + tid;
        bool isroot = false;
        if (ci < S) {
            const u32 r = lds_root(lparent, ci);
            lparent[ci] = r;
            isroot = r == ci;
        }
        const unsigned long long m = __ballot(isroot);
        if (m) {
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&nroots_s, (u32)__popcll(m));
            base = __shfl(base, 0);
            if (isroot) lidx[ci] = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
        }
    }
    __syncthreads();
    C2_PROBE(0, 4);   // roots stored, list indices
    const u32 R = nroots_s;
    if (R > (u32)rc) {
        if (tid == 0) ncomp[sidx] = C2_DENSE;
        return;
    }
    auto acc_add = [&](u32 k, const contrib& c) {
        atomicAdd(a_area + k, c.area);
        atomicAdd((unsigned long long*)(a_sx + k), (unsigned long long)c.sx);
        atomicAdd((unsigned long long*)(a_sy + k), (unsigned long long)c.sy);
        atomicMin(a_minx + k, c.minx);
        atomicMax(a_maxx + k, c.maxx);
        atomicMin(a_miny + k, c.miny);
        atomicMax(a_maxy + k, c.maxy);
        atomicMin(a_key + k, c.pad);
    };
    // second walk over the words, in the order that numbered the segments: statistics and numbering key (smallest segment id) of
    // every component, component index of every word's first segment (dense) and of the segments of words that hold several
    {
        u32 run = my_first;
        const int l32 = tid & 31;
        u32* wc = wordcomp + (size_t)frame * G.h * ww;
        u32* sc = segcomp + (size_t)frame * G.nids;
        const u32 NONE = 0xffffffffu;
        for (int rr = 0; rr < G.rows; rr += 8) {
            const int r = rr + (tid >> 5);
            for (int j0 = 0; j0 < ww; j0 += 32) {
                const int j = j0 + l32;
                const bool valid = r < nrows && j < ww;
                const u64 w = valid ? lbits[r * ww + j] : 0ull;
                const int y = y0 + r;
                u64 rem = w;
                const bool multi = nstarts(w) > 1u;
                contrib c0;
                contrib_zero(c0);
                c0.pad = NONE;
                u32 k0 = NONE;
                if (w) {
                    const int s = __ffsll((long long)rem) - 1;
                    const int e = run_end(rem, s);
                    rem &= ~bit_range(s, e);
                    k0 = lidx[lparent[run]];
                    run++;
                    wc[(size_t)y * ww + j] = k0;
                    const u32 id = seg_id(G, y, 64 * j + s);
                    if (multi) sc[id] = k0;
                    const u32 len = (u32)(e - s + 1);
                    const int xs = 64 * j + s, xe = 64 * j + e;
                    c0.area = len; c0.sx = (u64)len * (u64)(xs + xe) / 2ull; c0.sy = (u64)len * (u64)y;
                    c0.minx = xs; c0.maxx = xe; c0.miny = c0.maxy = y; c0.pad = id;
                }
                // a wave whose first segments all belong to one component (the inside of a blob, a full mask) combines them
                // with shuffles and adds once instead of queueing 64 lanes on the same LDS words
                const unsigned long long act = __ballot(k0 != NONE);
                if (act) {
                    const int lead = __ffsll((long long)act) - 1;
                    const u32 ref = __shfl(k0, lead);
                    if (__popcll(act) >= 8 && __all(k0 == NONE || k0 == ref)) {
                        wave_combine(c0);
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) c0.pad = min(c0.pad, (u32)__shfl_xor(c0.pad, d));
                        if (lane == lead) acc_add(ref, c0);
                    } else if (k0 != NONE) {
                        acc_add(k0, c0);
                    }
                }
                while (rem) {
                    const int s = __ffsll((long long)rem) - 1;
                    const int e = run_end(rem, s);
                    rem &= ~bit_range(s, e);
                    const u32 k = lidx[lparent[run]];
                    run++;
                    const u32 id = seg_id(G, y, 64 * j + s);
                    sc[id] = k;
                    contrib c;
                    const u32 len = (u32)(e - s + 1);
                    const int xs = 64 * j + s, xe = 64 * j + e;
                    c.area = len; c.sx = (u64)len * (u64)(xs + xe) / 2ull; c.sy = (u64)len * (u64)y;
                    c.minx = xs; c.maxx = xe; c.miny = c.maxy = y; c.pad = id;
                    acc_add(k, c);
                }
            }
        }
    }
    __syncthreads();
    C2_PROBE(0, 5);   // second walk: statistics, word / segment component indices
    contrib* out = recs + sidx * C2_RC;
    for (u32 k = tid; k < R; k += 256) {
        contrib c;
        c.area = a_area[k]; c.minx = a_minx[k]; c.maxx = a_maxx[k]; c.miny = a_miny[k]; c.maxy = a_maxy[k]; c.pad = a_key[k];
        c.sx = a_sx[k]; c.sy = a_sy[k];
        out[k] = c;
    }
    if (tid == 0) ncomp[sidx] = R;
    C2_PROBE(0, 6);   // list written
    C2_PROBE(0, 15);
#ifdef VP_PROBE
    if (tid == 0 && blockIdx.x < C2_PROBE_BLOCKS) g_c2_probe[0][blockIdx.x][14] = 0x600dc0deu;   // ran to the end
#endif
}

// ---- merge: one block of C2_THREADS threads per frame ---------------------------------------------------------------
__device__ __forceinline__ u32 c2_block_scan_excl(u32 v, u32* wtot, u32* total)   // exclusive scan over the block's threads
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    __syncthreads();                  // wtot may still be read from an earlier scan
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    u32 off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < C2_THREADS / 64; k++) { const u32 t = wtot[k]; if (k < wv) off += t; tot += t; }
    *total = tot;
    return off + inc - v;
}

__global__ __launch_bounds__(C2_THREADS) void k_ccl2_merge(const u64* __restrict__ bits, ccl_geom G, int strips, int mcap,
                                                           const u32* __restrict__ ncomp, const contrib* __restrict__ recs,
                                                           const c2_box* __restrict__ bgbox, const u32* __restrict__ wordcomp,
                                                           const u32* __restrict__ segcomp, u32* __restrict__ complabel,
                                                           u32* __restrict__ crowded, int32_t* __restrict__ nlabels,
                                                           int32_t* __restrict__ stats, double* __restrict__ cent, int max_labels)
{
    __shared__ u32 sbase[C2_MAXSTRIPS + 1];
    __shared__ u32 wtot[C2_THREADS / 64];
    __shared__ u32 par[C2_MCAP];
    __shared__ u32 lab[C2_MCAP];
    __shared__ u32 key[C2_MCAP];    // numbering key (smallest segment id); after the unions a root's entry holds its component's
    __shared__ u32 rkeys[C2_MCAP];  // keys of the roots, compacted
    __shared__ u32 a_area[C2_MCAP];
    __shared__ int a_minx[C2_MCAP], a_maxx[C2_MCAP], a_miny[C2_MCAP], a_maxy[C2_MCAP];
    __shared__ u64 a_sx[C2_MCAP], a_sy[C2_MCAP];
    __shared__ u64 tot_sx, tot_sy;
    __shared__ u32 tot_area;
    __shared__ c2_box bgs;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ww = G.ww;
    C2_PROBE_BEGIN;
    // list sizes -> offsets
    const u32 nc_raw = tid < strips ? ncomp[(size_t)f * strips + tid] : 0u;
    const bool dense = nc_raw == C2_DENSE;
    u32 C;
    const u32 ex = c2_block_scan_excl(dense ? 0u : nc_raw, wtot, &C);
    if (tid <= strips) sbase[tid] = ex;     // tid == strips holds the total (its own value is 0)
    if (tid == 0) { tot_sx = 0; tot_sy = 0; tot_area = 0; bgs.minx = INT_MAX; bgs.maxx = INT_MIN; bgs.miny = INT_MAX; bgs.maxy = INT_MIN; }
    const bool any_dense = __syncthreads_or(dense);
    if (any_dense || C > (u32)mcap) {
        if (tid == 0) crowded[f] = 1u;
        return;
    }
    if (tid == 0) crowded[f] = 0u;
    C2_PROBE(1, 0);   // list sizes read and scanned
    // component records -> LDS (wave per strip)
    for (int s = wv; s < strips; s += C2_THREADS / 64) {
        const u32 b0 = sbase[s], cnt = sbase[s + 1] - b0;
        const contrib* src = recs + ((size_t)f * strips + s) * C2_RC;
        for (u32 k = lane; k < cnt; k += 64) {
            const contrib c = src[k];
            const u32 i = b0 + k;
            par[i] = i;
            key[i] = c.pad;
            a_area[i] = c.area; a_minx[i] = c.minx; a_maxx[i] = c.maxx; a_miny[i] = c.miny; a_maxy[i] = c.maxy;
            a_sx[i] = c.sx; a_sy[i] = c.sy;
        }
    }
    // background box
    if (tid < strips) {
        const c2_box b = bgbox[(size_t)f * strips + tid];
        if (b.minx != INT_MAX) {
            atomicMin(&bgs.minx, b.minx); atomicMax(&bgs.maxx, b.maxx);
            atomicMin(&bgs.miny, b.miny); atomicMax(&bgs.maxy, b.maxy);
        }
    }
    __syncthreads();
    C2_PROBE(1, 1);   // records in LDS
    // unions across the strip boundaries: one thread per word of the first row of strips 1 ..
    {
        const u64* fb = bits + (size_t)f * G.h * ww;
        const u32* wc = wordcomp + (size_t)f * G.h * ww;
        const u32* sc = segcomp + (size_t)f * G.nids;
        const int items = (strips - 1) * ww;
        for (int t = tid; t < items; t += C2_THREADS) {
            const int b = t / ww, j = t - b * ww;
            const int y = (b + 1) * G.rows;
            const size_t idx = (size_t)y * ww + j;
            const u64 w = fb[idx];
            const u64 um = fb[idx - ww];
            const u64 ul = j > 0 ? fb[idx - ww - 1] : 0ull;
            const u64 ur = j + 1 < ww ? fb[idx - ww + 1] : 0ull;
            if (!w || !(um | (ul >> 63) | (ur & 1ull))) continue;
            const u32 lo = sbase[b], hi = sbase[b + 1];
            const u32 k_self = wc[idx];
            const u32 k_um = um ? wc[idx - ww] : 0u;
            const u32 k_ul = (ul >> 63) ? wc[idx - ww - 1] : 0u;
            const u32 k_ur = (ur & 1ull) ? wc[idx - ww + 1] : 0u;
            const int um_first = um ? __ffsll((long long)um) - 1 : 0;
            u64 rem = w;
            bool first = true;
            while (rem) {
                const int s = __ffsll((long long)rem) - 1;
                const int e = run_end(rem, s);
                const u64 Sg = bit_range(s, e);
                rem &= ~Sg;
                const u32 me = hi + (first ? k_self : sc[seg_id(G, y, 64 * j + s)]);
                first = false;
                u64 c = um & (Sg | (Sg << 1) | (Sg >> 1));
                while (c) {
                    const int bt = __ffsll((long long)c) - 1;
                    const int st = run_start(um, bt), en = run_end(um, bt);
                    lds_unite(par, me, lo + (st == um_first ? k_um : sc[seg_id(G, y - 1, 64 * j + st)]));
                    c &= ~bit_range(st, en);
                }
                if ((Sg & 1ull) && (ul >> 63) && !(um & 1ull)) {
                    const int st = run_start(ul, 63);
                    const bool ul_first = (ul & ((1ull << st) - 1ull)) == 0ull;
                    lds_unite(par, me, lo + (ul_first ? k_ul : sc[seg_id(G, y - 1, 64 * (j - 1) + st)]));
                }
                if ((Sg >> 63) && (ur & 1ull) && !(um >> 63)) lds_unite(par, me, lo + k_ur);
            }
        }
    }
    __syncthreads();
    C2_PROBE(1, 2);   // boundary unions
    // roots (read-only walks; every thread stores the root over its own entries), key of a component = smallest key of its members
    u32 root[C2_PER];
    u32 nroot_mine = 0;
#pragma unroll
    for (int q = 0; q < C2_PER; q++) {
        const u32 i = (u32)tid * C2_PER + q;
        root[q] = 0xffffffffu;
        if (i < C) {
            root[q] = lds_root(par, i);
            par[i] = root[q];
            if (root[q] == i) nroot_mine++;
            else atomicMin(key + root[q], key[i]);
        }
    }
    u32 R;
    u32 pos = c2_block_scan_excl(nroot_mine, wtot, &R);   // (its barriers also complete the keys)
#pragma unroll
    for (int q = 0; q < C2_PER; q++) {
        const u32 i = (u32)tid * C2_PER + q;
        if (i < C && root[q] == i) rkeys[pos++] = key[i];
    }
    __syncthreads();
    C2_PROBE(1, 3);   // roots, keys, compacted
    // cv2's label of a component = 1 + number of components with a smaller key (keys are distinct)
#pragma unroll
    for (int q = 0; q < C2_PER; q++) {
        const u32 i = (u32)tid * C2_PER + q;
        const bool isroot = i < C && root[q] == i;
        if (__any(isroot)) {
            const u32 mine = isroot ? key[i] : 0u;
            u32 cnt = 0;
            for (u32 j = 0; j < R; j++) cnt += rkeys[j] < mine ? 1u : 0u;
            if (isroot) lab[i] = cnt + 1u;
        }
    }
    __syncthreads();
    C2_PROBE(1, 4);   // ranks
    // statistics of absorbed components move to their roots
#pragma unroll
    for (int q = 0; q < C2_PER; q++) {
        const u32 i = (u32)tid * C2_PER + q;
        if (i < C && root[q] != i) {
            const u32 r = root[q];
            atomicAdd(a_area + r, a_area[i]);
            atomicAdd((unsigned long long*)(a_sx + r), (unsigned long long)a_sx[i]);
            atomicAdd((unsigned long long*)(a_sy + r), (unsigned long long)a_sy[i]);
            atomicMin(a_minx + r, a_minx[i]); atomicMax(a_maxx + r, a_maxx[i]);
            atomicMin(a_miny + r, a_miny[i]); atomicMax(a_maxy + r, a_maxy[i]);
        }
    }
    // (strip, component) -> label table
    for (int s = wv; s < strips; s += C2_THREADS / 64) {
        const u32 b0 = sbase[s], cnt = sbase[s + 1] - b0;
        u32* dst = complabel + ((size_t)f * strips + s) * C2_RC;
        for (u32 k = lane; k < cnt; k += 64) dst[k] = lab[par[b0 + k]];
    }
    __syncthreads();
    C2_PROBE(1, 5);   // statistics moved, label table written
    const int nl = (int)R + 1;
    if (tid == 0 && nlabels) nlabels[f] = nl;
    if (!stats && !cent) return;
    // rows of the roots; foreground totals for the background row
    {
        u32 t_area = 0;
        u64 t_sx = 0, t_sy = 0;
#pragma unroll
        for (int q = 0; q < C2_PER; q++) {
            const u32 i = (u32)tid * C2_PER + q;
            if (i < C && root[q] == i) {
                const u32 l = lab[i];
                t_area += a_area[i]; t_sx += a_sx[i]; t_sy += a_sy[i];
                if (l < (u32)max_labels) {
                    const size_t o = (size_t)f * max_labels + l;
                    if (stats) {
                        int32_t* sp = stats + o * 5;
                        sp[0] = a_minx[i];
                        sp[1] = a_miny[i];
                        sp[2] = (int32_t)((u32)a_maxx[i] - (u32)a_minx[i] + 1u);
                        sp[3] = (int32_t)((u32)a_maxy[i] - (u32)a_miny[i] + 1u);
                        sp[4] = (int32_t)a_area[i];
                    }
                    if (cent) {
                        const double area = (double)a_area[i];
                        cent[o * 2] = (double)a_sx[i] / area;
                        cent[o * 2 + 1] = (double)a_sy[i] / area;
                    }
                }
            }
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { t_area += __shfl_xor(t_area, d); t_sx += __shfl_xor(t_sx, d); t_sy += __shfl_xor(t_sy, d); }
        if (lane == 0 && t_area) {
            atomicAdd(&tot_area, t_area);
            atomicAdd((unsigned long long*)&tot_sx, (unsigned long long)t_sx);
            atomicAdd((unsigned long long*)&tot_sy, (unsigned long long)t_sy);
        }
    }
    __syncthreads();
    if (tid == 0) {
        const u64 W = (u64)G.w, H = (u64)G.h;
        const u32 area = (u32)(W * H) - tot_area;
        const u64 sx = H * (W * (W - 1ull) / 2ull) - tot_sx;
        const u64 sy = W * (H * (H - 1ull) / 2ull) - tot_sy;
        const size_t o = (size_t)f * max_labels;
        if (stats) {
            int32_t* sp = stats + o * 5;
            sp[0] = bgs.minx;
            sp[1] = bgs.miny;
            sp[2] = (int32_t)((u32)bgs.maxx - (u32)bgs.minx + 1u);
            sp[3] = (int32_t)((u32)bgs.maxy - (u32)bgs.miny + 1u);
            sp[4] = (int32_t)area;
        }
        if (cent) {
            cent[o * 2] = (double)sx / (double)area;
            cent[o * 2 + 1] = (double)sy / (double)area;
        }
    }
    C2_PROBE(1, 6);   // rows of the roots and the background
    // rows past the last label read as zeros
    for (int l = nl + tid; l < max_labels; l += C2_THREADS) {
        const size_t o = (size_t)f * max_labels + l;
        if (stats) { int32_t* sp = stats + o * 5; sp[0] = sp[1] = sp[2] = sp[3] = sp[4] = 0; }
        if (cent) { cent[o * 2] = 0.0; cent[o * 2 + 1] = 0.0; }
    }
    C2_PROBE(1, 7);   // zero rows
    C2_PROBE(1, 15);
#ifdef VP_PROBE
    if (tid == 0 && blockIdx.x < C2_PROBE_BLOCKS) g_c2_probe[1][blockIdx.x][14] = 0x600dc0deu;
#endif
}

// ---- label image --------------------------------------------------------------------------------------------------------
// As k_ccl_write (one lane = 4 px = one 16-B store, loads of a thread batched before its stores, 8 waves per SIMD), on a grid of
// (row groups, frames): a block never leaves its frame or its strip, so the crowded flag and the strip's label table are
// block-uniform.  Frames resolved by the two-level path carry component indices in the word / segment arrays and take the label
// from the strip's table, staged in LDS by loads issued together with the first batch of bit words (one global round trip, as
// before); crowded frames, finished by the one-level kernels, carry labels there.
__global__ __launch_bounds__(256, 8) void k_ccl2_write(const u64* __restrict__ bits, ccl_geom G, int strips, int rc, const u32* __restrict__ segcomp,
                                                    const u32* __restrict__ wordcomp, const u32* __restrict__ complabel,
                                                    const u32* __restrict__ crowded, int32_t* __restrict__ labels, u32 gpr, u32 gpr_magic)
{
    __shared__ u32 ltab[C2_RC];
    const u32 f = blockIdx.y;
    const u32 y0 = blockIdx.x * WR_ROWS;
    const u32 nrows = min((u32)WR_ROWS, (u32)G.h - y0);
    const u32 ngroups = nrows * gpr;
    int32_t* lrow0 = labels + ((size_t)f * G.h + y0) * G.w;
    const bool vec = (G.w & 3) == 0 && ((((uintptr_t)lrow0) & 15) == 0);
    const u64* brow0 = bits + ((size_t)f * G.h + y0) * G.ww;
    const u32* wrow0 = wordcomp + ((size_t)f * G.h + y0) * G.ww;
    const u32* sl = segcomp + (size_t)f * G.nids;
    const u32 direct = crowded[f];
    if ((int)threadIdx.x < rc) ltab[threadIdx.x] = complabel[((size_t)f * strips + y0 / (u32)G.rows) * C2_RC + threadIdx.x];
    bool staged = false;
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
        if (!staged) { __syncthreads(); staged = true; }   // the table is in LDS (block-uniform branch)
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
                const int y = (int)(y0 + rl[k]);
                // first run of the nibble, and what is left after it (at most one more run)
                const int tz = __ffs((int)nib) - 1;
                const u32 t = nib >> tz;
                const int runlen = __ffs((int)~t) - 1;
                m1[k] = ((1u << runlen) - 1u) << tz;
                m2[k] = nib & ~m1[k];
                la[k] = sl[seg_id(G, y, 64 * j + run_start(w[k], sub + tz))];
                if (m2[k]) lb[k] = sl[seg_id(G, y, 64 * j + sub + (__ffs((int)m2[k]) - 1))];
            }
        }
        if (!direct) {
#pragma unroll
            for (int k = 0; k < WR_K; k++) {
                la[k] = ltab[m1[k] ? la[k] : 0u];
                if (m2[k]) lb[k] = ltab[lb[k]];
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
