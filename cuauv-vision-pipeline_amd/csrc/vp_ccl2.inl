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
// more than C2_MCAP strip components) are flagged `crowded` by k_ccl2_merge, which appends them to a list; the crowded-frame kernels
// of vp_ccl3.inl (five launches behind the merge, on the same stream) finish the frames of that list and leave at once when it is
// empty.  Geometries those kernels do not take fall back to the one-level kernels of vp_ccl.hip (vpk_ccl).

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

// grid: n * strips blocks of 256 threads; frames up to 64 words wide (row masks are 64-bit).
//
// Built for masks that are mostly background with a few blobs:
//   * three 64-bit masks per row (word non-zero / all ones / bit 63 set), made by ballots while the strip is staged, answer the
//     questions about neighbouring words that would otherwise be LDS reads and scans: "does this word's first segment continue a
//     run from the left, and where does that run start" is a few bit operations;
//   * everything after staging walks the list of NON-ZERO words only, so lanes are busy on sparse masks;
//   * vertical links are not union-find unions: a contact between a segment and a segment of the row above is one
//     atomicMin(parent[leader of my run], leader of the run above) - leaders of a row above always have smaller indices, so the
//     parent pointers form a forest by construction.  Only when a run touches two DIFFERENT runs above (the returned old value tells)
//     does a real union remain; those go to a small queue and are united afterwards (compare-and-swap union-find).
// dynamic LDS: lbits[nw] u64 | wbase[nw] u32 | nzlist[nw] u16 (+pad) | lparent[cap] | tail: union queue, then root -> list index, then
// accumulators of rc components (44 B each)
__global__ __launch_bounds__(256) void k_ccl2_local(const u64* __restrict__ bits, ccl_geom G, int strips, int cap, int rc, int tail_words,
                                                    u32* __restrict__ ncomp, contrib* __restrict__ recs, c2_box* __restrict__ bgbox,
                                                    u32* __restrict__ wordcomp, u32* __restrict__ segcomp, u32* __restrict__ c3_ncrowded)
{
    extern __shared__ __attribute__((aligned(16))) u64 cl_lds[];
    __shared__ u64 m_nz[CL_ROWS], m_ones[CL_ROWS], m_b63[CL_ROWS];
    if (blockIdx.x == 0 && threadIdx.x == 0) *c3_ncrowded = 0u;   // the list k_ccl2_merge appends the frames it hands over to (vp_ccl3.inl)
    __shared__ u32 rowoff[CL_ROWS + 1];
    __shared__ u32 wsum[4];
    __shared__ u32 nroots_s, nqueue_s;
    const int ww = G.ww;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l32 = tid & 31;
    const int frame = blockIdx.x / strips, strip = blockIdx.x - frame * strips;
    const int y0 = strip * G.rows;
    const int nrows = min(G.rows, G.h - y0);
    const int nwmax = G.rows * ww;
    u64* lbits = cl_lds;
    u32* wbase = reinterpret_cast<u32*>(cl_lds + nwmax);
    unsigned short* nzlist = reinterpret_cast<unsigned short*>(wbase + nwmax);
    u32* lparent = wbase + nwmax + (nwmax + 1) / 2 + ((nwmax + (nwmax + 1) / 2) & 1);   // 8-byte aligned
    u32* tail = lparent + cap;                         // cap is even (host)
    u64* a_sx = reinterpret_cast<u64*>(tail);
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
    C2_PROBE_BEGIN;
    // ---- stage the strip (all loads of a thread first), row masks ---------------------------------------------------------------------
    // a word is "full" when all its pixels are set (the last word of a row: all its valid pixels; it never hands a run on, so the
    // run logic may take it as all ones)
    bool multi_mine = false;
    {
        u64 wreg[4][2];                                // (rows / 8) x ceil(ww / 32) <= 4 x 2 words per thread; indices are compile-time
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b2 = 0; b2 < 2; b2++) {
                const int r = a * 8 + (tid >> 5), j = b2 * 32 + l32;
                wreg[a][b2] = (a * 8 < G.rows && r < nrows && j < ww) ? fb[(size_t)(y0 + r) * ww + j] : 0ull;
            }
#pragma unroll
        for (int a = 0; a < 4; a++) {
            if (a * 8 >= G.rows) break;                // block-uniform
            const int r = a * 8 + (tid >> 5);
            u64 acc_nz = 0, acc_full = 0, acc_b63 = 0; // used by lanes 0 and 32: masks of the row their half-wave holds
#pragma unroll
            for (int b2 = 0; b2 < 2; b2++) {
                if (b2 * 32 >= ww) break;
                const int j = b2 * 32 + l32;
                const u64 w = wreg[a][b2];
                if (j < ww) lbits[r * ww + j] = w;
                multi_mine |= nstarts(w) > 1u;
                const u64 b_nz = __ballot(w != 0ull), b_full = __ballot(w == (j == ww - 1 ? lastmask : ~0ull)), b_63 = __ballot((w >> 63) != 0ull);
                const int sh = (lane & 32);            // half of the ballot that belongs to this lane's row
                acc_nz |= ((b_nz >> sh) & 0xffffffffull) << (b2 * 32);
                acc_full |= ((b_full >> sh) & 0xffffffffull) << (b2 * 32);
                acc_b63 |= ((b_63 >> sh) & 0xffffffffull) << (b2 * 32);
            }
            if (l32 == 0) { m_nz[r] = acc_nz; m_ones[r] = acc_full; m_b63[r] = acc_b63; }
        }
    }
    if (tid == 0) { nroots_s = 0; nqueue_s = 0; }
    const bool any_multi = __syncthreads_or(multi_mine);
    // ---- list of the non-zero words, row-major; bounding box of the zero pixels from the masks (the frame's background row needs it:
    // a row holds a zero pixel when one of its words is not full; the first / last such word of each row gives the x range) ----------------
    if (wv == 0) {
        const u32 c = lane < G.rows ? (u32)__popcll(m_nz[lane]) : 0u;
        u32 inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane <= G.rows) rowoff[lane] = inc - c;    // lane == rows: the total (c = 0 there)
    } else if (wv == 1) {
        c2_box bb = {INT_MAX, INT_MIN, INT_MAX, INT_MIN};
        const u64 colmask = ww >= 64 ? ~0ull : ((1ull << ww) - 1ull);
        if (lane < nrows) {
            const u64 nf = ~m_ones[lane] & colmask;    // words of this row that hold a zero pixel
            if (nf) {
                const int jf = __ffsll((long long)nf) - 1, jl = 63 - __clzll(nf);
                const u64 zf = ~lbits[lane * ww + jf] & (jf == ww - 1 ? lastmask : ~0ull);
                const u64 zl = ~lbits[lane * ww + jl] & (jl == ww - 1 ? lastmask : ~0ull);
                bb.minx = 64 * jf + (__ffsll((long long)zf) - 1);
                bb.maxx = 64 * jl + 63 - __clzll(zl);
                bb.miny = bb.maxy = y0 + lane;
            }
        }
        bb.minx = (int)(c3_wave_min((u32)bb.minx ^ 0x80000000u) ^ 0x80000000u);   // (DPP steps + lane reads, vp_ccl3.inl: no LDS crossbar)
        bb.maxx = (int)(c3_wave_max((u32)bb.maxx ^ 0x80000000u) ^ 0x80000000u);
        bb.miny = (int)(c3_wave_min((u32)bb.miny ^ 0x80000000u) ^ 0x80000000u);
        bb.maxy = (int)(c3_wave_max((u32)bb.maxy ^ 0x80000000u) ^ 0x80000000u);
        if (lane == 0) bgbox[sidx] = bb;
    }
    __syncthreads();
    const u32 nnz = rowoff[G.rows];
    C2_PROBE(0, 0);   // staged, masks, background box
    if (nnz == 0) {
        if (tid == 0) ncomp[sidx] = 0u;
        return;
    }
    // place of every non-zero word in the list; with one segment per word everywhere that place is also the segment's index
    for (int rr = 0; rr < G.rows; rr += 8) {
        const int r = rr + (tid >> 5);
        const u64 nzr = m_nz[r];
        for (int j0 = 0; j0 < ww; j0 += 32) {
            const int j = j0 + l32;
            if (j < ww && ((nzr >> j) & 1ull)) {
                const u32 pos = rowoff[r] + (u32)__popcll(nzr & ((1ull << j) - 1ull));
                nzlist[pos] = (unsigned short)((r << 8) | j);
                wbase[r * ww + j] = pos;
            }
        }
    }
    __syncthreads();
    // ---- index of every word's first segment ------------------------------------------------------------------------------------------
    u32 S;
    if (!any_multi) {
        S = nnz;                                       // one segment per non-zero word: wbase is already right
    } else {
        // exclusive scan of the segment counts in list order: thread = CH consecutive entries
        const u32 CH = (nnz + 255u) / 256u;
        const u32 t0 = (u32)tid * CH;
        u32 cnt = 0;
        for (u32 t = t0; t < min(t0 + CH, nnz); t++) { const u32 e = nzlist[t]; cnt += nstarts(lbits[(e >> 8) * ww + (e & 255u)]); }
        u32 inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        u32 off = 0, tot = 0;
        for (int k = 0; k < 4; k++) { if (k < wv) off += wsum[k]; tot += wsum[k]; }
        S = tot;
        u32 run = off + inc - cnt;
        if (S <= (u32)cap)
            for (u32 t = t0; t < min(t0 + CH, nnz); t++) {
                const u32 e = nzlist[t];
                const u32 i = (e >> 8) * ww + (e & 255u);
                wbase[i] = run;
                run += nstarts(lbits[i]);
            }
        if (S > (u32)cap) {                            // block-uniform
            if (tid == 0) ncomp[sidx] = C2_DENSE;
            return;
        }
        __syncthreads();
    }
    C2_PROBE(0, 1);   // word list, segment indices
    // does the first segment of word (r, j) continue a run that comes in from the left?
    auto continues = [&](int r, int j, u64 w) -> bool { return (w & 1ull) && j > 0 && ((m_b63[r] >> (j - 1)) & 1ull); };
    // index of the segment a run reaching word (r, j) from the left started with (only called when continues(r, j))
    auto run_leader = [&](int r, int j) -> u32 {
        const u64 notfull = ~m_ones[r] & ((1ull << j) - 1ull);     // words left of j that are not all ones
        const int jl = notfull ? 63 - __clzll(notfull) : -1;        // the nearest one; words jl+1 .. j-1 are all ones
        if (jl >= 0 && ((m_b63[r] >> jl) & 1ull)) { const int i = r * ww + jl; return wbase[i] + nstarts(lbits[i]) - 1u; }   // its last segment
        return wbase[r * ww + jl + 1];                              // the run starts with the all-ones word after it
    };
    // ---- parents: own index, or the run's first segment ----------------------------------------------------------------------------
    for (u32 t = tid; t < nnz; t += 256) {
        const u32 e = nzlist[t];
        const int r = (int)(e >> 8), j = (int)(e & 255u), i = r * ww + j;
        const u64 w = lbits[i];
        const u32 base = wbase[i];
        lparent[base] = continues(r, j, w) ? run_leader(r, j) : base;
        const u32 ns = nstarts(w);
        for (u32 k = 1; k < ns; k++) lparent[base + k] = base + k;
    }
    __syncthreads();
    C2_PROBE(0, 2);   // parents set, runs linked
    // ---- vertical contacts --------------------------------------------------------------------------------------------------------
    {
        u64* queue = reinterpret_cast<u64*>(tail);
        const u32 qcap = (u32)tail_words / 2u;
        auto link = [&](u32 L, u32 La) {                            // leader L of a run of this row touches leader La above (La < L)
            const u32 old = atomicMin(lparent + L, La);
            if (old != L && old != La) {                            // L already hung under another run above: those two are one component
                const u32 q = atomicAdd(&nqueue_s, 1u);
                if (q < qcap) queue[q] = ((u64)old << 32) | (u64)La;
            }
        };
        for (u32 t = tid; t < nnz; t += 256) {
            const u32 e = nzlist[t];
            const int r = (int)(e >> 8), j = (int)(e & 255u), i = r * ww + j;
            if (r == 0) continue;
            const u64 um = lbits[i - ww];
            const bool ul63 = j > 0 && ((m_b63[r - 1] >> (j - 1)) & 1ull);
            const u64 urw = j + 1 < ww ? lbits[i - ww + 1] : 0ull;
            if (!(um | (u64)ul63 | (urw & 1ull))) continue;
            const u64 w = lbits[i];
            const u32 base = wbase[i];
            const u32 ubase = um ? wbase[i - ww] : 0u;
            const u64 ustarts = um & ~(um << 1);
            const bool um_cont = um && continues(r - 1, j, um);     // the first segment of the word above is not its run's leader
            u64 rem = w;
            u32 me = base;
            while (rem) {
                const int s = __ffsll((long long)rem) - 1;
                const int en = run_end(rem, s);
                const u64 Sg = bit_range(s, en);
                rem &= ~Sg;
                const u32 L = (me == base && continues(r, j, w)) ? lparent[base] : me;   // non-leader entries are not written in this phase
                u64 c = um & (Sg | (Sg << 1) | (Sg >> 1));
                while (c) {
                    const int b = __ffsll((long long)c) - 1;
                    const u32 k = (u32)__popcll(ustarts & ((2ull << b) - 1ull)) - 1u;    // ordinal of the run of `um` that holds bit b
                    const u32 a = ubase + k;
                    link(L, (k == 0 && um_cont) ? lparent[a] : a);
                    c &= ~bit_range(b, run_end(um, b));
                }
                if ((Sg & 1ull) && ul63 && !(um & 1ull)) {          // diagonal contact with the last segment of the word above-left
                    const int il = i - ww - 1;
                    const u64 ul = lbits[il];
                    const u32 nsl = nstarts(ul);
                    const u32 a = wbase[il] + nsl - 1u;
                    link(L, (nsl == 1u && continues(r - 1, j - 1, ul)) ? lparent[a] : a);
                }
                if ((Sg >> 63) && (urw & 1ull) && !(um >> 63)) link(L, wbase[i - ww + 1]);   // above-right: that segment starts its run
                me++;
            }
        }
    }
    __syncthreads();
    C2_PROBE(0, 3);   // vertical links
    // ---- the rare real unions ---------------------------------------------------------------------------------------------------
    {
        const u32 nq = nqueue_s;                                    // block-uniform
        if (nq > (u32)tail_words / 2u) {
            if (tid == 0) ncomp[sidx] = C2_DENSE;
            return;
        }
        if (nq) {
            const u64* queue = reinterpret_cast<const u64*>(tail);
            for (u32 q = tid; q < nq; q += 256) { const u64 pr = queue[q]; lds_unite(lparent, (u32)(pr >> 32), (u32)pr); }
            __syncthreads();
        }
    }
    // ---- roots: pointer jumping, then a short walk; a root takes the next free place of the strip's (unordered) list -----------------
    u32* lidx = tail;                                               // list index under the root's segment index
    for (int round = 0; round < 5; round++)
        for (u32 ci = tid; ci < S; ci += 256) {
            const u32 pp = lparent[ci];
            const u32 g = lparent[pp];
            if (g != pp) lparent[ci] = g;                           // any value stored is an ancestor: no barrier needed between rounds
        }
    __syncthreads();
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
            u32 basek = 0;
            if (lane == 0) basek = atomicAdd(&nroots_s, (u32)__popcll(m));
            basek = __shfl(basek, 0);
            if (isroot) lidx[ci] = basek + (u32)__popcll(m & ((1ull << lane) - 1ull));
        }
    }
    __syncthreads();
    const u32 R = nroots_s;
    if (R > (u32)rc) {
        if (tid == 0) ncomp[sidx] = C2_DENSE;
        return;
    }
    for (u32 ci = tid; ci < S; ci += 256) lparent[ci] = lidx[lparent[ci]];   // segment -> place of its component in the list
    __syncthreads();
    C2_PROBE(0, 4);   // roots, list places
    if (tid < rc) {
        a_sx[tid] = 0; a_sy[tid] = 0; a_area[tid] = 0; a_key[tid] = 0xffffffffu;
        a_minx[tid] = INT_MAX; a_maxx[tid] = INT_MIN; a_miny[tid] = INT_MAX; a_maxy[tid] = INT_MIN;
    }
    __syncthreads();
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
    // ---- statistics and numbering key (smallest segment id) of every component; component index of every word's first segment
    // (dense) and of the segments of words that hold several ----------------------------------------------------------------------
    {
        u32* wc = wordcomp + (size_t)frame * G.h * ww;
        u32* sc = segcomp + (size_t)frame * G.nids;
        const u32 NONE = 0xffffffffu;
        for (u32 t0 = 0; t0 < nnz; t0 += 256) {
            const u32 t = t0 + tid;
            contrib c0;
            contrib_zero(c0);
            c0.pad = NONE;
            u32 k0 = NONE;
            u64 rem = 0;
            int j = 0, y = 0;
            u32 nextseg = 0;
            if (t < nnz) {
                const u32 e = nzlist[t];
                const int r = (int)(e >> 8);
                j = (int)(e & 255u);
                y = y0 + r;
                const int i = r * ww + j;
                const u64 w = lbits[i];
                const u32 base = wbase[i];
                const int s = __ffsll((long long)w) - 1;
                const int en = run_end(w, s);
                rem = w & ~bit_range(s, en);
                k0 = lparent[base];
                nextseg = base + 1u;
                wc[(size_t)y * ww + j] = k0;
                const u32 id = seg_id(G, y, 64 * j + s);
                if (rem) sc[id] = k0;
                const u32 len = (u32)(en - s + 1);
                const int xs = 64 * j + s, xe = 64 * j + en;
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
                    c3_wave_combine(c0);
                    c0.pad = c3_wave_min(c0.pad);
                    if (lane == lead) acc_add(ref, c0);
                } else if (k0 != NONE) {
                    acc_add(k0, c0);
                }
            }
            while (rem) {
                const int s = __ffsll((long long)rem) - 1;
                const int en = run_end(rem, s);
                rem &= ~bit_range(s, en);
                const u32 k = lparent[nextseg++];
                const u32 id = seg_id(G, y, 64 * j + s);
                sc[id] = k;
                contrib c;
                const u32 len = (u32)(en - s + 1);
                const int xs = 64 * j + s, xe = 64 * j + en;
                c.area = len; c.sx = (u64)len * (u64)(xs + xe) / 2ull; c.sy = (u64)len * (u64)y;
                c.minx = xs; c.maxx = xe; c.miny = c.maxy = y; c.pad = id;
                acc_add(k, c);
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
                                                           int32_t* __restrict__ stats, double* __restrict__ cent, int max_labels,
                                                           u32* __restrict__ c3_ncrowded, u32* __restrict__ c3_clist, c3_state* __restrict__ c3_st,
                                                           u32* __restrict__ c3_barr, int c3_strips)
{
    __shared__ u32 sbase[C2_MAXSTRIPS + 1];
    __shared__ u32 wtot[C2_THREADS / 64];
    __shared__ u32 par[C2_MCAP];
    __shared__ u32 lab[C2_MCAP];
    __shared__ u32 key[C2_MCAP];    // numbering key (smallest segment id); after the unions a root's entry holds its component's
    __shared__ u32 rkeys[C2_MCAP];  // keys of the roots, compacted (few roots) or grouped by the strip their key lies in (many)
    __shared__ u32 bbase[C2_MAXSTRIPS + 1], bfill[C2_MAXSTRIPS + 1];   // roots per key strip: exclusive prefix, fill pointer
    __shared__ u32 a_area[C2_MCAP];
    __shared__ int a_minx[C2_MCAP], a_maxx[C2_MCAP], a_miny[C2_MCAP], a_maxy[C2_MCAP];
    __shared__ u64 a_sx[C2_MCAP], a_sy[C2_MCAP];
    __shared__ u64 tot_sx, tot_sy;
    __shared__ u32 tot_area;
    __shared__ c2_box bgs;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int ww = G.ww;
    const u64* fb = bits + (size_t)f * G.h * ww;
    const u32* wc = wordcomp + (size_t)f * G.h * ww;
    const u32* sc = segcomp + (size_t)f * G.nids;
    C2_PROBE_BEGIN;
    // Everything that does not depend on the list sizes is requested first, so that it travels with them: the strips' background
    // boxes and, per thread, the first boundary word it will unite across (bit words of the two rows and their component indices).
    const u32 nc_raw = tid < strips ? ncomp[(size_t)f * strips + tid] : 0u;
    c2_box mybox = {INT_MAX, INT_MIN, INT_MAX, INT_MIN};
    if (tid < strips) mybox = bgbox[(size_t)f * strips + tid];
    const int items = (strips - 1) * ww;
    u64 p_w = 0, p_um = 0, p_ul = 0, p_ur = 0;
    u32 p_ks = 0, p_kum = 0, p_kul = 0, p_kur = 0;
    int p_b = 0, p_j = 0;
    if (tid < items) {
        p_b = tid / ww; p_j = tid - p_b * ww;
        const size_t idx = (size_t)(p_b + 1) * G.rows * ww + p_j;
        p_w = fb[idx]; p_um = fb[idx - ww];
        p_ks = wc[idx]; p_kum = wc[idx - ww];
        if (p_j > 0) { p_ul = fb[idx - ww - 1]; p_kul = wc[idx - ww - 1]; }
        if (p_j + 1 < ww) { p_ur = fb[idx - ww + 1]; p_kur = wc[idx - ww + 1]; }
    }
    // list sizes -> offsets
    const bool dense = nc_raw == C2_DENSE;
    u32 C;
    const u32 ex = c2_block_scan_excl(dense ? 0u : nc_raw, wtot, &C);
    if (tid <= strips) sbase[tid] = ex;     // tid == strips holds the total (its own value is 0)
    if (tid == 0) { tot_sx = 0; tot_sy = 0; tot_area = 0; bgs.minx = INT_MAX; bgs.maxx = INT_MIN; bgs.miny = INT_MAX; bgs.maxy = INT_MIN; }
    const bool any_dense = __syncthreads_or(dense);
    if (any_dense || C > (u32)mcap) {
        // handed over: to the crowded-frame kernels (c3_strips > 0: their per-frame counters start from zero, the frame joins their
        // list) or to the one-level kernels
        if (c3_strips > 0)
            for (int k = tid; k < 2 * (c3_strips + 1); k += C2_THREADS) c3_barr[(size_t)f * 3 * (c3_strips + 1) + k] = 0u;   // arrivals per boundary, boundaries done per strip
        if (tid == 0) {
            crowded[f] = 1u;
            if (c3_strips > 0) {
                c3_state z;
                z.bdone = 0; z.ddone = 0; z.fg_area = 0; z.pad = 0; z.fg_sx = 0; z.fg_sy = 0;
                z.bg_minx = INT_MAX; z.bg_maxx = INT_MIN; z.bg_miny = INT_MAX; z.bg_maxy = INT_MIN;
                c3_st[f] = z;
                c3_clist[atomicAdd(c3_ncrowded, 1u)] = (u32)f;
            }
        }
        return;
    }
    if (tid == 0) crowded[f] = 0u;
    C2_PROBE(1, 0);   // list sizes read and scanned
    // component records -> LDS: component i = tid + q * C2_THREADS sits at place i - sbase[s] of the list of the strip s it falls in
    u32 slot[C2_PER];                        // strip * C2_RC + place, for the label table
#pragma unroll
    for (int q = 0; q < C2_PER; q++) {
        const u32 i = (u32)tid + (u32)q * C2_THREADS;
        slot[q] = 0xffffffffu;
        if (i < C) {
            int lo = 0, hi = strips;         // largest s with sbase[s] <= i
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sbase[mid] <= i) lo = mid; else hi = mid; }
            slot[q] = (u32)lo * C2_RC + (i - sbase[lo]);
        }
    }
    contrib rec[C2_PER];
#pragma unroll
    for (int q = 0; q < C2_PER; q++)
        if (slot[q] != 0xffffffffu) rec[q] = recs[(size_t)f * strips * C2_RC + slot[q]];
#pragma unroll
    for (int q = 0; q < C2_PER; q++) {
        const u32 i = (u32)tid + (u32)q * C2_THREADS;
        if (slot[q] != 0xffffffffu) {
            par[i] = i;
            key[i] = rec[q].pad;
            a_area[i] = rec[q].area; a_minx[i] = rec[q].minx; a_maxx[i] = rec[q].maxx; a_miny[i] = rec[q].miny; a_maxy[i] = rec[q].maxy;
            a_sx[i] = rec[q].sx; a_sy[i] = rec[q].sy;
        }
    }
    if (tid < strips && mybox.minx != INT_MAX) {
        atomicMin(&bgs.minx, mybox.minx); atomicMax(&bgs.maxx, mybox.maxx);
        atomicMin(&bgs.miny, mybox.miny); atomicMax(&bgs.maxy, mybox.maxy);
    }
    __syncthreads();
    C2_PROBE(1, 1);   // records in LDS
    // unions across the strip boundaries: one thread per word of the first row of strips 1 ..
    {
        auto unite_word = [&](int b, int j, u64 w, u64 um, u64 ul, u64 ur, u32 k_self, u32 k_um, u32 k_ul, u32 k_ur) {
            if (!w || !(um | (ul >> 63) | (ur & 1ull))) return;
            const int y = (b + 1) * G.rows;
            const u32 lo = sbase[b], hi = sbase[b + 1];
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
        };
        if (tid < items) unite_word(p_b, p_j, p_w, p_um, p_ul, p_ur, p_ks, p_kum, p_kul, p_kur);
        for (int t = tid + C2_THREADS; t < items; t += C2_THREADS) {
            const int b = t / ww, j = t - b * ww;
            const size_t idx = (size_t)(b + 1) * G.rows * ww + j;
            const u64 w = fb[idx];
            const u64 um = fb[idx - ww];
            const u64 ul = j > 0 ? fb[idx - ww - 1] : 0ull;
            const u64 ur = j + 1 < ww ? fb[idx - ww + 1] : 0ull;
            if (!w || !(um | (ul >> 63) | (ur & 1ull))) continue;
            unite_word(b, j, w, um, ul, ur, wc[idx], um ? wc[idx - ww] : 0u, (ul >> 63) ? wc[idx - ww - 1] : 0u, (ur & 1ull) ? wc[idx - ww + 1] : 0u);
        }
    }
    __syncthreads();
    C2_PROBE(1, 2);   // boundary unions
    // roots (read-only walks; every thread stores the root over its own entries), key of a component = smallest key of its members
    u32 root[C2_PER];
    u32 nroot_mine = 0;
#pragma unroll
    for (int q = 0; q < C2_PER; q++) {
        const u32 i = (u32)tid + (u32)q * C2_THREADS;
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
    // cv2's label of a component = 1 + number of components with a smaller key (keys are distinct).  With a handful of roots every
    // root counts over all keys.  A speckled frame has hundreds (S1 with a loose threshold: 600 - 970), and R x R comparisons were 24 us
    // of this launch: a key is a segment id, ids grow with the row, so the keys are grouped by the strip they lie in first - a root
    // then counts the roots of earlier strips (a prefix) plus the smaller keys of its own group (R / strips of them).
    const bool grouped = R > 64u;                        // block-uniform
    const u32 ips = (u32)G.rows * (u32)G.wb;              // segment ids per strip, in either numbering
    if (!grouped) {
#pragma unroll
        for (int q = 0; q < C2_PER; q++) {
            const u32 i = (u32)tid + (u32)q * C2_THREADS;
            if (i < C && root[q] == i) rkeys[pos++] = key[i];
        }
        __syncthreads();
        C2_PROBE(1, 3);   // roots, keys, compacted
#pragma unroll
        for (int q = 0; q < C2_PER; q++) {
            const u32 i = (u32)tid + (u32)q * C2_THREADS;
            const bool isroot = i < C && root[q] == i;
            if (__any(isroot)) {
                const u32 mine = isroot ? key[i] : 0u;
                u32 cnt = 0;
                for (u32 j = 0; j < R; j++) cnt += rkeys[j] < mine ? 1u : 0u;
                if (isroot) lab[i] = cnt + 1u;
            }
        }
    } else {
        for (int k = tid; k <= strips; k += C2_THREADS) bfill[k] = 0u;
        __syncthreads();
        u32 grp[C2_PER];
#pragma unroll
        for (int q = 0; q < C2_PER; q++) {
            const u32 i = (u32)tid + (u32)q * C2_THREADS;
            grp[q] = 0xffffffffu;
            if (i < C && root[q] == i) { grp[q] = min(key[i] / ips, (u32)strips - 1u); atomicAdd(bfill + grp[q], 1u); }
        }
        __syncthreads();
        u32 tot_unused;
        const u32 cnt_mine = tid < strips ? bfill[tid] : 0u;          // (strips <= C2_MAXSTRIPS <= C2_THREADS)
        const u32 ex2 = c2_block_scan_excl(cnt_mine, wtot, &tot_unused);
        if (tid <= strips) { bbase[tid] = ex2; bfill[tid] = ex2; }    // tid == strips: the total
        __syncthreads();
#pragma unroll
        for (int q = 0; q < C2_PER; q++) {
            const u32 i = (u32)tid + (u32)q * C2_THREADS;
            if (grp[q] != 0xffffffffu) rkeys[atomicAdd(bfill + grp[q], 1u)] = key[i];
        }
        __syncthreads();
        C2_PROBE(1, 3);   // roots, keys, grouped by strip
#pragma unroll
        for (int q = 0; q < C2_PER; q++) {
            const u32 i = (u32)tid + (u32)q * C2_THREADS;
            if (grp[q] != 0xffffffffu) {
                const u32 mine = key[i], j0 = bbase[grp[q]], j1 = bbase[grp[q] + 1];
                u32 cnt = j0;
                for (u32 j = j0; j < j1; j++) cnt += rkeys[j] < mine ? 1u : 0u;
                lab[i] = cnt + 1u;
            }
        }
    }
    __syncthreads();
    C2_PROBE(1, 4);   // ranks
    // statistics of absorbed components move to their roots; (strip, component) -> label table
#pragma unroll
    for (int q = 0; q < C2_PER; q++) {
        const u32 i = (u32)tid + (u32)q * C2_THREADS;
        if (i < C) {
            const u32 r = root[q];
            complabel[(size_t)f * strips * C2_RC + slot[q]] = lab[r];
            if (r != i) {
                atomicAdd(a_area + r, a_area[i]);
                atomicAdd((unsigned long long*)(a_sx + r), (unsigned long long)a_sx[i]);
                atomicAdd((unsigned long long*)(a_sy + r), (unsigned long long)a_sy[i]);
                atomicMin(a_minx + r, a_minx[i]); atomicMax(a_maxx + r, a_maxx[i]);
                atomicMin(a_miny + r, a_miny[i]); atomicMax(a_maxy + r, a_maxy[i]);
            }
        }
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
            const u32 i = (u32)tid + (u32)q * C2_THREADS;
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
    // rows past the last label read as zeros
    for (int l = nl + tid; l < max_labels; l += C2_THREADS) {
        const size_t o = (size_t)f * max_labels + l;
        if (stats) { int32_t* sp = stats + o * 5; sp[0] = sp[1] = sp[2] = sp[3] = sp[4] = 0; }
        if (cent) { cent[o * 2] = 0.0; cent[o * 2 + 1] = 0.0; }
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
    C2_PROBE(1, 6);   // rows
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
template <int VARIANT>
__global__ __launch_bounds__(256, 8) void k_ccl2_write(const u64* __restrict__ bits, ccl_geom G, int strips, int rc, const u32* __restrict__ segcomp,
                                                    const u32* __restrict__ wordcomp, const u32* __restrict__ complabel,
                                                    const u32* __restrict__ crowded, int32_t* __restrict__ labels, u32 gpr, u32 gpr_magic,
                                                    int crowded_elsewhere)
{
    __shared__ u32 ltab[C2_RC];
    if (crowded_elsewhere && crowded[blockIdx.y]) return;   // block-uniform: the crowded-frame kernels wrote this frame's labels themselves
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
        if (VARIANT == 3) __syncthreads();
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
