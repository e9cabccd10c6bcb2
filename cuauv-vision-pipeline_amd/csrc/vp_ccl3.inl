// Labelling of CROWDED frames (included by vp_ccl.hip): masks the two-level path of vp_ccl2.inl hands over - speckle, raw noise, a
// threshold mask nobody cleaned (modules/red_buoy.py:38 runs its contour stage on the un-cleaned mask).
//
// Round 1's one-level kernels resolved such frames with arrays indexed by segment id in global memory: strips with more segments
// than a 962-entry LDS union-find fell back to compare-and-swap unions in global memory (5.9 ms per 128 frames of 50 % noise), and
// every segment paid four dependent uncoalesced reads plus seven global atomics for its statistics (1.9 - 3.5 ms).  Here:
//
//   * strips are R rows with R * ceil(w/2) <= 16384 segment ids (1080p: 16 rows), so a strip's union-find ALWAYS fits in LDS - it is
//     indexed by the strip-relative segment id itself, no compaction, no capacity fallback; links point at the smaller id, so a
//     strip-local root is its component's smallest id (cv2's numbering key, section 4.3 of DESIGN.md);
//   * contacts between segments are bits of three masks per word, walked by a thread per half-word; most are settled by one
//     atomicMin (first links), the rest by compare-and-swap unions;
//   * what leaves the strip is one contiguous block of u16 "root of every segment id" (30 KB) and the bits of its local roots;
//   * strips meet at their boundaries through a global union-find over LOCAL ROOTS only; a root that absorbs another is marked
//     "has members elsewhere";
//   * ranks (= labels) come from the root bitmap + popcount prefix as before;
//   * a second pass per strip reloads the strip's u16 block into LDS and does everything else there: labels of the local roots
//     (from the strip's own slice of bitmap + prefix; only absorbed roots walk global memory), statistics accumulated in LDS per
//     local component, rows of components that are complete within the strip written straight to the output (no atomics), the rest
//     added to per-label accumulators, and the strip's part of the label image stored from LDS.
//
// Five short launches, each a fixed number of blocks that loop over (frame, strip) items of the frames k_ccl2_merge handed over and
// leave at once when there are none: k_ccl3_link (strips), k_ccl3_bound (boundaries), k_ccl3_rank (root counts and prefixes per slice,
// absorbed roots flattened), k_ccl3_label (labels, statistics, label image), k_ccl3_rows (rows of components that span strips).  A
// first form chained the stages inside two launches with arrival counters ("the strip that finishes second does the boundary, the
// block that finishes a frame's last boundary ranks the frame"): every per-frame stage then ran on ONE block (ranks 3 ms, rows 6 ms
// per 128 frames), agent-scope fences around every hand-over cost another 2.5 ms, and second arrivers were 1.7 x busier than the
// average block.  Kernel boundaries are the cheaper seam here.
#define C3_IDS 16384           // most segment ids per strip = entries of the LDS union-find (1080p: 16 rows = 15,360; VP_C3_IDS=8192: 8 rows)
#define C3_LINK_THREADS 512    // k_ccl3_link, strips of up to 8192 ids; twice that for taller ones: a thread per 32-bit HALF of a word
#define C3_LABEL_THREADS 512   // k_ccl3_label, likewise
#define C3_ACC 2304            // local components whose statistics are accumulated per pass over the strip (the labelling launch has a CU's LDS to itself either way: one pass for raw noise at 10 % and 50 %)
#define C3_LIGHT_ROOTS 1024    // strips with at most this many local roots are labelled by the LIGHT instantiation of k_ccl3_label: tables for that many roots
#define C3_LIGHT_ACC 768       // ... and accumulators for that many per pass: ~72 KB of LDS, two blocks per CU hide each other's fixed latencies
#define C3_TAB 128             // entries of a strip's table of partial components (beyond it: straight to global memory)
#define C3_MAX_STRIPS 512      // per-strip root counts of a frame are scanned in LDS by every block of the later launches

#ifdef VP_PROBE   // measurement builds only: time per phase (100 MHz wall clock ticks), summed over a block's items
__device__ unsigned long long g_c3_probe[3][2048][16];
#define C3_PROBE_DECL __shared__ unsigned long long pr_acc[16]; unsigned long long pr_t = 0; if (threadIdx.x == 0) { for (int q_ = 0; q_ < 16; q_++) pr_acc[q_] = 0; pr_t = wall_clock64(); }
#define C3_PROBE(i) do { if (threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); pr_acc[i] += n_ - pr_t; pr_t = n_; } } while (0)
#define C3_PROBE_END(k) do { if (threadIdx.x == 0 && blockIdx.x < 2048) for (int q_ = 0; q_ < 16; q_++) g_c3_probe[k][blockIdx.x][q_] = pr_acc[q_]; } while (0)
#else
#define C3_PROBE_DECL do { } while (0)
#define C3_PROBE(i) do { } while (0)
#define C3_PROBE_END(k) do { } while (0)
#endif

struct c3_state {              // per frame; zeroed by k_ccl2_merge when it hands the frame over
    u32 bdone, ddone;          // boundaries processed / strips labelled
    u32 fg_area, pad;
    u64 fg_sx, fg_sy;
    int bg_minx, bg_maxx, bg_miny, bg_maxy;
};

struct c3_plan {
    int R, strips;             // rows per strip (even), strips per frame
    u32 ids;                   // R * wb: multiple of 32, <= C3_IDS
    int ok;
};

static c3_plan c3_make_plan(const ccl_geom& G, u32 max_ids)
{
    c3_plan P = {0, 0, 0, 0};
    for (int R = 32; R >= 2; R >>= 1) {
        const u32 ids = (u32)R * (u32)G.wb;
        if (ids <= max_ids && (ids % 32u) == 0 && R * G.ww <= 512) { P.R = R; P.ids = ids; break; }
    }
    if (!P.R || G.ww > 64) return P;
    P.strips = (G.h + P.R - 1) / P.R;
    P.ok = P.strips <= C3_MAX_STRIPS ? 1 : 0;
    return P;
}
static size_t c3_max_strips(int h) { return (size_t)(h + 1) / 2 + 1; }

// strip-relative segment id of the segment starting at pixel x of strip row r (strips start at even rows, so r and y have the same parity)
__device__ __forceinline__ u32 c3_rel(const ccl_geom& G, int r, int x)
{
    if (G.numbering == VP_CCL_BLOCK2X2) return (((u32)(r >> 1) * (u32)G.wb + (u32)(x >> 1)) << 1) | (u32)(r & 1);
    return (u32)r * (u32)G.wb + (u32)(x >> 1);
}

// links the larger root under the smaller (global memory, device-scope CAS): the absorbed root loses its bit in the root bitmap,
// the absorbing one is marked as having members outside its own strip-local component
__device__ __forceinline__ void c3_unite(u32* p, u32* flags, u32* child, u32 a, u32 b)
{
    // both walks advance together: every step is a round trip to the memory side (agent-scope loads pass the L2 of this die), and the
    // two chains do not depend on each other
    u32 qa = ld_rlx(p + a), qb = ld_rlx(p + b);
    for (;;) {
        while (qa != a || qb != b) {                          // a step: the grandparents of both, the entry left behind now skips one
            const bool ma = qa != a, mb = qb != b;
            const u32 ga = ma ? ld_rlx(p + qa) : qa, gb = mb ? ld_rlx(p + qb) : qb;
            if (ma) { if (ga != qa) st_rlx(p + a, ga); a = qa; qa = ga; }
            if (mb) { if (gb != qb) st_rlx(p + b, gb); b = qb; qb = gb; }
        }
        if (a == b) return;
        if (a < b) { const u32 t = a; a = b; b = t; const u32 tq = qa; qa = qb; qb = tq; }
        const u32 old = atomicCAS(p + a, a, b);
        if (old == a) {
            atomicAnd(flags + (a >> 5), ~(1u << (a & 31)));
            // (the big component's root takes a link from every fragment of every strip: set its mark once - read-modify-writes of one
            // address queue up in L2, reads of it do not)
            if (!(ld_rlx(child + (b >> 5)) & (1u << (b & 31)))) atomicOr(child + (b >> 5), 1u << (b & 31));
            return;
        }
        qa = old;                                             // somebody linked `a` first: go on from where it points now
        qb = ld_rlx(p + b);
    }
}

// A contact between two segments of a strip (LDS, strip-relative ids).  Most contacts are the FIRST link of their larger end: one
// atomicMin on parent[larger] settles them without a find (parents only ever decrease, so the pointers stay a forest).  When the
// larger id already hung under something else, that something and the smaller id are one component: a real union
// (compare-and-swap union-find with its finds) - a third of the contacts in raw noise instead of all of them.
__device__ __forceinline__ void c3_contact(u32* lpar, u32 a, u32 b, int dbg = 0)
{
    const u32 hi = max(a, b), lo0 = min(a, b);
    if (dbg & 2) { if (hi == 0xfffffff0u) lpar[0] = lo0; return; }     // (timing experiments: the walk over the contacts alone)
    // ... under where the smaller end points NOW, up to four levels up (any ancestor of it will do): the rows of a strip are linked
    // all at once, so the forest of first links would otherwise be as deep as the strip has rows, and the finds of the real unions walk
    // it (strip union-find of 50 % noise: 1,055 -> 820 us per 128 frames)
    u32 lo = lo0;
#pragma unroll
    for (int hop = 0; hop < 4; hop++) {
        const u32 q = lpar[lo];
        if (q == lo) break;
        lo = q;
    }
    const u32 old = atomicMin(lpar + hi, lo);
    if (old != hi && old != lo && !(dbg & 4)) lds_unite(lpar, old, lo);  // (dbg 4: ... with the first links, without the real unions)
}

// first bit of the run of 1s of `w` that holds set bit x, on 32-bit halves (64-bit shifts and counts run at a quarter of the rate)
__device__ __forceinline__ int c3_run_start(u64 w, int x)
{
    const u32 lo = (u32)w, hi = (u32)(w >> 32);
    if (x >= 32) {
        const u32 z = ~hi & ((1u << (x - 32)) - 1u);
        if (z) return 64 - __clz(z);
        const u32 zl = ~lo;
        return zl ? 32 - __clz(zl) : 0;
    }
    const u32 z = ~lo & ((1u << x) - 1u);
    return z ? 32 - __clz(z) : 0;
}

// last bit of the run of 1s of `w` that starts at bit sb, on 32-bit halves
__device__ __forceinline__ int c3_run_end(u64 w, int sb)
{
    const u32 lo = (u32)w, hi = (u32)(w >> 32);
    if (sb < 32) {
        const u32 z = ~(lo >> sb);                      // bit 0 is clear; the zeros shifted in at the top end the count at 32 - sb
        const int n = z ? __ffs((int)z) - 1 : 32;       // 32 - sb: the run reaches the upper half
        if (n < 32 - sb) return sb + n - 1;
        const u32 zh = ~hi;
        return zh ? 32 + (__ffs((int)zh) - 1) - 1 : 63;
    }
    const u32 z = ~(hi >> (sb - 32));
    const int n = z ? __ffs((int)z) - 1 : 32;
    return (n < 64 - sb) ? sb + n - 1 : 63;
}

// Wave-wide reductions to a wave-uniform value without the LDS crossbar: four DPP steps make every row of 16 lanes uniform (quad
// swaps, then the mirrored half-row and row - any lane of the other half serves once the halves are uniform), four lane reads
// combine the rows.  (__shfl_xor is a ds_bpermute: 36 of them per turn of the statistics loop kept the LDS pipe busier than the
// atomics they save.)  Every lane of the wave must be active.
template <int CTRL>
__device__ __forceinline__ u32 c3_dpp(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
#define C3_WAVE_REDUCE(name, OP)                                                                                   \
    __device__ __forceinline__ u32 name(u32 v)                                                                     \
    {                                                                                                              \
        v = OP(v, c3_dpp<0xB1>(v)); v = OP(v, c3_dpp<0x4E>(v)); v = OP(v, c3_dpp<0x141>(v)); v = OP(v, c3_dpp<0x140>(v)); \
        const u32 r0 = (u32)__builtin_amdgcn_readlane((int)v, 0), r1 = (u32)__builtin_amdgcn_readlane((int)v, 16),  \
                  r2 = (u32)__builtin_amdgcn_readlane((int)v, 32), r3 = (u32)__builtin_amdgcn_readlane((int)v, 48); \
        return OP(OP(r0, r1), OP(r2, r3));                                                                         \
    }
#define C3_OP_ADD(a, b) ((a) + (b))
#define C3_OP_MIN(a, b) min((a), (b))
#define C3_OP_MAX(a, b) max((a), (b))
#define C3_OP_OR(a, b) ((a) | (b))
C3_WAVE_REDUCE(c3_wave_add, C3_OP_ADD)
C3_WAVE_REDUCE(c3_wave_min, C3_OP_MIN)
C3_WAVE_REDUCE(c3_wave_max, C3_OP_MAX)
C3_WAVE_REDUCE(c3_wave_or, C3_OP_OR)

template <int CTRL>
__device__ __forceinline__ u64 c3_dpp64(u64 v) { return ((u64)c3_dpp<CTRL>((u32)(v >> 32)) << 32) | (u64)c3_dpp<CTRL>((u32)v); }
__device__ __forceinline__ u64 c3_wave_add64(u64 v)
{
    v += c3_dpp64<0xB1>(v); v += c3_dpp64<0x4E>(v); v += c3_dpp64<0x141>(v); v += c3_dpp64<0x140>(v);
    u64 r = 0;
#pragma unroll
    for (int q = 0; q < 64; q += 16)
        r += ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), q) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)(u32)v, q);
    return r;
}
// every lane gets the wave's combined record (all lanes active)
__device__ __forceinline__ void c3_wave_combine(contrib& c)
{
    c.area = c3_wave_add(c.area);
    c.sx = c3_wave_add64(c.sx);
    c.sy = c3_wave_add64(c.sy);
    c.minx = (int)(c3_wave_min((u32)c.minx ^ 0x80000000u) ^ 0x80000000u);      // signed order through the unsigned reductions
    c.maxx = (int)(c3_wave_max((u32)c.maxx ^ 0x80000000u) ^ 0x80000000u);
    c.miny = (int)(c3_wave_min((u32)c.miny ^ 0x80000000u) ^ 0x80000000u);
    c.maxy = (int)(c3_wave_max((u32)c.maxy ^ 0x80000000u) ^ 0x80000000u);
}

// adds a partial component to the strip's table (LDS), keyed by label; a full table sends it straight to the frame's accumulators
__device__ __forceinline__ void c3_table_add(u32* t_label, contrib* t_rec, u32 label, const contrib& c, ccl_acc* facc)
{
    u32 slot = (label * 2654435761u) >> 25;                   // 7 bits
    for (int probe = 0; probe < 8; probe++, slot = (slot + 1) & (C3_TAB - 1)) {
        const u32 cur = atomicCAS(t_label + slot, 0u, label);
        if (cur == 0u || cur == label) {
            contrib* t = t_rec + slot;
            atomicAdd(&t->area, c.area);
            atomicAdd((unsigned long long*)&t->sx, (unsigned long long)c.sx);
            atomicAdd((unsigned long long*)&t->sy, (unsigned long long)c.sy);
            atomicMin(&t->minx, c.minx); atomicMax(&t->maxx, c.maxx);
            atomicMin(&t->miny, c.miny); atomicMax(&t->maxy, c.maxy);
            return;
        }
    }
    acc_commit(facc + label, c);
}

// every (row, word) of the strip, one word per thread and pass
#define C3_FOR_WORDS(r, j, i, NT)                                                                  \
    for (int i = threadIdx.x, r = i / ww, j = i - r * ww; i < nrows * ww; i += NT, r = i / ww, j = i - r * ww)

// ---- ranks without a pass over the frame ---------------------------------------------------------------------------------------
// A strip's slice of the root bitmap is final once the boundaries above and below it have been processed; the block that completes
// the second of them counts the slice's roots (scount) and stores the exclusive popcount prefix of its words RELATIVE to the slice.
// A label is then  1 + (roots in the strips before) + (relative prefix of the word) + (roots below in the word):  the first term is
// a scan over at most a few hundred per-strip counts, which every block of the later launches does for itself.
template <int NT>
__device__ void c3_finish_slice(const ccl_geom& G, const c3_plan& P, int strip, const u32* __restrict__ ffl, u32* __restrict__ fpf,
                                u32* __restrict__ scount, u32* red /* LDS, NT / 64 + 1 words */)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 k0 = (u32)strip * (P.ids / 32), k1 = (strip == P.strips - 1) ? G.nw32 : min(k0 + P.ids / 32, G.nw32);
    u32 run = 0;
    for (u32 kb = k0; kb < k1; kb += NT) {                   // one pass for every supported geometry (a slice is at most 256 words)
        const u32 k = kb + tid;
        const u32 c = k < k1 ? (u32)__popc(ld_rlx(ffl + k)) : 0u;
        u32 inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        __syncthreads();
        if (lane == 63) red[wv] = inc;
        __syncthreads();
        u32 off = 0, tot = 0;
        for (int q = 0; q < NT / 64; q++) { const u32 t = red[q]; if (q < wv) off += t; tot += t; }
        if (k < k1) fpf[k] = run + off + inc - c;
        run += tot;
    }
    if (tid == 0) scount[strip] = run;
}

// exclusive prefix of the frame's per-strip root counts into LDS (sb[0 .. strips], sb[strips] = all roots of the frame)
template <int NT>
__device__ void c3_strip_bases(const c3_plan& P, const u32* __restrict__ scount, u32* sb, u32* red)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    u32 run = 0;
    for (int kb = 0; kb < P.strips; kb += NT) {
        const int k = kb + tid;
        const u32 c = k < P.strips ? scount[k] : 0u;
        u32 inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        __syncthreads();
        if (lane == 63) red[wv] = inc;
        __syncthreads();
        u32 off = 0, tot = 0;
        for (int q = 0; q < NT / 64; q++) { const u32 t = red[q]; if (q < wv) off += t; tot += t; }
        if (k < P.strips) sb[k] = run + off + inc - c;
        run += tot;
    }
    if (tid == 0) sb[P.strips] = run;
    __syncthreads();
}

// ---- K1: strip-local union-find, boundaries, ranks --------------------------------------------------------------------------------
// dynamic LDS: lbits[R * ww] u64 | lpar[ids] u32 | lroots[ids / 32] u32
template <int LT>
__global__ __launch_bounds__(LT, 8) void k_ccl3_link(const u64* __restrict__ bits, ccl_geom G, c3_plan P, const u32* __restrict__ ncrowded,
                                                               const u32* __restrict__ clist, u32* __restrict__ parent, u32* __restrict__ flags,
                                                               u32* __restrict__ child, u32* __restrict__ lrootbits, u32* __restrict__ root16,
                                                               ccl_acc* __restrict__ acc, int max_labels, int acc_clear, int dbg)
{
    const u32 nc = *ncrowded;
    if (nc == 0) return;                                      // the common case: nothing was handed over
    extern __shared__ __attribute__((aligned(16))) u64 c3_lds[];
    const int NT = LT;
    const int ww = G.ww, tid = threadIdx.x;
    const int nwmax = P.R * ww;
    u64* lbits = c3_lds;
    u32* lpar = reinterpret_cast<u32*>(c3_lds + nwmax);
    u32* lroots = lpar + P.ids;
    const u32 total = nc * (u32)P.strips;
    C3_PROBE_DECL;
    for (u32 item = blockIdx.x; item < total; item += gridDim.x) {
        const u32 f = clist[item / (u32)P.strips];
        const int s = (int)(item % (u32)P.strips);
        const int y0 = s * P.R;
        const int nrows = min(P.R, G.h - y0);
        const u64* fb = bits + (size_t)f * G.h * ww;
        u32* fpar = parent + (size_t)f * G.nids;
        u32* ffl = flags + (size_t)f * G.nw32;
        u32* fch = child + (size_t)f * G.nw32;
        u32* flr = lrootbits + (size_t)f * G.nw32;
        u32* f16 = root16 + (size_t)f * G.nids;               // the frame's own slot of the u32 segment array, used as two u16 per word
        const u32 base = (u32)s * P.ids;
        __syncthreads();                                      // the previous item's LDS is done with
        C3_FOR_WORDS(r, j, i, NT) lbits[i] = fb[(size_t)(y0 + r) * ww + j];
        for (u32 k = tid; k < P.ids / 32; k += NT) lroots[k] = 0u;
        __syncthreads();
        C3_PROBE(0);   // bits staged
        // every segment its own parent.  A thread takes one 32-bit half of a word in this and the two passes below (a word of noise
        // holds sixteen segments: per-segment work is serial in its thread)
        for (int i2 = tid; i2 < nrows * ww * 2; i2 += NT) {
            const int i = i2 >> 1, half = i2 & 1, r = i / ww, j = i - r * ww;
            const u64 w = lbits[i];
            u32 st = (u32)((w & ~(w << 1)) >> (32 * half));
            while (st) {
                const int sb = __ffs((int)st) - 1 + 32 * half;
                st &= st - 1;
                const u32 id = c3_rel(G, r, 64 * j + sb);
                lpar[id] = id;
            }
        }
        __syncthreads();
        C3_PROBE(1);   // parents set
        // contacts: with the segment that ends the previous word of the row, and with the 8-connected segments of the row above.
        // Every (segment, segment above) pair shows as ONE bit of three masks: the first bit of their vertical overlap; a segment's
        // last bit with a run above starting one column further (and nothing straight above); a segment's first bit with a run above
        // ending one column before.  A thread walks the bits of its half of the masks.
        for (int i2 = tid; i2 < nrows * ww * 2; i2 += NT) {
            const int i = i2 >> 1, half = i2 & 1, r = i / ww, j = i - r * ww;
            const u64 w = lbits[i];
            if (!w) continue;
            if (half == 0 && (w & 1ull) && j > 0 && (lbits[i - 1] >> 63))
                c3_contact(lpar, c3_rel(G, r, 64 * j), c3_rel(G, r, 64 * (j - 1) + c3_run_start(lbits[i - 1], 63)), dbg);
            if (r == 0) continue;
            const u64 um = lbits[i - ww];
            const u64 ul = j > 0 ? lbits[i - ww - 1] : 0ull;
            const u64 ur = j + 1 < ww ? lbits[i - ww + 1] : 0ull;
            const u64 up_r = (um >> 1) | (ur << 63), up_l = (um << 1) | (ul >> 63);      // the row above, one column to the right / left
            const u64 V = w & um;
            u32 vs = (u32)((V & ~(V << 1)) >> (32 * half));
            u32 dr = (u32)((w & ~(w >> 1) & up_r & ~um) >> (32 * half));
            u32 dl = (u32)((w & ~(w << 1) & up_l & ~um) >> (32 * half));
            while (vs) {
                const int x = __ffs((int)vs) - 1 + 32 * half;
                vs &= vs - 1;
                c3_contact(lpar, c3_rel(G, r, 64 * j + c3_run_start(w, x)), c3_rel(G, r - 1, 64 * j + c3_run_start(um, x)), dbg);
            }
            while (dr) {
                const int x = __ffs((int)dr) - 1 + 32 * half;
                dr &= dr - 1;
                c3_contact(lpar, c3_rel(G, r, 64 * j + c3_run_start(w, x)), c3_rel(G, r - 1, 64 * j + x + 1), dbg);
            }
            while (dl) {
                const int x = __ffs((int)dl) - 1 + 32 * half;
                dl &= dl - 1;
                const u32 b = x > 0 ? c3_rel(G, r - 1, 64 * j + c3_run_start(um, x - 1)) : c3_rel(G, r - 1, 64 * (j - 1) + c3_run_start(ul, 63));
                c3_contact(lpar, c3_rel(G, r, 64 * j + x), b, dbg);
            }
        }
        __syncthreads();
        C3_PROBE(2);   // contacts
        // flatten (read-only walks; every thread stores the root over its OWN entries), note the local roots
        for (int i2 = tid; i2 < nrows * ww * 2; i2 += NT) {
            const int i = i2 >> 1, half = i2 & 1, r = i / ww, j = i - r * ww;
            const u64 w = lbits[i];
            u32 st = (u32)((w & ~(w << 1)) >> (32 * half));
            while (st) {
                const int sb = __ffs((int)st) - 1 + 32 * half;
                st &= st - 1;
                const u32 id = c3_rel(G, r, 64 * j + sb);
                const u32 root = lds_root(lpar, id);
                lpar[id] = root;
                if (root == id) atomicOr(lroots + (id >> 5), 1u << (id & 31));
                // The frame's union-find (k_ccl3_bound) only ever starts from roots of segments in a strip's first or last row, and only
                // roots it united are read back later (k_ccl3_rank, k_ccl3_label): the others need no entry in global memory.  (Every
                // segment of such a row stores its root's entry: the same value from each, and 4-byte stores to scattered lines - one
                // per local root, 170 k per frame of 10 % noise - were half of what this launch wrote.)
                if (r == 0 || r == P.R - 1) fpar[base + root] = base + root;
            }
        }
        __syncthreads();
        C3_PROBE(3);   // flatten + roots
        // what leaves the strip: the root of every segment id (u16, contiguous), its slice of the root bitmap, a clean "has members" slice
        {
            const u32 lim = min(P.ids, G.nids - base);        // (the last strip may reach past the frame's id range)
            for (u32 k = tid; k < lim / 2; k += NT) f16[base / 2 + k] = (lpar[2 * k] & 0xffffu) | (lpar[2 * k + 1] << 16);
            for (u32 k = tid; k < lim / 32; k += NT) { ffl[base / 32 + k] = lroots[k]; fch[base / 32 + k] = 0u; flr[base / 32 + k] = lroots[k]; }
            if (s == P.strips - 1)                            // ids past the last strip (the frame's id range is rounded up): no roots there
                for (u32 k = (base + lim) / 32 + tid; k < G.nw32; k += NT) { ffl[k] = 0u; fch[k] = 0u; flr[k] = 0u; }
        }
        C3_PROBE(4);   // dump issued
        // this strip's share of the frame's accumulators, cleared for the labelling launch (only components that span strips use one,
        // but which labels those are is not known before the ranks are).  Not in the usual case: the context keeps a set of accumulators
        // that k_ccl3_rows hands back clean (acc_clear == 0), so nothing sized by max_labels is written per frame.
        if (acc_clear) {
            const u32 wpe = (u32)(sizeof(ccl_acc) / 4);       // words per entry: area, minx, miny, maxx, maxy, pad, sx, sy
            const u64 words = (u64)max_labels * wpe;
            const u64 per = (words + (u64)P.strips - 1) / (u64)P.strips;
            const u64 w0 = min((u64)s * per, words), w1 = min(w0 + per, words);
            u32* aw = reinterpret_cast<u32*>(acc + (size_t)f * max_labels);
            for (u64 q = w0 + tid; q < w1; q += NT) {
                const u32 e = (u32)(q % wpe);
                aw[q] = (e == 1u || e == 2u) ? (u32)INT_MAX : (e == 3u || e == 4u) ? (u32)INT_MIN : 0u;
            }
        }
        C3_PROBE(5);   // accumulators cleared
    }
    C3_PROBE_END(0);
}

// ---- K1b: the strips of a frame meet.  One item per (handed-over frame, boundary): a launch of its own, so that every boundary finds
// both its strips complete without anybody waiting, the work is spread evenly (handled by whichever strip finished second it was not:
// 1.7 x between the busiest and the average block) and plain cached loads serve. ------------------------------------------------------
// dynamic LDS: the u16 roots of the two rows that meet, two per word
__global__ __launch_bounds__(256) void k_ccl3_bound(const u64* __restrict__ bits, ccl_geom G, c3_plan P, const u32* __restrict__ ncrowded,
                                                    const u32* __restrict__ clist, u32* __restrict__ parent, u32* __restrict__ flags,
                                                    u32* __restrict__ child, const u32* __restrict__ root16, int dbg)
{
    const u32 nc = *ncrowded;
    if (nc == 0 || P.strips < 2) return;
    extern __shared__ __attribute__((aligned(16))) u64 c3_lds[];
    __shared__ u32 pairs[512];                                // 256 recently united (lower root, upper root) pairs of the boundary at hand
    const int NT = 256, tid = threadIdx.x, ww = G.ww;
    u32* stage = reinterpret_cast<u32*>(c3_lds);
    const u32 nb = (u32)(P.strips - 1);
    const u32 total = nc * nb;
    C3_PROBE_DECL;
    for (u32 item = blockIdx.x; item < total; item += gridDim.x) {
        const u32 f = clist[item / nb];
        const int b = (int)(item % nb) + 1;                   // the boundary between strips b - 1 and b
        const u64* fb = bits + (size_t)f * G.h * ww;
        u32* fpar = parent + (size_t)f * G.nids;
        u32* ffl = flags + (size_t)f * G.nw32;
        u32* fch = child + (size_t)f * G.nw32;
        const u32* f16 = root16 + (size_t)f * G.nids;
        const int y = b * P.R;
        const u32 blo = (u32)b * P.ids, bup = blo - P.ids;
        // The ids of a row and of its partner in the 2x2 numbering interleave, so a row pair is one contiguous range of 2 * wb ids:
        // the first of the lower strip, the last of the upper one.  (Pixel numbering: one row = wb ids.)
        const u32 span = (G.numbering == VP_CCL_BLOCK2X2) ? 2u * (u32)G.wb : (u32)G.wb;
        const u32 up0 = P.ids - span;                         // first id of the upper strip's last row (pair)
        const u32 upoff = (bup + up0) & 1u;                   // (an odd first id: the pair loads start one entry early)
        const u32 nlo = (span + 1) / 2, nup = (span + upoff + 1) / 2;
        u32* s_lo = stage;
        u32* s_up16 = stage + nlo;
        __syncthreads();
        {   // (all loads of a thread first: a load-then-store loop waits for each in turn)
            u32 vl[4], vu[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const u32 k = tid + (u32)q * NT; vl[q] = k < nlo ? f16[blo / 2 + k] : 0u; vu[q] = k < nup ? f16[(bup + up0 - upoff) / 2 + k] : 0u; }
#pragma unroll
            for (int q = 0; q < 4; q++) { const u32 k = tid + (u32)q * NT; if (k < nlo) s_lo[k] = vl[q]; if (k < nup) s_up16[k] = vu[q]; }
            for (u32 k = tid + 4u * NT; k < nlo; k += NT) s_lo[k] = f16[blo / 2 + k];
            for (u32 k = tid + 4u * NT; k < nup; k += NT) s_up16[k] = f16[(bup + up0 - upoff) / 2 + k];
        }
        for (u32 k = tid; k < 512; k += NT) pairs[k] = 0xffffffffu;
        __syncthreads();
        C3_PROBE(6);   // boundary rows staged
        const unsigned short* lo16 = reinterpret_cast<const unsigned short*>(s_lo);
        const unsigned short* up16 = reinterpret_cast<const unsigned short*>(s_up16) + upoff;
        // eight threads share a word of the boundary row: one byte each of the three contact masks (see k_ccl3_link)
        for (int j = tid >> 3; j < ww && !(dbg & 1); j += NT / 8) {
            const size_t idx = (size_t)y * ww + j;
            const u64 w = fb[idx];
            if (!w) continue;
            const u64 um = fb[idx - ww];
            const u64 ul = j > 0 ? fb[idx - ww - 1] : 0ull;
            const u64 ur = j + 1 < ww ? fb[idx - ww + 1] : 0ull;
            if (!(um | (ul >> 63) | (ur & 1ull))) continue;
            const int sh = 8 * (tid & 7);
            const u64 up_r = (um >> 1) | (ur << 63), up_l = (um << 1) | (ul >> 63);
            const u64 V = w & um;
            u32 vs = (u32)((V & ~(V << 1)) >> sh) & 0xffu;
            u32 dr = (u32)((w & ~(w >> 1) & up_r & ~um) >> sh) & 0xffu;
            u32 dl = (u32)((w & ~(w << 1) & up_l & ~um) >> sh) & 0xffu;
            auto meet = [&](int sb, int xup) {
                const u32 a = blo + (u32)lo16[c3_rel(G, 0, 64 * j + sb)];
                const u32 bb = bup + (u32)up16[c3_rel(G, P.R - 1, xup) - up0];
                // a pair some thread of the block has already united (the big component, over and over) is not united again
                const u32 slot = ((a * 2654435761u) ^ (bb * 40503u)) >> 24;
                if (pairs[2 * slot] == a && pairs[2 * slot + 1] == bb) return;
                pairs[2 * slot] = a; pairs[2 * slot + 1] = bb;                 // (a torn entry only costs a repeated union)
                c3_unite(fpar, ffl, fch, a, bb);
            };
            while (vs) {
                const int x = __ffs((int)vs) - 1 + sh;
                vs &= vs - 1;
                meet(c3_run_start(w, x), 64 * j + c3_run_start(um, x));
            }
            while (dr) {
                const int x = __ffs((int)dr) - 1 + sh;
                dr &= dr - 1;
                meet(c3_run_start(w, x), 64 * j + x + 1);
            }
            while (dl) {
                const int x = __ffs((int)dl) - 1 + sh;
                dl &= dl - 1;
                meet(x, x > 0 ? 64 * j + c3_run_start(um, x - 1) : 64 * (j - 1) + c3_run_start(ul, 63));
            }
        }
        __syncthreads();
        C3_PROBE(7);   // boundary unions
    }
    C3_PROBE_END(1);
}

// ---- ranks: one item per (handed-over frame, strip) once every boundary of the launch above is through ---------------------------------
// ... and every local root that was absorbed gets the frame's root as its parent (the unions are complete, so that is final): the
// labelling launch then finds the label of such a component's pieces with one look-up instead of a walk.
// ... and the strip is marked for the labelling launch that suits it: LIGHT (few local roots: raw noise at 2 % and at 50 %, where
// most of a strip is one component) or HEAVY.  klass[item] = 1 / 2; each labelling launch walks all items and takes its own (work lists
// appended to with atomics were tried: thousands of blocks on one counter queue up at the memory side - this launch took 110 us
// instead of 37 - and 64 sublists with a prefix per labelling block cost the labelling kernels registers they do not have).
__global__ __launch_bounds__(256) void k_ccl3_rank(ccl_geom G, c3_plan P, const u32* __restrict__ ncrowded, const u32* __restrict__ clist,
                                                   const u32* __restrict__ flags, u32* __restrict__ prefix, u32* __restrict__ barr,
                                                   const u32* __restrict__ lrootbits, u32* __restrict__ parent, unsigned char* __restrict__ klass)
{
    const u32 nc = *ncrowded;
    if (nc == 0) return;
    __shared__ u32 red[256 / 64 + 1];
    __shared__ u32 nlocal;
    const u32 total = nc * (u32)P.strips;
    for (u32 item = blockIdx.x; item < total; item += gridDim.x) {
        const u32 f = clist[item / (u32)P.strips];
        const int s = (int)(item % (u32)P.strips);
        c3_finish_slice<256>(G, P, s, flags + (size_t)f * G.nw32, prefix + (size_t)f * G.nw32, barr + (size_t)f * 3 * (P.strips + 1) + 2 * (P.strips + 1), red);
        {
            const u32* ffl = flags + (size_t)f * G.nw32;
            const u32* flr = lrootbits + (size_t)f * G.nw32;
            u32* fpar = parent + (size_t)f * G.nids;
            const u32 k0 = (u32)s * (P.ids / 32), k1 = (s == P.strips - 1) ? G.nw32 : min(k0 + P.ids / 32, G.nw32);
            if (threadIdx.x == 0) nlocal = 0u;
            __syncthreads();
            u32 mine = 0;
            for (u32 k = k0 + threadIdx.x; k < k1; k += 256) mine += (u32)__popc(flr[k]);
            mine = c3_wave_add(mine);
            if ((threadIdx.x & 63) == 0) atomicAdd(&nlocal, mine);
            __syncthreads();
            if (threadIdx.x == 0) klass[item] = nlocal <= (u32)C3_LIGHT_ROOTS ? (unsigned char)1 : (unsigned char)2;
            for (u32 k = k0 + threadIdx.x; k < k1; k += 256) {
                u32 m = flr[k] & ~ffl[k];                    // local roots of the strip that are no longer roots of the frame
                while (m) {
                    const int b = __ffs((int)m) - 1;
                    m &= m - 1;
                    const u32 g0 = (k << 5) + (u32)b;
                    u32 g = g0;
                    for (u32 q = fpar[g]; q != g; q = fpar[g]) g = q;   // (entries along the way may already hold the root: still an ancestor)
                    fpar[g0] = g;
                }
            }
        }
        __syncthreads();
    }
}

// ---- K2: labels, statistics, label image ------------------------------------------------------------------------------------------
// One item per (handed-over frame, strip).  Everything about the strip is in LDS: its bits, the local rank of every segment's root
// (u16; the link launch left root ids, rewritten here once the roots are ranked), the label and the id of every local root by rank,
// the strip's slices of the bitmaps, statistics accumulators for C3_ACC components at a time, a small table of partial components.
// dynamic LDS: lbits[R * ww] u64 | lab[nrmax] u32 | lr16[ids] u16 | rid[nrmax] u16 | lrb, lrp, gfl, gpf, gch [ids / 32] u32 | 5 x ACCN u32
// Two instantiations share the items (k_ccl3_rank sorts the strips into two lists): NRCAP = 0 / ACCN = C3_ACC holds any strip (nrmax =
// ids / 2 roots; a CU's LDS to itself), NRCAP = C3_LIGHT_ROOTS / ACCN = C3_LIGHT_ACC holds strips with few local roots in half the LDS.
template <int AT, int NRCAP, int ACCN>
__global__ __launch_bounds__(AT, 4) void k_ccl3_label(const u64* __restrict__ bits, ccl_geom G, c3_plan P, const u32* __restrict__ ncrowded,
                                                                 const unsigned char* __restrict__ klass,
                                                                 const u32* __restrict__ clist, const u32* __restrict__ parent,
                                                                 const u32* __restrict__ flags, const u32* __restrict__ child,
                                                                 const u32* __restrict__ prefix, const u32* __restrict__ lrootbits,
                                                                 const u32* __restrict__ root16, const u32* __restrict__ barr,
                                                                 c3_state* __restrict__ state, contrib* __restrict__ tot, int tot_stride,
                                                                 int32_t* __restrict__ nlabels,
                                                                 ccl_acc* __restrict__ acc, int max_labels, int32_t* __restrict__ labels,
                                                                 int32_t* __restrict__ stats, double* __restrict__ cent, int dbg)
{
    const u32 nc = *ncrowded;
    if (nc == 0) return;
    const u32 total = nc * (u32)P.strips;
    extern __shared__ __attribute__((aligned(16))) u64 c3_lds[];
    __shared__ u32 red[AT / 64 + 1];
    __shared__ u32 sbase[C3_MAX_STRIPS + 2];                   // roots in the strips before each strip
    __shared__ u32 t_label[C3_TAB];                            // labels of the components this strip only holds a part of ...
    __shared__ contrib t_rec[C3_TAB];                          // ... and what the strip adds to them
    __shared__ contrib wtot[AT / 64];                          // the waves' shares of the strip's totals
    const int NT = AT;
    const int ww = G.ww, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nwmax = P.R * ww;
    const u32 nsl = P.ids / 32, nrmax = NRCAP ? (u32)NRCAP : P.ids / 2;
    u64* lbits = c3_lds;
    u32* lab = reinterpret_cast<u32*>(c3_lds + nwmax);
    unsigned short* lr16 = reinterpret_cast<unsigned short*>(lab + nrmax);
    unsigned short* rid = lr16 + P.ids;
    u32* lrb = reinterpret_cast<u32*>(rid + nrmax);
    u32* lrp = lrb + nsl;
    u32* gfl = lrp + nsl;
    u32* gpf = gfl + nsl;
    u32* gch = gpf + nsl;
    // area | sum of y | sum of x of a local component in ONE 64-bit word (16 | 20 | 28 bits: a strip holds at most 32,768 pixels in at
    // most 32 rows of at most 4,096 columns - c3_make_plan), so that a segment costs one LDS add for the three
    unsigned long long* a_pack = reinterpret_cast<unsigned long long*>(gch + nsl + (nsl & 1u));   // (five bitmaps of nsl words before it: an odd nsl leaves a gap of one word)
    u32* a_minx = reinterpret_cast<u32*>(a_pack + ACCN);
    u32* a_maxx = a_minx + ACCN;
    u32* a_rows = a_maxx + ACCN;
    const u64 lastmask = (G.w & 63) ? ((1ull << (G.w & 63)) - 1ull) : ~0ull;
    const u32 gpr = (u32)((G.w + 3) / 4);
    C3_PROBE_DECL;
    // The block's items, 64 at a time: every wave reads the classes of the same 64 (one round trip instead of one per item) and all of
    // them walk the ones marked for this instantiation.
    for (u32 it0 = blockIdx.x; it0 < total; it0 += 64u * gridDim.x)
    for (unsigned long long todo = __ballot(it0 + (u32)(threadIdx.x & 63) * gridDim.x < total &&
                                            klass[min(it0 + (u32)(threadIdx.x & 63) * gridDim.x, total - 1u)] == (NRCAP ? 1 : 2));
         todo; todo &= todo - 1ull) {
        const u32 item = it0 + (u32)(__ffsll((long long)todo) - 1) * gridDim.x;
        const u32 f = clist[item / (u32)P.strips];
        const int s = (int)(item % (u32)P.strips);
        const int y0 = s * P.R;
        const int nrows = min(P.R, G.h - y0);
        const u64* fb = bits + (size_t)f * G.h * ww;
        const u32* fpar = parent + (size_t)f * G.nids;
        const u32* ffl = flags + (size_t)f * G.nw32;
        const u32* fpf = prefix + (size_t)f * G.nw32;
        const u32* fch = child + (size_t)f * G.nw32;
        const u32* flr = lrootbits + (size_t)f * G.nw32;
        const u32* f16 = root16 + (size_t)f * G.nids;
        const u32* scount = barr + (size_t)f * 3 * (P.strips + 1) + 2 * (P.strips + 1);
        ccl_acc* facc = acc + (size_t)f * max_labels;
        const u32 base = (u32)s * P.ids;
        const u32 lim = min(P.ids, G.nids - base);
        __syncthreads();                                      // the previous item's LDS is done with
        C3_PROBE(9);   // item set up, waited for the block
        // ---- everything the strip needs from global memory, requested together --------------------------------------------------------
        u32 sc[(C3_MAX_STRIPS + AT - 1) / AT];
#pragma unroll
        for (int q = 0; q < (C3_MAX_STRIPS + AT - 1) / AT; q++) sc[q] = (q * NT + tid < P.strips) ? scount[q * NT + tid] : 0u;
        {   // every load of a thread is out before its first result is stored (loops of load-then-store pairs wait for each load in
            // turn: 6 us per strip): its word of the strip, two 16-byte pieces of the u16 roots, its word of the four bitmaps
            const int wr = tid / ww, wj = tid - wr * ww;
            const bool hw = tid < nrows * ww;
            const u64 v_bits = hw ? fb[(size_t)(y0 + wr) * ww + wj] : 0ull;
            const uint4* src = reinterpret_cast<const uint4*>(f16 + base / 2);     // base / 2 and lim / 2 are multiples of 16 words
            const u32 n4 = lim / 8;
            uint4 v16[2];
#pragma unroll
            for (int q = 0; q < 2; q++) { const u32 k = tid + (u32)q * NT; if (k < n4) v16[q] = src[k]; }
            const bool in = (u32)tid < lim / 32;
            const u32 v_fl = in ? ffl[base / 32 + tid] : 0u, v_pf = in ? fpf[base / 32 + tid] : 0u, v_ch = in ? fch[base / 32 + tid] : 0u,
                      v_lr = in ? flr[base / 32 + tid] : 0u;
            if (hw) lbits[tid] = v_bits;
            uint4* dst = reinterpret_cast<uint4*>(lr16);
#pragma unroll
            for (int q = 0; q < 2; q++) { const u32 k = tid + (u32)q * NT; if (k < n4) dst[k] = v16[q]; }
            if ((u32)tid < nsl) { gfl[tid] = v_fl; gpf[tid] = v_pf; gch[tid] = v_ch; lrb[tid] = v_lr; }
            // (not reached for any plan: a strip has at most NT / 2 words, 2 * NT pieces of roots and NT bitmap words)
            for (int i = tid + NT; i < nrows * ww; i += NT) lbits[i] = fb[(size_t)(y0 + i / ww) * ww + i % ww];
            for (u32 k = tid + 2u * NT; k < n4; k += NT) dst[k] = src[k];
            for (u32 k = tid + NT; k < nsl; k += NT) {
                const bool in2 = k < lim / 32;
                gfl[k] = in2 ? ffl[base / 32 + k] : 0u; gpf[k] = in2 ? fpf[base / 32 + k] : 0u;
                gch[k] = in2 ? fch[base / 32 + k] : 0u; lrb[k] = in2 ? flr[base / 32 + k] : 0u;
            }
        }
        for (int k = tid; k < C3_TAB; k += NT) { t_label[k] = 0u; contrib_zero(t_rec[k]); }
        C3_PROBE(8);   // this thread's loads have landed in LDS
        // roots in the strips before each strip: exclusive scan of the per-strip counts (the same in every block of the frame)
        {
            u32 run = 0;
#pragma unroll
            for (int q = 0; q < (C3_MAX_STRIPS + AT - 1) / AT; q++) {
                if (q * NT >= P.strips) break;                // block-uniform
                const u32 c = sc[q];
                u32 inc = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
                __syncthreads();
                if (lane == 63) red[wv] = inc;
                __syncthreads();
                u32 off = 0, tot = 0;
                for (int k = 0; k < NT / 64; k++) { const u32 t = red[k]; if (k < wv) off += t; tot += t; }
                if (q * NT + tid < P.strips) sbase[q * NT + tid] = run + off + inc - c;
                run += tot;
            }
            if (tid == 0) {
                sbase[P.strips] = run;
                if (s == 0) {                                 // the frame's label count: background + every root
                    if (nlabels) nlabels[f] = (int32_t)(run + 1u);
                    state[f].pad = run + 1u;
                }
            }
        }
        __syncthreads();
        C3_PROBE(0);   // strip bases, everything staged
        // ---- local ranks of the local roots (prefix of the local-root bitmap), their ids by rank --------------------------------------
        u32 nroots;
        {
            const u32 c = tid < nsl ? (u32)__popc(lrb[tid]) : 0u;      // a slice is at most 256 words: one per thread
            u32 inc = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
            if (lane == 63) red[wv] = inc;
            __syncthreads();
            u32 off = 0, tot = 0;
            for (int k = 0; k < NT / 64; k++) { const u32 t = red[k]; if (k < wv) off += t; tot += t; }
            nroots = tot;
            if (tid < nsl) {
                u32 rk = off + inc - c;
                lrp[tid] = rk;
                u32 m = lrb[tid];
                while (m) {                                   // ids of the local roots by rank
                    const int b = __ffs((int)m) - 1;
                    m &= m - 1;
                    rid[rk++] = (unsigned short)(((u32)tid << 5) + (u32)b);
                }
            }
        }
        __syncthreads();
        // labels of the local roots, one rank per thread and turn so that the look-ups of absorbed roots (two dependent reads of global
        // memory each) are in flight together instead of queueing in the few threads whose bitmap words hold them
        for (u32 r0 = tid; r0 < nroots; r0 += 4 * NT) {
            u32 idv[4], gv[4];
            bool own[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32 rk = r0 + (u32)q * NT;
                idv[q] = rk < nroots ? (u32)rid[rk] : 0u;
                own[q] = (gfl[idv[q] >> 5] >> (idv[q] & 31)) & 1u;        // still a root of the frame: its rank comes from the strip's own slice
                gv[q] = (rk < nroots && !own[q]) ? fpar[base + idv[q]] : 0u;   // absorbed: the ranking launch left the frame's root in its parent entry
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32 rk = r0 + (u32)q * NT;
                if (rk >= nroots) continue;
                const u32 id = idv[q], g = gv[q];
                u32 label;
                if (own[q]) label = sbase[s] + gpf[id >> 5] + (u32)__popc(gfl[id >> 5] & ((1u << (id & 31)) - 1u)) + 1u;
                else label = sbase[g / P.ids] + fpf[g >> 5] + (u32)__popc(ffl[g >> 5] & ((1u << (g & 31)) - 1u)) + 1u;
                lab[rk] = label;
            }
        }
        __syncthreads();
        C3_PROBE(1);   // local roots ranked and labelled
        // every segment's entry: the id of its root -> the local rank of its root (a thread per half-word, as in k_ccl3_link)
        for (int i2 = tid; i2 < nrows * ww * 2; i2 += NT) {
            const int i = i2 >> 1, half = i2 & 1, r = i / ww, j = i - r * ww;
            const u64 w = lbits[i];
            u32 st = (u32)((w & ~(w << 1)) >> (32 * half));
            u32 ids_[4], rk_[4];                              // (reads first, writes after: a root's own entry is somebody's read)
            while (st) {
                int cnt = 0;
                while (st && cnt < 4) {
                    const int sb = __ffs((int)st) - 1 + 32 * half;
                    st &= st - 1;
                    const u32 id = c3_rel(G, r, 64 * j + sb);
                    const u32 root = lr16[id];
                    ids_[cnt] = id;
                    rk_[cnt] = lrp[root >> 5] + (u32)__popc(lrb[root >> 5] & ((1u << (root & 31)) - 1u));
                    cnt++;
                }
                for (int q = 0; q < cnt; q++) lr16[ids_[q]] = (unsigned short)rk_[q];
            }
        }
        __syncthreads();
        C3_PROBE(2);   // ranks in place of root ids
        // ---- statistics per local component, ACCN components per pass ---------------------------------------------------------------
        contrib tot_c;
        contrib_zero(tot_c);
        for (u32 c0 = 0; c0 < nroots && !(dbg & 8); c0 += ACCN) {
            for (u32 k = tid; k < ACCN; k += NT) { a_pack[k] = 0ull; a_minx[k] = 0xffffffffu; a_maxx[k] = 0; a_rows[k] = 0; }
            __syncthreads();
            // Every segment adds to its component's accumulators.  A wave's lanes mostly name the same component when one is large (half
            // the pixels of 50 % noise belong to one): 64 LDS atomics on one word take 64 turns, so the lanes that agree with the first
            // active lane are combined with shuffles first and added once.
            for (int i0 = 0; i0 < nrows * ww * 2; i0 += NT) {
                const int i2 = i0 + tid;
                const bool valid = i2 < nrows * ww * 2;
                const int i = i2 >> 1, half = i2 & 1;
                const int r = valid ? i / ww : 0, j = valid ? i - r * ww : 0;
                const u64 w = valid ? lbits[i] : 0ull;
                u32 rem = (u32)((w & ~(w << 1)) >> (32 * half));      // the segments that START in this thread's half of the word
                while (__any(rem != 0u)) {
                    bool act = rem != 0u;
                    u32 k = 0xffffffffu, len = 0, sxv = 0, syv = 0, xs = 0xffffffffu, xe = 0, rowbit = 0;
                    if (act) {
                        const int sb = __ffs((int)rem) - 1 + 32 * half;
                        rem &= rem - 1;
                        const int eb = c3_run_end(w, sb);
                        k = (u32)lr16[c3_rel(G, r, 64 * j + sb)] - c0;
                        act = k < (u32)ACCN;
                        len = (u32)(eb - sb + 1);
                        xs = (u32)(64 * j + sb); xe = (u32)(64 * j + eb);
                        sxv = len * (xs + xe) / 2u; syv = len * (u32)r; rowbit = 1u << r;
                    }
                    const unsigned long long am = __ballot(act);
                    if (!am) continue;
                    const int lead = __ffsll((long long)am) - 1;
                    const u32 kd = __shfl(k, lead);
                    const bool same = act && k == kd;
                    if (__popcll(__ballot(same)) >= 8) {
                        const u32 t_len = c3_wave_add(same ? len : 0u), t_sx = c3_wave_add(same ? sxv : 0u), t_sy = c3_wave_add(same ? syv : 0u),
                                  t_xs = c3_wave_min(same ? xs : 0xffffffffu), t_xe = c3_wave_max(same ? xe : 0u), t_rb = c3_wave_or(same ? rowbit : 0u);
                        if (lane == lead) {
                            atomicAdd(a_pack + kd, ((unsigned long long)t_len << 48) | ((unsigned long long)t_sy << 28) | (unsigned long long)t_sx);
                            atomicMin(a_minx + kd, t_xs); atomicMax(a_maxx + kd, t_xe); atomicOr(a_rows + kd, t_rb);
                        }
                        act = act && !same;
                    }
                    if (act) {
                        atomicAdd(a_pack + k, ((unsigned long long)len << 48) | ((unsigned long long)syv << 28) | (unsigned long long)sxv);
                        atomicMin(a_minx + k, xs); atomicMax(a_maxx + k, xe); atomicOr(a_rows + k, rowbit);
                    }
                }
            }
            __syncthreads();
            C3_PROBE(3);   // accumulate
            // One lane per local root of this pass, in the order of the ranks = of the labels: neighbouring lanes write neighbouring rows.
            // Complete within the strip (still a root of the frame, no members elsewhere): its row goes straight out.  Otherwise its sums
            // join its label's entry of the strip's table, lanes of a wave that carry the same label combined with shuffles first (at
            // 50 % noise a strip holds about a thousand fragments of the one big component); the table goes to the frame's accumulators
            // once per strip, not once per fragment.
            for (u32 kb = 0; kb < (u32)ACCN; kb += NT) {
                const u32 k = kb + (u32)tid;
                bool commit = false;
                u32 label = 0;
                contrib c;
                contrib_zero(c);
                if (k < (u32)ACCN && c0 + k < nroots) {
                    const u32 id = rid[c0 + k];
                    const u32 bit = 1u << (id & 31);
                    const unsigned long long pk = a_pack[k];
                    c.area = (u32)(pk >> 48);
                    c.sx = pk & 0xfffffffull;
                    c.sy = ((pk >> 28) & 0xfffffull) + (u64)c.area * (u64)y0;
                    c.minx = (int)a_minx[k]; c.maxx = (int)a_maxx[k];
                    c.miny = y0 + (__ffs((int)a_rows[k]) - 1); c.maxy = y0 + (31 - __clz((int)a_rows[k]));
                    tot_c.area += c.area; tot_c.sx += c.sx; tot_c.sy += c.sy;
                    label = lab[c0 + k];
                    if (label < (u32)max_labels) {
                        if ((gfl[id >> 5] & bit) && !(gch[id >> 5] & bit)) {
                            const size_t o = (size_t)f * max_labels + label;
                            if (stats) {
                                int32_t* sp = stats + o * 5;
                                sp[0] = c.minx; sp[1] = c.miny; sp[2] = c.maxx - c.minx + 1; sp[3] = c.maxy - c.miny + 1; sp[4] = (int32_t)c.area;
                            }
                            if (cent) {
                                const double area = (double)c.area;
                                cent[o * 2] = (double)c.sx / area;
                                cent[o * 2 + 1] = (double)c.sy / area;
                            }
                        } else {
                            commit = true;
                        }
                    }
                }
                unsigned long long cm = __ballot(commit);
                while (cm) {                                                        // wave-uniform: one turn per distinct label among the lanes
                    const int lead = __ffsll((long long)cm) - 1;
                    const u32 ld = __shfl(label, lead);
                    const bool same = commit && label == ld;
                    const unsigned long long sm = __ballot(same);
                    if (__popcll(sm) >= 4) {
                        contrib g = c;
                        if (!same) contrib_zero(g);
                        c3_wave_combine(g);
                        if (lane == lead) c3_table_add(t_label, t_rec, ld, g, facc);
                    } else if (same) {
                        c3_table_add(t_label, t_rec, label, c, facc);
                    }
                    commit = commit && !same;
                    cm &= ~sm;
                }
            }
            __syncthreads();
            C3_PROBE(4);   // emit rows / table
        }
        // the table of this strip -> the frame's accumulators
        for (int k = tid; k < C3_TAB; k += NT)
            if (t_label[k]) acc_commit(facc + t_label[k], t_rec[k]);
        C3_PROBE(5);   // table flushed
        // ---- the strip's part of the label image: one lane = 4 px = one 16-byte store, labels from LDS --------------------------------
        if (labels && !(dbg & 16)) {
            int32_t* lrow0 = labels + ((size_t)f * G.h + y0) * G.w;
            const bool vec = (G.w & 3) == 0 && ((((uintptr_t)lrow0) & 15) == 0);
            const u32 ngroups = (u32)nrows * gpr;
            for (u32 q = tid; q < ngroups; q += NT) {
                const u32 r = q / gpr, g = q - r * gpr;
                const int x0 = (int)g * 4, j = x0 >> 6, sub = x0 & 63;
                const u64 w = lbits[r * ww + j];
                const u32 nib = (u32)(w >> sub) & 0xfu;
                int vv[4] = {0, 0, 0, 0};
                if (nib) {
                    const int tz = __ffs((int)nib) - 1;
                    const u32 t = nib >> tz;
                    const int runlen = __ffs((int)~t) - 1;
                    const u32 m1 = ((1u << runlen) - 1u) << tz, m2 = nib & ~m1;
                    const u32 la = lab[lr16[c3_rel(G, (int)r, 64 * j + run_start(w, sub + tz))]];
                    const u32 lb = m2 ? lab[lr16[c3_rel(G, (int)r, x0 + (__ffs((int)m2) - 1))]] : 0u;
#pragma unroll
                    for (int b = 0; b < 4; b++) vv[b] = ((m1 >> b) & 1u) ? (int)la : (((m2 >> b) & 1u) ? (int)lb : 0);
                }
                int32_t* d = lrow0 + (size_t)r * G.w + x0;
                if (vec && x0 + 4 <= G.w) {
                    vp_store16(d, (u32)vv[0], (u32)vv[1], (u32)vv[2], (u32)vv[3]);
                } else {
                    for (int b = 0; b < 4; b++)
                        if (x0 + b < G.w) d[b] = vv[b];
                }
            }
        }
        C3_PROBE(6);   // label stores issued
        // ---- totals of the strip for the frame's background row: foreground sums, bounding box of the zero pixels
        C3_FOR_WORDS(r, j, i, NT) {
            const u64 z = ~lbits[i] & (j == ww - 1 ? lastmask : ~0ull);
            if (!z) continue;
            tot_c.minx = min(tot_c.minx, 64 * j + (__ffsll((long long)z) - 1));
            tot_c.maxx = max(tot_c.maxx, 64 * j + 63 - __clzll(z));
            tot_c.miny = min(tot_c.miny, y0 + r);
            tot_c.maxy = max(tot_c.maxy, y0 + r);
        }
        // (a record per strip, summed by k_ccl3_rows: atomics on the frame's one set of totals queue up at the memory side - every wave
        // of every strip of the frame - and, retiring in order with everything else a wave sends there, they held back the next
        // strip's loads)
        c3_wave_combine(tot_c);
        if (lane == 0) wtot[wv] = tot_c;
        __syncthreads();
        if (wv == 0) {
            contrib c;
            contrib_zero(c);
            if (lane < AT / 64) c = wtot[lane];
            c3_wave_combine(c);
            if (lane == 0) tot[(size_t)f * tot_stride + s] = c;
        }
        C3_PROBE(7);   // totals
    }
    C3_PROBE_END(2);
}

// ---- K3: rows of the components that span strips, the background row, zeros past the last label -----------------------------------
// one item per (handed-over frame, strip): the roots of the strip's slice that gathered members elsewhere have their sums in the
// accumulators by now (the labelling launch is complete)
__global__ __launch_bounds__(256) void k_ccl3_rows(ccl_geom G, c3_plan P, const u32* __restrict__ ncrowded, const u32* __restrict__ clist,
                                                   const u32* __restrict__ flags, const u32* __restrict__ child, const u32* __restrict__ prefix,
                                                   const u32* __restrict__ barr, const c3_state* __restrict__ state, const contrib* __restrict__ tot,
                                                   int tot_stride, ccl_acc* __restrict__ acc, int max_labels, int32_t* __restrict__ stats,
                                                   double* __restrict__ cent, int acc_clean)
{
    const u32 nc = *ncrowded;
    if (nc == 0 || (!stats && !cent && !acc_clean)) return;
    __shared__ u32 sbase[C3_MAX_STRIPS + 2];
    __shared__ u32 red[256 / 64 + 1];
    const int NT = 256, tid = threadIdx.x;
    const u32 total = nc * (u32)P.strips;
    for (u32 item = blockIdx.x; item < total; item += gridDim.x) {
        const u32 f = clist[item / (u32)P.strips];
        const int s = (int)(item % (u32)P.strips);
        const u32* ffl = flags + (size_t)f * G.nw32;
        const u32* fch = child + (size_t)f * G.nw32;
        const u32* fpf = prefix + (size_t)f * G.nw32;
        ccl_acc* facc = acc + (size_t)f * max_labels;
        __syncthreads();
        c3_strip_bases<256>(P, barr + (size_t)f * 3 * (P.strips + 1) + 2 * (P.strips + 1), sbase, red);
        const u32 k0 = (u32)s * (P.ids / 32), k1 = (s == P.strips - 1) ? G.nw32 : min(k0 + P.ids / 32, G.nw32);
        for (u32 k = k0 + tid; k < k1; k += NT) {
            const u32 fl = ffl[k];
            u32 m = fl & fch[k];
            while (m) {
                const int b = __ffs((int)m) - 1;
                m &= m - 1;
                const u32 label = sbase[s] + fpf[k] + (u32)__popc(fl & ((1u << b) - 1u)) + 1u;
                if (label >= (u32)max_labels) continue;
                const ccl_acc a = facc[label];
                if (acc_clean) {      // hand the entry back as the next call expects it (only entries of components that span strips were ever touched)
                    ccl_acc z;
                    z.area = 0; z.minx = INT_MAX; z.miny = INT_MAX; z.maxx = INT_MIN; z.maxy = INT_MIN; z.pad = 0; z.sx = 0; z.sy = 0;
                    facc[label] = z;
                }
                const size_t o = (size_t)f * max_labels + label;
                if (stats) {
                    int32_t* sp = stats + o * 5;
                    sp[0] = a.minx; sp[1] = a.miny; sp[2] = a.maxx - a.minx + 1; sp[3] = a.maxy - a.miny + 1; sp[4] = (int32_t)a.area;
                }
                if (cent) {
                    const double area = (double)a.area;
                    cent[o * 2] = (double)a.sx / area;
                    cent[o * 2 + 1] = (double)a.sy / area;
                }
            }
        }
        // this strip's share of the rows past the last label
        const c3_state st = state[f];
        const int nl = max((int)st.pad, 1);
        if (nl < max_labels) {
            const long long span = (long long)max_labels - nl, per = (span + P.strips - 1) / P.strips;
            const long long l0 = nl + (long long)s * per, l1 = min(l0 + per, (long long)max_labels);
            for (long long l = l0 + tid; l < l1; l += NT) {
                const size_t o = (size_t)f * max_labels + (size_t)l;
                if (stats) { int32_t* sp = stats + o * 5; sp[0] = sp[1] = sp[2] = sp[3] = sp[4] = 0; }
                if (cent) { cent[o * 2] = 0.0; cent[o * 2 + 1] = 0.0; }
            }
        }
        if (s == 0) {                                         // the background row: the frame's totals minus the foreground's (block-uniform)
            __shared__ contrib wsum[256 / 64];
            contrib c;
            contrib_zero(c);
            for (int k = tid; k < P.strips; k += NT) contrib_merge(c, tot[(size_t)f * tot_stride + k]);
            wave_combine(c);
            __syncthreads();
            if ((tid & 63) == 0) wsum[tid >> 6] = c;
            __syncthreads();
            if (tid == 0) {
                for (int k = 1; k < 256 / 64; k++) contrib_merge(c, wsum[k]);
                const u64 W = (u64)G.w, H = (u64)G.h;
                const u32 area = (u32)(W * H) - c.area;
                const u64 sx = H * (W * (W - 1ull) / 2ull) - c.sx;
                const u64 sy = W * (H * (H - 1ull) / 2ull) - c.sy;
                const size_t o = (size_t)f * max_labels;
                if (stats) {
                    int32_t* sp = stats + o * 5;
                    sp[0] = c.minx; sp[1] = c.miny;
                    sp[2] = (int32_t)((u32)c.maxx - (u32)c.minx + 1u);
                    sp[3] = (int32_t)((u32)c.maxy - (u32)c.miny + 1u);
                    sp[4] = (int32_t)area;
                }
                if (cent) {
                    cent[o * 2] = (double)sx / (double)area;
                    cent[o * 2 + 1] = (double)sy / (double)area;
                }
            }
        }
    }
}

static size_t c3_link_lds(const ccl_geom& G, const c3_plan& P) { return (size_t)P.R * G.ww * 8 + (size_t)P.ids * 4 + (size_t)P.ids / 32 * 4; }
static size_t c3_label_lds(const ccl_geom& G, const c3_plan& P, size_t nrcap, size_t accn)
{
    const size_t nrmax = nrcap ? nrcap : (size_t)P.ids / 2;
    return (size_t)P.R * G.ww * 8 + nrmax * 4 + (size_t)P.ids * 2 + nrmax * 2 + (size_t)P.ids / 32 * 4 * 5 + 8 + accn * 4 * 5;
}
