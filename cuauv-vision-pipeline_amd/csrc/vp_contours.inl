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
// bitmap of start pixels + popcount prefix for the order, and segment-parallel border following (below).
//
// RETR_EXTERNAL: OpenCV decides "inside a hole" from the sign of the last border mark left of a start pixel.  A mark is negative
// exactly when the tracer examined the pixel's east neighbour as background, i.e. when the east crack belongs to the traced
// border - and every crack belongs to exactly one border (see below) - so the mark rule and the topological rule used here
// select the same borders (no difference on 460 noise / thin-wall masks: tests/test_gpu_contours.py, tools/exp_external_rule.py).

struct ct_frame_out {      // per frame, device
    int32_t n_contours;
    int32_t n_points;
};

// outside[] bit per background root: the region touches the image frame
__global__ __launch_bounds__(256) void k_ct_outside(const u64* __restrict__ bits, ccl_geom G, const u32* __restrict__ parent, u32* __restrict__ outside,
                                                    const u32* __restrict__ only)
{
    // one thread per frame-border word: rows 0 and h-1 fully, columns 0 and ww-1 of the other rows
    const int f = blockIdx.y;
    if (only && !only[f]) return;                            // this frame's background was not resolved: nobody will ask
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
        // nearly every frame-touching segment belongs to the one big outside region: test before setting, or a couple of thousand
        // atomics per frame queue up on a single word (28 us of a single-frame call were exactly that)
        if (!((ld_rlx(o + (r >> 5)) >> (r & 31)) & 1u)) atomicOr(o + (r >> 5), 1u << (r & 31));
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

// seeds: bitmaps in the layout of a bit image.  startmap = start pixel of every border of the image (every cycle of the
// border-following map gets exactly one), holemap = it is a hole border, selmap = the retrieval mode returns it.
// grid (ceil(nw32/256), n): thread = one 32-bit word of the root bitmaps.
__global__ __launch_bounds__(256) void k_ct_seeds(const u64* __restrict__ bits, ccl_geom Gf, ccl_geom Gb, const u32* __restrict__ fg_flags,
                                                  const u32* __restrict__ bg_flags, const u32* __restrict__ bg_parent,
                                                  const u32* __restrict__ outside, int mode, u64* __restrict__ startmap, u64* __restrict__ holemap,
                                                  u64* __restrict__ selmap, u32* __restrict__ partsum2, int nparts, const u32* __restrict__ bg_only)
{
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    if (t >= Gf.nw32) return;
    const int f = blockIdx.y;
    // RETR_EXTERNAL of a frame in which no component's box lies strictly inside another's (k_ct_needs_bg): every outer border is
    // external and no hole border is asked for, so the background arrays of this frame were never made and are not looked at
    const bool nobg = bg_only && !bg_only[f];
    const size_t fo = (size_t)f * Gf.h * Gf.ww;
    const u64* fb = bits + fo;
    u64* sm = startmap + fo;
    u64* hm = holemap + fo;
    u64* sel = selmap + fo;
    u32* ps2 = partsum2 + (size_t)f * nparts;      // selected starts per 256 words of the frame (k_ct_prefix scans them)
    auto select = [&](size_t wi, int bit) {
        const unsigned long long old = atomicOr((unsigned long long*)&sel[wi], 1ull << bit);
        if (!((old >> bit) & 1ull)) atomicAdd(ps2 + (wi >> 8), 1u);
    };
    const u32* bp = bg_parent + (size_t)f * Gb.nids;
    const u32* out = outside + (size_t)f * Gb.nw32;
    u32 m = fg_flags[(size_t)f * Gf.nw32 + t];
    while (m) {
        const int b = __ffs((int)m) - 1;
        m &= m - 1;
        int y, x;
        ct_root_pixel(Gf, fb, t * 32 + b, y, x);
        bool keep = true;
        if (mode == 0 && x > 0 && !nobg) {   // RETR_EXTERNAL: the region left of the first pixel must reach the frame
            const int xl = x - 1, j = xl >> 6;
            const u64 wb = ccl_word(Gb, fb, y * Gb.ww + j, j);
            u32 r = seg_id(Gb, y, 64 * j + run_start(wb, xl & 63));
            for (u32 q = bp[r]; q != r; q = bp[r]) r = q;
            keep = (out[r >> 5] >> (r & 31)) & 1u;
        }
        const size_t wi = (size_t)y * Gf.ww + (x >> 6);
        atomicOr((unsigned long long*)&sm[wi], 1ull << (x & 63));
        if (keep) select(wi, x & 63);
    }
    u32 hb = nobg ? 0u : (bg_flags[(size_t)f * Gb.nw32 + t] & ~out[t]);   // hole borders: background regions that do not reach the frame
    while (hb) {
        const int b = __ffs((int)hb) - 1;
        hb &= hb - 1;
        int y, x;
        ct_root_pixel(Gb, fb, t * 32 + b, y, x);
        const int xs = x - 1;   // a hole never touches column 0
        const size_t wi = (size_t)y * Gf.ww + (xs >> 6);
        atomicOr((unsigned long long*)&sm[wi], 1ull << (xs & 63));
        atomicOr((unsigned long long*)&hm[wi], 1ull << (xs & 63));
        if (mode == 1) select(wi, xs & 63);
    }
}

// ---- segment-parallel border following --------------------------------------------------------------------------------
// A border is a cycle of states (pixel, s) of the Suzuki-Abe follower, s = direction of the pixel it came from; the next state
// depends on the 3x3 neighbourhood only: s' = first foreground neighbour counter-clockwise after s, move there, s = s' + 4.
// Between s and s' the follower sweeps over background neighbours.  Every crack (edge between a foreground pixel and a
// 4-adjacent background pixel) is swept by exactly one state of exactly one border, and that state can be written down from
// the crack alone: s = first foreground neighbour clockwise from the crack's direction.  So the states that sweep a W or E
// crack (and, to cut long flat edges, an N or S crack at x % 8 == 0) are enumerable with bit operations - the "heads" - and
// they cut every border into short segments that are followed independently, one thread each:
//   k_ct_headmaps   4 head bitmaps per word (a state that sweeps several eligible cracks belongs to the first one swept); clears the
//                   seed bitmaps k_ct_seeds fills next
//   k_ct_prefix     popcount prefix -> dense head index + head list (pixel, type), terminal marks cleared; the same kernel
//                   ranks the selected start pixels (cv2's contour order)
//   k_ct_starts     start pixel of every border -> its start state -> the head that owns it = the terminal of that cycle
//   k_ct_seg<false> follow each segment to the next head: node[k] = (next head, points emitted)
//   k_ct_jump       pointer jumping on (next, distance) pairs until every head points at its terminal; then per contour:
//                   length = distance of the terminal around its cycle, exclusive scan -> offsets
//   k_ct_seg<true>  follow each segment again, writing its points at offset[contour] + (length - distance to terminal)
// OpenCV's start rule (icvFetchContour: first neighbour clockwise from W for an outer border, from E for a hole border) is
// the head rule for the W / E crack of the start pixel, and its CHAIN_APPROX_SIMPLE filter (keep a point when the direction
// changes) is local to a state: s' != s ^ 4.
#define CT_TERM 0x80000000u
#define CT_NONE 0xffffffffu
#define CT_UNSEL 0xfffffffeu
#define CT_EL_NS 0x0101010101010101ull   // N / S cracks are heads only in columns x % 8 == 0
#define CT_JUMP_ROUNDS 40

struct ct_aux { u32 nheads; u32 nsel; };

__host__ __device__ inline size_t ct_hcap(int w, int h) { const size_t npx = (size_t)w * h; return (npx + npx / 4 + 64 + 63) / 64 * 64; }

// neighbour bitmaps of word (y, j): bit b of f[d] = neighbour of pixel (y, 64j + b) in direction d of {E, NE, N, NW, W, SW, S, SE}
__device__ __forceinline__ void ct_neighbours(const ccl_geom& G, const u64* __restrict__ fb, int y, int j, u64& c, u64 f[8])
{
    const bool up = y > 0, dn = y + 1 < G.h, lf = j > 0, rt = j + 1 < G.ww;
    const u64* row = fb + (size_t)y * G.ww + j;
    const u64 cm = row[0];
    const u64 cl = lf ? row[-1] : 0ull, cr = rt ? row[1] : 0ull;
    const u64 um = up ? row[-G.ww] : 0ull, ul = (up && lf) ? row[-G.ww - 1] : 0ull, ur = (up && rt) ? row[-G.ww + 1] : 0ull;
    const u64 dm = dn ? row[G.ww] : 0ull, dl = (dn && lf) ? row[G.ww - 1] : 0ull, dr = (dn && rt) ? row[G.ww + 1] : 0ull;
    c = cm;
    f[0] = (cm >> 1) | (cr << 63);
    f[4] = (cm << 1) | (cl >> 63);
    f[2] = um; f[1] = (um >> 1) | (ur << 63); f[3] = (um << 1) | (ul >> 63);
    f[6] = dm; f[7] = (dm >> 1) | (dr << 63); f[5] = (dm << 1) | (dl >> 63);
}

// head bitmaps, 4 per word: [W, E, N, S].  A crack's state owns the head unless, going clockwise from the crack towards the
// state's s, another eligible crack comes first (that one is swept earlier).
// block sum of per-thread counts -> partsum[f][blockIdx.x] (grid.x blocks of 256 words per frame)
__device__ __forceinline__ void ct_block_sum(u32 c, u32* __restrict__ partsum)
{
    __shared__ u32 ws4[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if (lane == 0) ws4[wv] = c;
    __syncthreads();
    if (threadIdx.x == 0) partsum[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = ws4[0] + ws4[1] + ws4[2] + ws4[3];
}

// (also clears this word of the three seed bitmaps and the block's entry of the selected-start partial sums, which k_ct_seeds fills
// afterwards: one launch instead of a memset, this one and a counting pass)
__global__ __launch_bounds__(256) void k_ct_headmaps(const u64* __restrict__ bits, ccl_geom G, u64* __restrict__ hmaps, u32* __restrict__ partsum,
                                                     u64* __restrict__ maps3, size_t mstride, u32* __restrict__ partsum2, uint8_t* __restrict__ cnt8)
{
    const int nwords = G.h * G.ww;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    u32 cnt = 0;
    if (threadIdx.x == 0) partsum2[(size_t)f * gridDim.x + blockIdx.x] = 0u;
    if (idx < nwords) {
        const size_t wi = (size_t)f * nwords + idx;
        maps3[wi] = 0ull; maps3[mstride + wi] = 0ull; maps3[2 * mstride + wi] = 0ull;
        const u64* fb = bits + (size_t)f * nwords;
        const int y = idx / G.ww, j = idx - y * G.ww;
        u64 c, n[8];
        ct_neighbours(G, fb, y, j, c, n);
        const u64 el = CT_EL_NS;
        u64 hw = 0, he = 0, hn = 0, hs = 0;
        if (c) {
            hw = c & ~n[4] & (n[3] | n[2] | (~el & (n[1] | n[0])));
            he = c & ~n[0] & (n[7] | n[6] | (~el & (n[5] | n[4])));
            hn = c & ~n[2] & el & (n[1] | n[0]);
            hs = c & ~n[6] & el & (n[5] | n[4]);
        }
        cnt = (u32)(__popcll(hw) + __popcll(he) + __popcll(hn) + __popcll(hs));
        // The bitmaps are only ever read back for words that hold a head (the head list, ct_head_index); every word's COUNT is what the
        // prefix needs.  A mask is mostly words without heads: 1 byte per word instead of 32 (132 -> 4 MB per 128 frames of 1080p).
        cnt8[(size_t)f * nwords + idx] = (uint8_t)cnt;       // (at most 4 x 64 heads, and a word with 256 would need every pixel to be two heads)
        if (cnt) {
            ulonglong2* o = reinterpret_cast<ulonglong2*>(hmaps + ((size_t)f * nwords + idx) * 4);
            o[0] = make_ulonglong2(hw, he);
            o[1] = make_ulonglong2(hn, hs);
        }
    }
    ct_block_sum(cnt, partsum);
}

// exclusive prefix of the popcounts of `nm` bitmaps per word -> base[word]; total -> total_out[f * tstride].
// grid (ceil(nwords/256), n): block b adds up the block sums before it, then scans its 256 words.
__global__ __launch_bounds__(256) void k_ct_prefix(const u64* __restrict__ maps, int nm, int nwords, const u32* __restrict__ partsum,
                                                   u32* __restrict__ base, u32* __restrict__ total_out, int tstride, int w, int ww,
                                                   u32* __restrict__ head_pix, u32* __restrict__ hrank, size_t hcap, const uint8_t* __restrict__ cnt8)
{
    __shared__ u32 wsum[4], wtot[4];
    const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32* ps = partsum + (size_t)f * gridDim.x;
    u32 c = 0;
    for (int q = tid; q < (int)blockIdx.x; q += 256) c += ps[q];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if (lane == 0) wtot[wv] = c;
    const int i = blockIdx.x * 256 + tid;
    u32 cnt = 0;
    if (i < nwords) {
        if (cnt8) cnt = cnt8[(size_t)f * nwords + i];         // the counts k_ct_headmaps left (the bitmaps of words without heads were never stored)
        else for (int t = 0; t < nm; t++) cnt += (u32)__popcll(maps[((size_t)f * nwords + i) * nm + t]);
    }
    u32 inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    const u32 carry = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    u32 woff = 0;
    for (int k = 0; k < wv; k++) woff += wsum[k];
    if (i < nwords) base[(size_t)f * nwords + i] = carry + woff + inc - cnt;
    if (head_pix && cnt) {   // head list: head_pix[k] = pixel index | type << 29; hrank[k] = CT_NONE (not a terminal)
        const int y = i / ww, j = i - y * ww;
        const u32 pix0 = (u32)(y * w + 64 * j);
        u32 k = carry + woff + inc - cnt;
        u32* hp = head_pix + (size_t)f * hcap;
        u32* hr = hrank + (size_t)f * hcap;
        for (int t = 0; t < 4; t++) {
            u64 m = maps[((size_t)f * nwords + i) * 4 + t];
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                hp[k] = (pix0 + (u32)b) | ((u32)t << 29);
                hr[k] = CT_NONE;
                k++;
            }
        }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) total_out[(size_t)f * tstride] = carry + wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// dense index of head (word idx, bit b, type t)
__device__ __forceinline__ u32 ct_head_index(const u64* __restrict__ hm, const u32* __restrict__ hbase, int idx, int b, int t)
{
    const ulonglong2 m01 = reinterpret_cast<const ulonglong2*>(hm + (size_t)idx * 4)[0];
    const ulonglong2 m23 = reinterpret_cast<const ulonglong2*>(hm + (size_t)idx * 4)[1];
    const u64 low = (1ull << b) - 1ull;
    u32 k = hbase[idx];
    const u64 mt = t == 0 ? m01.x : (t == 1 ? m01.y : (t == 2 ? m23.x : m23.y));
    if (t > 0) k += (u32)__popcll(m01.x);
    if (t > 1) k += (u32)__popcll(m01.y);
    if (t > 2) k += (u32)__popcll(m23.x);
    return k + (u32)__popcll(mt & low);
}

// ---- the follower ----
// 8x8-pixel tile of the mask in one register pair: bit (8*r + c) = pixel (ty + r, tx + c); pixels outside the image are 0.
struct ct_tile { u64 bits; int ty, tx; };

__device__ __forceinline__ void ct_tile_load(const ccl_geom& G, const u64* __restrict__ fb, int ty, int tx, ct_tile& T)
{
    T.ty = ty; T.tx = tx;
    const int j0 = tx >> 6;            // arithmetic shift: tx may be negative (floor)
    const int sh = tx & 63;
    // branch-free: clamped addresses, all 16 loads in flight together, out-of-image parts masked afterwards
    const int ja = min(max(j0, 0), G.ww - 1), jb = min(max(j0 + 1, 0), G.ww - 1);
    const u64 ma = (j0 >= 0 && j0 < G.ww) ? ~0ull : 0ull;
    const u64 mb = (sh > 56 && j0 + 1 >= 0 && j0 + 1 < G.ww) ? ~0ull : 0ull;
    u64 lo[8], hi[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int yc = min(max(ty + r, 0), G.h - 1);
        const u64* row = fb + (size_t)yc * G.ww;
        lo[r] = row[ja];
        hi[r] = row[jb];
    }
    u64 acc = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int y = ty + r;
        const u64 my = (y >= 0 && y < G.h) ? ~0ull : 0ull;
        u64 v = (lo[r] & ma) >> sh;
        v |= sh ? ((hi[r] & mb) << (64 - sh)) : 0ull;
        acc |= (v & my & 0xffull) << (8 * r);
    }
    T.bits = acc;
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
// 8-neighbourhood deltas packed 2 bits each (value + 1): a table indexed at run time would live in memory
#define dx8(s) ((int)((0x901Au >> (2 * (s))) & 3u) - 1)
#define dy8(s) ((int)((0xA901u >> (2 * (s))) & 3u) - 1)

// first foreground neighbour clockwise from direction d (ring must be non-zero)
__device__ __forceinline__ int ct_first_cw(u32 R, int d)
{
    const u32 rr = ((R | (R << 8)) >> d) & 0xfeu;   // bit i = direction d + i, i = 1..7
    return (d + (31 - __clz((int)rr))) & 7;
}
// state (s, sweep length t = number of background neighbours swept before s') at column x: type of the head that owns it
// (0 W, 1 E, 2 N, 3 S) or -1
__device__ __forceinline__ int ct_head_type(int s, int t, int x)
{
    const int iw = (3 - s) & 7, ie = (7 - s) & 7, in = (1 - s) & 7, is = (5 - s) & 7;
    const bool ns = (x & 7) == 0;
    int best = 8, type = -1;
    if (iw < t) { best = iw; type = 0; }
    if (ie < t && ie < best) { best = ie; type = 1; }
    if (ns && in < t && in < best) { best = in; type = 2; }
    if (ns && is < t && is < best) { best = is; type = 3; }
    return type;
}

// start pixel of every border -> terminal head.  starts[r] = pixel | hole << 31 and shead[r] = head index (CT_NONE: single
// pixel) for the selected borders in raster order; hrank[head] = r, or CT_UNSEL for a border the mode does not return.
// (the rank of a word's first selected start = number of selected starts before it in raster order, from the per-block counts
// k_ct_seeds made plus a scan of this block's words: cv2's contour order; the last block publishes the total)
__global__ __launch_bounds__(256) void k_ct_starts(const u64* __restrict__ bits, ccl_geom G, const u64* __restrict__ startmap,
                                                   const u64* __restrict__ holemap, const u64* __restrict__ selmap, const u32* __restrict__ partsum2,
                                                   u32* __restrict__ nsel_out, const u64* __restrict__ hmaps, const u32* __restrict__ hbase,
                                                   u32* __restrict__ hrank, size_t hcap, u32* __restrict__ starts, u32* __restrict__ shead,
                                                   int max_contours)
{
    __shared__ u32 wsum[4], wtot[4];
    const int nwords = G.h * G.ww;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int idx = blockIdx.x * 256 + tid;
    const int f = blockIdx.y;
    const size_t fo = (size_t)f * nwords;
    const bool valid = idx < nwords;
    const u32* ps = partsum2 + (size_t)f * gridDim.x;
    u32 c = 0;
    for (int q = tid; q < (int)blockIdx.x; q += 256) c += ps[q];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if (lane == 0) wtot[wv] = c;
    const u64 sel = valid ? selmap[fo + idx] : 0ull;
    const u32 cnt = (u32)__popcll(sel);
    u32 inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    const u32 carry = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    u32 woff = 0;
    for (int k = 0; k < wv; k++) woff += wsum[k];
    if (blockIdx.x == gridDim.x - 1 && tid == 0) nsel_out[(size_t)f * 2] = carry + wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (!valid) return;
    u64 sm = startmap[fo + idx];
    if (!sm) return;
    const u64 hm = holemap[fo + idx];
    const u64* fb = bits + fo;
    const int y = idx / G.ww, j = idx - y * G.ww;
    u32 r = carry + woff + inc - cnt;
    while (sm) {
        const int b = __ffsll((long long)sm) - 1;
        sm &= sm - 1;
        const int x = 64 * j + b;
        const bool hole = (hm >> b) & 1ull, selected = (sel >> b) & 1ull;
        ct_tile T;
        ct_tile_load(G, fb, y - 3, x - 3, T);
        const u32 R = ct_ring(T, y, x);
        u32 head = CT_NONE;
        if (R) {
            const int s = ct_first_cw(R, hole ? 0 : 4);
            const u32 q = (R | (R << 8)) >> (s + 1);
            const int t = __ffs((int)q) - 1;
            const int type = ct_head_type(s, t, x);   // >= 0: the sweep crosses the W (outer) / E (hole) crack
            if (type >= 0) {
                head = ct_head_index(hmaps + fo * 4, hbase + fo, idx, b, type);
                hrank[(size_t)f * hcap + head] = selected ? r : CT_UNSEL;
            }
        }
        if (selected) {
            if ((int)r < max_contours) {
                starts[(size_t)f * max_contours + r] = (u32)(y * G.w + x) | (hole ? 0x80000000u : 0u);
                shead[(size_t)f * max_contours + r] = head;
            }
            r++;
        }
    }
}

// one thread per head: follow the border from the head's state to the next head.
//   !WRITE: node[k] = (next head | CT_TERM if that is a terminal) << 32 | points emitted
//    WRITE: the points go to their final place (see ct_offsets_body)
template <bool WRITE>
__global__ __launch_bounds__(256) void k_ct_seg(const u64* __restrict__ bits, ccl_geom G, const u64* __restrict__ hmaps, const u32* __restrict__ hbase,
                                                const u32* __restrict__ head_pix, const u32* __restrict__ hrank, size_t hcap,
                                                const ct_aux* __restrict__ aux, unsigned long long* __restrict__ node, int method,
                                                const int32_t* __restrict__ offsets, int32_t* __restrict__ points, int max_contours,
                                                long long max_points)
{
    const int f = blockIdx.y;
    const int nwords = G.h * G.ww;
    const size_t fo = (size_t)f * nwords;
    const u64* fb = bits + fo;
    const u64* hm = hmaps + fo * 4;
    const u32* hb = hbase + fo;
    const u32* hr = hrank + (size_t)f * hcap;
    unsigned long long* nd = node + (size_t)f * hcap;
    const u32 H = aux[f].nheads;
    for (u32 k = blockIdx.x * 256 + threadIdx.x; k < H; k += gridDim.x * 256) {
        int32_t* out = nullptr;
        if (WRITE) {
            const unsigned long long v = nd[k];
            const u32 J = (u32)(v >> 32);
            if (!(J & CT_TERM)) continue;               // did not converge (never seen; see CT_JUMP_ROUNDS)
            const u32 T = J & ~CT_TERM;
            const u32 r = hr[T];
            if (r >= (u32)max_contours) continue;       // CT_UNSEL, or beyond the caller's capacity
            const u32 total = (u32)nd[T];
            const long long base = offsets[(size_t)f * max_contours + r];
            if (base + (long long)total > max_points) continue;
            const u32 off = (k == T) ? 0u : total - (u32)v;
            out = points + 2 * ((size_t)f * max_points + base + off);
        }
        const u32 hp = head_pix[(size_t)f * hcap + k];
        const int pix = (int)(hp & 0x1fffffffu), type = (int)(hp >> 29);
        int y = pix / G.w, x = pix - y * G.w;
        ct_tile T;
        ct_tile_load(G, fb, y - 3, x - 3, T);
        u32 R = ct_ring(T, y, x);
        int s = ct_first_cw(R, type == 0 ? 4 : (type == 1 ? 0 : (type == 2 ? 2 : 6)));
        u32 cnt = 0, succ = k;
        bool first = true;
        // a border visits a pixel at most once per incoming direction: bound the walk so that a corrupted image cannot
        // keep the wave alive forever
        for (long long guard = 8ll * G.w * G.h + 16; guard > 0; guard--) {
            const u32 q = (R | (R << 8)) >> (s + 1);
            const int t = __ffs((int)q) - 1;
            if (!first) {
                const int ht = ct_head_type(s, t, x);
                if (ht >= 0) { succ = ct_head_index(hm, hb, y * G.ww + (x >> 6), x & 63, ht); break; }
            }
            first = false;
            const int s2 = (s + 1 + t) & 7;
            if (s2 != (s ^ 4) || method == 1) {
                if (WRITE) { out[2 * cnt] = x; out[2 * cnt + 1] = y; }
                cnt++;
            }
            x += dx8(s2); y += dy8(s2);
            s = (s2 + 4) & 7;
            if (!ct_tile_covers(T, y, x)) {
                const int dy = dy8(s2), dx = dx8(s2);
                ct_tile_load(G, fb, y - (dy > 0 ? 1 : (dy < 0 ? 6 : 3)), x - (dx > 0 ? 1 : (dx < 0 ? 6 : 3)), T);
            }
            R = ct_ring(T, y, x);
        }
        if (!WRITE) {
            const u32 term = (hr[succ] != CT_NONE) ? CT_TERM : 0u;
            nd[k] = ((unsigned long long)(succ | term) << 32) | cnt;
        }
    }
}

// per frame: pointer jumping.  node = (J, D): D points lie between this head and head J along the border.  A pair read in one
// 64-bit load is always a consistent (older or newer) statement of that invariant, so the rounds need no double buffering.
// Up to CTJ_LDS heads the table lives in LDS for the rounds.
#define CTJ_LDS 8192
// per frame: contour lengths from the terminals, exclusive scan -> offsets, total -> out[f].n_points; single-pixel contours are
// written here.  All NT threads of the block take part.
template <int NT>
__device__ __forceinline__ void ct_offsets_body(const ccl_geom& G, const ct_aux* __restrict__ aux, const u32* __restrict__ starts,
                                                const u32* __restrict__ shead, const unsigned long long* __restrict__ node, size_t hcap,
                                                int32_t* __restrict__ counts, uint8_t* __restrict__ is_hole_out, int32_t* __restrict__ offsets,
                                                int32_t* __restrict__ points, ct_frame_out* __restrict__ out, int max_contours, long long max_points)
{
    __shared__ u32 wsum[NT / 64];
    __shared__ u32 carry;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nsel = (int)aux[f].nsel;
    const int K = min(nsel, max_contours);
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < K; base += NT) {
        const int i = base + tid;
        u32 v = 0, st = 0, sh = 0;
        if (i < K) {
            st = starts[(size_t)f * max_contours + i];
            sh = shead[(size_t)f * max_contours + i];
            v = (sh == CT_NONE) ? 1u : (u32)node[(size_t)f * hcap + sh];
            counts[(size_t)f * max_contours + i] = (int32_t)v;
            is_hole_out[(size_t)f * max_contours + i] = (uint8_t)(st >> 31);
        }
        u32 inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        u32 woff = 0;
        for (int k = 0; k < wv; k++) woff += wsum[k];
        if (i < K) {
            const u32 off = carry + woff + inc - v;
            offsets[(size_t)f * max_contours + i] = (int32_t)off;
            if (sh == CT_NONE && (long long)off + 1 <= max_points) {
                const int pix = (int)(st & 0x7fffffffu);
                const int y = pix / G.w;
                int32_t* p = points + 2 * ((size_t)f * max_points + off);
                p[0] = pix - y * G.w; p[1] = y;
            }
        }
        __syncthreads();
        if (tid == NT - 1) carry += woff + inc;
        __syncthreads();
    }
    if (tid == 0) { out[f].n_contours = nsel; out[f].n_points = (int32_t)carry; }
}

// One block per frame: pointer jumping over the (next head, distance) pairs until every head points at its terminal, then the
// contour lengths / offsets from the terminals (what used to be a launch of its own).
__global__ __launch_bounds__(1024) void k_ct_jump(ccl_geom G, const ct_aux* __restrict__ aux, unsigned long long* __restrict__ node, size_t hcap,
                                                  const u32* __restrict__ starts, const u32* __restrict__ shead, int32_t* __restrict__ counts,
                                                  uint8_t* __restrict__ is_hole_out, int32_t* __restrict__ offsets, int32_t* __restrict__ points,
                                                  ct_frame_out* __restrict__ out, int max_contours, long long max_points)
{
    __shared__ unsigned long long tab[CTJ_LDS];
    const int f = blockIdx.x;
    const u32 H = aux[f].nheads;
    unsigned long long* nd = node + (size_t)f * hcap;
    // A cycle that holds a terminal is through after ceil(log2(its heads)) rounds; a cycle without one (a hole border whose start was not
    // looked for: k_ct_seeds with `nobg`) never is - so the rounds are bounded by the head count, not only by "nothing changed".
    const int max_rounds = min(CT_JUMP_ROUNDS, 34 - __clz((int)(H | 1u)));
    if (H <= CTJ_LDS) {
        for (u32 k = threadIdx.x; k < H; k += 1024) tab[k] = nd[k];
        __syncthreads();
        for (int round = 0; round < max_rounds; round++) {
            int changed = 0;
            for (u32 k = threadIdx.x; k < H; k += 1024) {
                const unsigned long long v = tab[k];
                const u32 J = (u32)(v >> 32);
                if (J & CT_TERM) continue;
                const unsigned long long v2 = tab[J];
                tab[k] = (v2 & 0xffffffff00000000ull) | (u32)((u32)v + (u32)v2);
                changed = 1;
            }
            if (!__syncthreads_or(changed)) break;
        }
        for (u32 k = threadIdx.x; k < H; k += 1024) nd[k] = tab[k];
    } else {
        for (int round = 0; round < max_rounds; round++) {
            int changed = 0;
            for (u32 k = threadIdx.x; k < H; k += 1024) {
                const unsigned long long v = __hip_atomic_load(nd + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u32 J = (u32)(v >> 32);
                if (J & CT_TERM) continue;
                const unsigned long long v2 = __hip_atomic_load(nd + J, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(nd + k, (v2 & 0xffffffff00000000ull) | (u32)((u32)v + (u32)v2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                changed = 1;
            }
            if (!__syncthreads_or(changed)) break;
        }
    }
    __threadfence_block();
    __syncthreads();               // the block's own stores to node[] are visible to all its threads from here on
    ct_offsets_body<1024>(G, aux, starts, shead, node, hcap, counts, is_hole_out, offsets, points, out, max_contours, max_points);
}
#undef dx8
#undef dy8

// per contour: polygon moments by Green's formula (imgproc/src/moments.cpp contourMoments: a00 = sum(x[i-1]*y[i] - x[i]*y[i-1]),
// a10 = sum(dxy * (x[i-1] + x[i])), a01 likewise; m00 = a00/2, m10 = a10/6, m01 = a01/6, all with the sign of a00), cv2.contourArea
// and the bounding box.  The sums are integers, accumulated in int64, so the order of the additions does not matter and the
// doubles are the ones a sequential CPU loop produces.  grid (max_contours, n), 64 threads: one wave per contour
__global__ __launch_bounds__(64) void k_ct_features(const ct_frame_out* __restrict__ info, const int32_t* __restrict__ counts,
                                                    const int32_t* __restrict__ offsets, const int32_t* __restrict__ points, int max_contours,
                                                    long long max_points, double* __restrict__ features)
{
    const int f = blockIdx.y, r = blockIdx.x, lane = threadIdx.x;
    double* o = features + ((size_t)f * max_contours + r) * 8;
    const int K = min(info[f].n_contours, max_contours);
    const int cnt = r < K ? counts[(size_t)f * max_contours + r] : 0;
    const long long off = r < K ? offsets[(size_t)f * max_contours + r] : 0;
    if (cnt <= 0 || off + cnt > max_points) {
        if (lane < 8) o[lane] = 0.0;
        return;
    }
    const int32_t* p = points + 2 * ((size_t)f * max_points + off);
    long long a00 = 0, a10 = 0, a01 = 0;
    int minx = INT_MAX, miny = INT_MAX, maxx = INT_MIN, maxy = INT_MIN;
    for (int i = lane; i < cnt; i += 64) {
        const int ip = i == 0 ? cnt - 1 : i - 1;
        const long long x = p[2 * i], y = p[2 * i + 1], xp = p[2 * ip], yp = p[2 * ip + 1];
        const long long dxy = xp * y - x * yp;
        a00 += dxy;
        a10 += dxy * (xp + x);
        a01 += dxy * (yp + y);
        minx = min(minx, (int)x); maxx = max(maxx, (int)x);
        miny = min(miny, (int)y); maxy = max(maxy, (int)y);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a00 += __shfl_xor(a00, d); a10 += __shfl_xor(a10, d); a01 += __shfl_xor(a01, d);
        minx = min(minx, __shfl_xor(minx, d)); maxx = max(maxx, __shfl_xor(maxx, d));
        miny = min(miny, __shfl_xor(miny, d)); maxy = max(maxy, __shfl_xor(maxy, d));
    }
    if (lane == 0) {
        const double d00 = (double)a00, d10 = (double)a10, d01 = (double)a01;
        double m00 = 0, m10 = 0, m01 = 0;
        if (fabs(d00) > 1.1920929e-07) {
            const double s2 = d00 > 0 ? 0.5 : -0.5, s6 = d00 > 0 ? 1.0 / 6 : -1.0 / 6;
            m00 = d00 * s2; m10 = d10 * s6; m01 = d01 * s6;
        }
        o[0] = m00; o[1] = m10; o[2] = m01; o[3] = fabs(d00 * 0.5);
        o[4] = (double)minx; o[5] = (double)miny; o[6] = (double)(maxx - minx + 1); o[7] = (double)(maxy - miny + 1);
    }
}

int vpk_contour_features(vp_ctx* ctx, const int32_t* d_info, const int32_t* d_counts, const int32_t* d_offsets, const int32_t* d_points, int n,
                         int max_contours, long long max_points, double* d_features)
{
    vp_prof_scope ps(ctx, VPK_OTHER);
    hipLaunchKernelGGL(k_ct_features, dim3((unsigned)max_contours, (unsigned)n), dim3(64), 0, ctx->stream, reinterpret_cast<const ct_frame_out*>(d_info),
                       d_counts, d_offsets, d_points, max_contours, max_points, d_features);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

size_t vp_contours_ws_bytes(int w, int h, int n, int max_contours)
{
    const size_t nids = vp_ccl_nids(w, h);
    const size_t words = (size_t)n * h * vp_ww(w);
    const size_t hcap = ct_hcap(w, h) * n;
    return 2 * vp_align(nids * 4 * n) + 3 * vp_align(nids / 8 * n) + 3 * vp_align(words * 8) + vp_align(words * 32) + 2 * vp_align(words * 4) +
           2 * vp_align(hcap * 4) + vp_align(hcap * 8) + 2 * vp_align((size_t)n * max_contours * 4) + vp_align(sizeof(ct_aux) * n) + 2 * vp_align(words / 64 + 4 * n) + vp_align((size_t)n * 4) + 8192;
}

// Root bitmap of the foreground from a labelling that already exists: a component's first pixel in raster order lies in the top row of
// its box and is the first pixel of that row that carries its label; the segment starting there is the component's smallest segment id
// (pixel numbering) - the bit the foreground union-find would have left.  One wave per label; frames whose statistics table was too
// small for their labels are marked in `only` and go through the union-find as before.  grid (ceil((max_labels - 1) / 4), n) x 256.
__global__ __launch_bounds__(256) void k_ct_roots_from_labels(ccl_geom G, const int32_t* __restrict__ labels, const int32_t* __restrict__ stats,
                                                              const int32_t* __restrict__ nlabels, int max_labels, u32* __restrict__ flags,
                                                              u32* __restrict__ only)
{
    const int f = blockIdx.y, lane = threadIdx.x & 63;
    const int nl = nlabels[f];
    if (blockIdx.x == 0 && threadIdx.x == 0) only[f] = nl > max_labels ? 1u : 0u;
    if (nl > max_labels) return;
    const int L = blockIdx.x * 4 + (threadIdx.x >> 6) + 1;
    if (L >= nl) return;
    const int32_t* st = stats + ((size_t)f * max_labels + L) * 5;
    const int x0 = st[0], y = st[1], x1 = st[0] + st[2];
    const int32_t* row = labels + ((size_t)f * G.h + y) * G.w;
    for (int xb = x0; xb < x1; xb += 64) {
        const int x = xb + lane;
        const unsigned long long m = __ballot(x < x1 && row[x] == L);
        if (m) {
            if (lane == 0) {
                const u32 id = seg_id(G, y, xb + __ffsll((long long)m) - 1);
                atomicOr(flags + (size_t)f * G.nw32 + (id >> 5), 1u << (id & 31));
            }
            return;
        }
    }
}

// RETR_EXTERNAL with the components' boxes at hand (the chain has just labelled this mask): a component inside a hole of another one
// has that one's pixels on all four sides, so its box lies STRICTLY inside the other's.  A frame without such a pair of boxes has only
// external components: its background - the long half of a contour pass - is not needed at all.  bg_only[f] = 1: resolve it (a pair
// exists, or the statistics table did not hold all labels, or there are too many labels to compare in passing).  grid n x 256.
#define CT_NEEDS_BG_MAX 512
__global__ __launch_bounds__(256) void k_ct_needs_bg(const int32_t* __restrict__ stats, const int32_t* __restrict__ nlabels, int max_labels,
                                                     u32* __restrict__ bg_only)
{
    const int f = blockIdx.x;
    const int nl = nlabels[f];
    if (nl > max_labels || nl > CT_NEEDS_BG_MAX) { if (threadIdx.x == 0) bg_only[f] = 1u; return; }
    const int32_t* st = stats + (size_t)f * max_labels * 5;
    int found = 0;
    for (int a = 1 + (int)threadIdx.x; a < nl && !found; a += 256) {
        const int ax0 = st[a * 5], ay0 = st[a * 5 + 1], ax1 = ax0 + st[a * 5 + 2] - 1, ay1 = ay0 + st[a * 5 + 3] - 1;
        for (int b = 1; b < nl; b++) {
            const int bx0 = st[b * 5], by0 = st[b * 5 + 1], bx1 = bx0 + st[b * 5 + 2] - 1, by1 = by0 + st[b * 5 + 3] - 1;
            if (bx0 < ax0 && by0 < ay0 && bx1 > ax1 && by1 > ay1) { found = 1; break; }
        }
    }
    const int any = __syncthreads_or(found);
    if (threadIdx.x == 0) bg_only[f] = any ? 1u : 0u;
}

// A contour pass in three steps, so that a caller with other work for the context's stream (the chain: its own labelling and the
// label write, 260 us per 128 frames) can put it between the second and the third:
//   ct_pass_setup        geometry, scratch carved from the workspace
//   ct_pass_background   the background half - 4-connected union-find of the inverted mask, which regions reach the frame - queued
//                        on the context's SIDE stream behind everything queued so far on its own stream
//   ct_pass_finish       the foreground half, the join, seeds, starts, follower passes on the context's stream
// vpk_find_contours runs the three back to back.
static bool ct_known_usable(const vp_known_labels* known, const ccl_geom& Gf)
{
    size_t cap_unused;
    static const bool known_off = getenv("VP_CT_KNOWN") && atoi(getenv("VP_CT_KNOWN")) == 0;
    return known && !known_off && known->labels && known->stats && known->nlabels && known->max_labels >= 2 && known->max_labels <= 4096 &&
           ccl_local_lds(Gf, cap_unused) <= 64 * 1024;
}

struct ct_pass {
    ccl_geom Gf, Gb;
    int w, h, n, nwords;
    size_t hcap, nparts, mstride;
    u32 *fg_parent, *bg_parent, *fg_flags, *bg_flags, *outside, *hbase, *head_pix, *hrank, *starts, *shead, *partsum, *partsum2;
    uint8_t* cnt8;             // heads per word
    u64 *maps3, *hmaps;
    unsigned long long* node;
    ct_aux* aux;
    u32* bg_only;              // nullable, per frame: 1 = the background is resolved, 0 = not needed (k_ct_needs_bg)
    hipError_t bg_join;        // result of recording the side stream's end event
    bool bg_queued;
    int rc_bg;                 // what queueing the background half returned (vpk_contours_begin / _finish)
};

static int ct_pass_setup(vp_ctx* ctx, int w, int h, int n, int max_contours, ct_pass* P)
{
    if ((size_t)w * h >= (1u << 29)) return vp_fail(ctx, VP_ERR_INVALID, "contours: image too large");
    ccl_geom& Gf = P->Gf;
    ccl_geom& Gb = P->Gb;
    ccl_make_geom(Gf, w, h, VP_CCL_PIXEL, 0, 0);
    ccl_make_geom(Gb, w, h, VP_CCL_PIXEL, 1, 1);
    // One image (a module's findContours call) leaves most of the chip idle with 32-row strips (34 blocks at 1080p), and a block's
    // time grows faster than its strip: shorter strips while the blocks still fit the CUs (1080p, one image: 0.184 -> 0.161 ms per
    // call with 8 rows).  The strip height only moves work between the strip pass and the boundary pass; parent[] and the root bitmap
    // come out the same.
    if (!getenv("VP_CL_ROWS"))
        for (int r = 8; r < Gf.rows; r *= 2)
            if (((u32)r * (u32)Gf.wb) % 32u == 0 && (size_t)n * ((h + r - 1) / r) <= (size_t)2 * ctx->num_cu) { Gf.rows = Gb.rows = r; break; }
    P->w = w; P->h = h; P->n = n;
    const size_t nids = Gf.nids;
    P->nwords = h * Gf.ww;
    const size_t words = (size_t)n * P->nwords;
    P->hcap = ct_hcap(w, h);
    P->fg_parent = (u32*)vp_ws_take(ctx, nids * 4 * n);
    P->bg_parent = (u32*)vp_ws_take(ctx, nids * 4 * n);
    P->fg_flags = (u32*)vp_ws_take(ctx, nids / 8 * n);
    P->bg_flags = (u32*)vp_ws_take(ctx, nids / 8 * n);
    P->outside = (u32*)vp_ws_take(ctx, nids / 8 * n);
    P->maps3 = (u64*)vp_ws_take(ctx, 3 * vp_align(words * 8));   // startmap, holemap, selmap: cleared together
    P->hmaps = (u64*)vp_ws_take(ctx, words * 32);
    P->hbase = (u32*)vp_ws_take(ctx, words * 4);
    u32* sbase = (u32*)vp_ws_take(ctx, words * 4);   // (the first quarter holds the per-word head counts; the rest is unused since k_ct_starts ranks the starts itself)
    P->cnt8 = reinterpret_cast<uint8_t*>(sbase);
    P->head_pix = (u32*)vp_ws_take(ctx, P->hcap * n * 4);
    P->hrank = (u32*)vp_ws_take(ctx, P->hcap * n * 4);
    P->node = (unsigned long long*)vp_ws_take(ctx, P->hcap * n * 8);
    P->starts = (u32*)vp_ws_take(ctx, (size_t)n * max_contours * 4);
    P->shead = (u32*)vp_ws_take(ctx, (size_t)n * max_contours * 4);
    P->aux = (ct_aux*)vp_ws_take(ctx, sizeof(ct_aux) * n);
    P->nparts = (size_t)(P->nwords + 255) / 256;
    P->partsum = (u32*)vp_ws_take(ctx, P->nparts * n * 4);
    P->partsum2 = (u32*)vp_ws_take(ctx, P->nparts * n * 4);
    if (!P->partsum || !P->partsum2 || !P->fg_parent || !P->bg_parent || !P->fg_flags || !P->bg_flags || !P->outside || !P->maps3 || !P->hmaps || !P->hbase ||
        !sbase || !P->head_pix || !P->hrank || !P->node || !P->starts || !P->shead || !P->aux)
        return vp_fail(ctx, VP_ERR_NOMEM, "contour workspace");
    P->mstride = vp_align(words * 8) / 8;
    P->bg_only = nullptr;
    P->bg_queued = false;
    P->bg_join = hipSuccess;
    return VP_OK;
}

// Background regions (4-connected union-find, then which of them reach the frame) on the context's side stream.  Always leaves the
// side stream's end recorded in ev_fb_join (also after an error), for ct_pass_finish to wait on.
static int ct_pass_background(vp_ctx* ctx, const u64* d_bits, ct_pass* P)
{
    hipStream_t s = ctx->stream;
    hipStream_t side = ctx->fb_stream;
    VP_HIP(ctx, hipEventRecord(ctx->ev_fb_fork, s));
    VP_HIP(ctx, hipStreamWaitEvent(side, ctx->ev_fb_fork, 0));
    ctx->stream = side;
    int rc = ccl_roots(ctx, d_bits, P->Gb, P->n, P->bg_parent, P->bg_flags, P->bg_only, P->outside);   // (clears `outside` on the way)
    ctx->stream = s;
    if (rc == VP_OK)
        hipLaunchKernelGGL(k_ct_outside, dim3((unsigned)((2 * P->Gb.ww + 2 * P->h + 255) / 256), (unsigned)P->n), dim3(256), 0, side, d_bits, P->Gb, P->bg_parent,
                           P->outside, P->bg_only);
    P->bg_join = hipEventRecord(ctx->ev_fb_join, side);
    P->bg_queued = true;
    return rc;
}

// d_counts / d_is_hole / d_offsets: [n][max_contours]; d_points: [n][max_points][2]; d_info: [n] {n_contours, n_points}.  Contours are
// stored in discovery order (raster order of the start pixel); cv2 returns them reversed - the caller reverses.
// `rc_bg`: what ct_pass_background returned (the side stream is joined whatever happened).
static int ct_pass_finish(vp_ctx* ctx, const u64* d_bits, ct_pass* P, int rc_bg, int mode, int method, int32_t* d_counts, uint8_t* d_is_hole,
                          int32_t* d_offsets, int32_t* d_points, int max_contours, long long max_points, int32_t* d_info, const vp_known_labels* known)
{
    const ccl_geom& Gf = P->Gf;
    const ccl_geom& Gb = P->Gb;
    const int n = P->n, w = P->w, nwords = P->nwords;
    const size_t hcap = P->hcap, nparts = P->nparts, mstride = P->mstride;
    u32 *fg_parent = P->fg_parent, *bg_parent = P->bg_parent, *fg_flags = P->fg_flags, *bg_flags = P->bg_flags, *outside = P->outside, *hbase = P->hbase,
        *head_pix = P->head_pix, *hrank = P->hrank, *starts = P->starts, *shead = P->shead, *partsum = P->partsum, *partsum2 = P->partsum2;
    u64 *maps3 = P->maps3, *hmaps = P->hmaps;
    unsigned long long* node = P->node;
    ct_aux* aux = P->aux;
    u64* startmap = maps3;
    u64* holemap = maps3 + mstride;
    u64* selmap = maps3 + 2 * mstride;
    hipStream_t s = ctx->stream;
    ct_frame_out* info = reinterpret_cast<ct_frame_out*>(d_info);
    vp_prof_scope ps(ctx, VPK_OTHER);
    const dim3 wgrid((unsigned)((nwords + 255) / 256), (unsigned)n);
    // heads per frame are not known on the host: a fixed number of blocks per frame walks the head list (a real mask has a few
    // thousand heads; empty blocks of a grid sized for the worst case would cost more than the work)
    const dim3 hgrid((unsigned)std::min<size_t>((hcap + 255) / 256, (size_t)std::max(32, std::min(1024, 8192 / n))), (unsigned)n);
    int rc = rc_bg;
    {
        // the foreground's root bitmap: from the caller's labelling of the same mask when there is one (frames it could not hold keep
        // the union-find through its per-frame switch), otherwise the union-find for every frame
        const bool use_known = ct_known_usable(known, Gf);
        if (rc == VP_OK && use_known) {
            // (no early return in here: the side stream is joined below whatever happens, and the caller reuses the scratch after an error)
            u32* only = (u32*)vp_ws_take(ctx, (size_t)n * 4);
            if (!only) rc = vp_fail(ctx, VP_ERR_NOMEM, "contour workspace");
            if (rc == VP_OK) {
                const hipError_t em = hipMemsetAsync(fg_flags, 0, (size_t)Gf.nw32 * 4 * n, s);
                if (em != hipSuccess) rc = vp_fail(ctx, VP_ERR_HIP, "hipMemsetAsync (root bitmap)", em);
            }
            if (rc == VP_OK) {
                hipLaunchKernelGGL(k_ct_roots_from_labels, dim3((unsigned)((known->max_labels - 1 + 3) / 4), (unsigned)n), dim3(256), 0, s, Gf, known->labels,
                                   known->stats, known->nlabels, known->max_labels, fg_flags, only);
                rc = ccl_roots(ctx, d_bits, Gf, n, fg_parent, fg_flags, only);
            }
        } else if (rc == VP_OK) {
            rc = ccl_roots(ctx, d_bits, Gf, n, fg_parent, fg_flags);
        }
    }
    if (rc == VP_OK) {
        hipLaunchKernelGGL(k_ct_headmaps, wgrid, dim3(256), 0, s, d_bits, Gf, hmaps, partsum, maps3, mstride, partsum2, P->cnt8);
        hipLaunchKernelGGL(k_ct_prefix, wgrid, dim3(256), 0, s, hmaps, 4, nwords, partsum, hbase, &aux->nheads, 2, w, Gf.ww, head_pix, hrank, hcap, P->cnt8);
    }
    // join whatever was queued on the side stream, also after an error
    const hipError_t j1 = P->bg_queued ? P->bg_join : hipSuccess;
    const hipError_t j2 = (P->bg_queued && j1 == hipSuccess) ? hipStreamWaitEvent(s, ctx->ev_fb_join, 0) : j1;
    P->bg_queued = false;
    if (rc != VP_OK) return rc;
    if (j1 != hipSuccess) return vp_fail(ctx, VP_ERR_HIP, "hipEventRecord", j1);
    if (j2 != hipSuccess) return vp_fail(ctx, VP_ERR_HIP, "hipStreamWaitEvent", j2);
    hipLaunchKernelGGL(k_ct_seeds, dim3((unsigned)((Gf.nw32 + 255) / 256), (unsigned)n), dim3(256), 0, s, d_bits, Gf, Gb, fg_flags, bg_flags, bg_parent,
                       outside, mode, startmap, holemap, selmap, partsum2, (int)nparts, P->bg_only);
    hipLaunchKernelGGL(k_ct_starts, wgrid, dim3(256), 0, s, d_bits, Gf, startmap, holemap, selmap, partsum2, &aux->nsel, hmaps, hbase, hrank, hcap, starts,
                       shead, max_contours);
    hipLaunchKernelGGL((k_ct_seg<false>), hgrid, dim3(256), 0, s, d_bits, Gf, hmaps, hbase, head_pix, hrank, hcap, aux, node, method, d_offsets, d_points,
                       max_contours, max_points);
    hipLaunchKernelGGL(k_ct_jump, dim3((unsigned)n), dim3(1024), 0, s, Gf, aux, node, hcap, starts, shead, d_counts, d_is_hole, d_offsets, d_points, info,
                       max_contours, max_points);
    hipLaunchKernelGGL((k_ct_seg<true>), hgrid, dim3(256), 0, s, d_bits, Gf, hmaps, hbase, head_pix, hrank, hcap, aux, node, method, d_offsets, d_points,
                       max_contours, max_points);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// The same in two calls, for a caller that has other work for the context's stream in between (vp_api.hip chain_core: the background
// half is queued as soon as the mask exists and runs beside the chain's own labelling and label write).  `*pass` is owned by the pair:
// vpk_contours_finish frees it (also on failure); a caller that cannot reach finish calls it with d_counts == NULL to join and free.
int vpk_contours_begin(vp_ctx* ctx, const u64* d_bits, int w, int h, int n, int max_contours, void** pass)
{
    ct_pass* P = new (std::nothrow) ct_pass();
    if (!P) return vp_fail(ctx, VP_ERR_NOMEM, "contour pass");
    int rc = ct_pass_setup(ctx, w, h, n, max_contours, P);
    if (rc != VP_OK) { delete P; *pass = nullptr; return rc; }
    P->rc_bg = ct_pass_background(ctx, d_bits, P);
    *pass = P;
    return VP_OK;          // (a failure of the background half is reported by finish, after the join)
}

int vpk_contours_finish(vp_ctx* ctx, void* pass, const u64* d_bits, int mode, int method, int32_t* d_counts, uint8_t* d_is_hole, int32_t* d_offsets,
                        int32_t* d_points, int max_contours, long long max_points, int32_t* d_info, const vp_known_labels* known)
{
    ct_pass* P = static_cast<ct_pass*>(pass);
    if (!P) return vp_fail(ctx, VP_ERR_INVALID, "contour pass");
    int rc;
    if (!d_counts) {       // abandoned: only the join
        rc = (P->bg_queued && P->bg_join == hipSuccess && hipStreamWaitEvent(ctx->stream, ctx->ev_fb_join, 0) != hipSuccess) ? VP_ERR_HIP : VP_OK;
    } else if ((mode != 0 && mode != 1) || (method != 1 && method != 2)) {
        (void)(P->bg_queued && P->bg_join == hipSuccess && hipStreamWaitEvent(ctx->stream, ctx->ev_fb_join, 0));
        rc = vp_fail(ctx, VP_ERR_INVALID, "contour mode / approximation");
    } else {
        rc = ct_pass_finish(ctx, d_bits, P, P->rc_bg, mode, method, d_counts, d_is_hole, d_offsets, d_points, max_contours, max_points, d_info, known);
    }
    delete P;
    return rc;
}

int vpk_find_contours(vp_ctx* ctx, const u64* d_bits, int w, int h, int n, int mode, int method, int32_t* d_counts, uint8_t* d_is_hole,
                      int32_t* d_offsets, int32_t* d_points, int max_contours, long long max_points, int32_t* d_info, const vp_known_labels* known)
{
    if (mode != 0 && mode != 1) return vp_fail(ctx, VP_ERR_INVALID, "contour mode");
    if (method != 1 && method != 2) return vp_fail(ctx, VP_ERR_INVALID, "contour approximation");
    ct_pass P;
    const int rc_setup = ct_pass_setup(ctx, w, h, n, max_contours, &P);
    if (rc_setup != VP_OK) return rc_setup;
    // Two independent halves up to the seeds: background regions on the context's side stream, foreground components and the head
    // bitmaps on its own stream.  Each half is a chain of short, latency-bound launches, so side by side they take the time of one
    // (one 1080p frame: 0.19 -> 0.15 ms per call).
    static const bool skip_off = getenv("VP_CT_SKIP_BG") && atoi(getenv("VP_CT_SKIP_BG")) == 0;
    if (mode == 0 && !skip_off && ct_known_usable(known, P.Gf)) {
        // RETR_EXTERNAL of a mask whose components' boxes are known: frames without nested boxes skip the background half
        P.bg_only = (u32*)vp_ws_take(ctx, (size_t)n * 4);
        if (!P.bg_only) return vp_fail(ctx, VP_ERR_NOMEM, "contour workspace");
        hipLaunchKernelGGL(k_ct_needs_bg, dim3((unsigned)n), dim3(256), 0, ctx->stream, known->stats, known->nlabels, known->max_labels, P.bg_only);
    }
    const int rc_bg = ct_pass_background(ctx, d_bits, &P);
    return ct_pass_finish(ctx, d_bits, &P, rc_bg, mode, method, d_counts, d_is_hole, d_offsets, d_points, max_contours, max_points, d_info, known);
}
