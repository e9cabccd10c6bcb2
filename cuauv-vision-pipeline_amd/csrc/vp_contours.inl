// Contour extraction on the GPU (included by vp_ccl.hip).
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
//
// No labelling is needed for any of it (rounds 1-3 ran two union-finds, foreground and background, to find the start pixels;
// they were two thirds of a contour pass).  A border is a cycle of follower states (below), every cycle of the follower map is a
// border, and the cycle itself says which one:
//   * Take the cycle's vertical cracks (edges between a foreground pixel and a background pixel left or right of it) and order them
//     as the scan meets them: by row, then by the x of the crack.  The smallest one is either the W crack of the component's first
//     pixel - every other crack of an outer border belongs to a later pixel of the component - or the E crack of the pixel left of the
//     background region's first pixel - every other crack of a hole border touches a later pixel of the region.  So: smallest
//     crack is a W crack = outer border, start = its pixel; an E crack = hole border, start = its (foreground) pixel.  The state that
//     sweeps that crack is OpenCV's start state (icvFetchContour: first neighbour clockwise from W / from E).
//   * Scan order of the cracks = raster order of the start pixels (an outer and a hole border never start at the same pixel).
//   * RETR_EXTERNAL: a component is inside a hole iff the background region left of its first pixel does not reach the frame.  Walk
//     left from the first pixel along its row: no foreground pixel = the region reaches the frame; otherwise the E crack of the first
//     foreground pixel met belongs to a border of the SAME region: a hole border = inside a hole; the outer border of another
//     component = that component lies in the same region, so the answer is its answer (a chain of "same as" that strictly
//     decreases in scan order, resolved by pointer jumping).
// OpenCV decides "inside a hole" from the sign of the last border mark left of a start pixel.  A mark is negative exactly when the
// tracer examined the pixel's east neighbour as background, i.e. when the east crack belongs to the traced border - and every crack
// belongs to exactly one border - so the mark rule and the rule above select the same borders (no difference on 460 noise /
// thin-wall masks: tests/test_gpu_contours.py, tools/exp_external_rule.py).

struct ct_frame_out {      // per frame, device
    int32_t n_contours;
    int32_t n_points;
};

// ---- segment-parallel border following --------------------------------------------------------------------------------
// A border is a cycle of states (pixel, s) of the Suzuki-Abe follower, s = direction of the pixel it came from; the next state
// depends on the 3x3 neighbourhood only: s' = first foreground neighbour counter-clockwise after s, move there, s = s' + 4.
// Between s and s' the follower sweeps over background neighbours.  Every crack (edge between a foreground pixel and a
// 4-adjacent background pixel) is swept by exactly one state of exactly one border, and that state can be written down from
// the crack alone: s = first foreground neighbour clockwise from the crack's direction.  So the states that sweep a W or E
// crack in every second (or fourth) row or an N or S crack in every fourth (or eighth) column - ct_cut - (and every W or E crack that
// a border could start at) are
// enumerable with bit operations - the "heads" - and they cut every border into short segments that are followed independently, one
// thread each.  A pixel without neighbours is a
// border of its own: one head (listed with the W heads), one point.
//   k_ct_headmaps   4 head bitmaps per word (a state that sweeps several eligible cracks belongs to the first one swept)
//   k_ct_prefix     popcount prefix -> dense head index + head list (pixel, type)
//   k_ct_seg<false> follow each segment to the next head: node[k] = (next head, points emitted); key[k] = the smallest vertical
//                   crack the head's state sweeps (none: an N / S head); for RETR_EXTERNAL, where the row left of a possible first
//                   pixel ends: the head that owns the E crack met, or the frame
//   k_ct_jump       one block per frame.  (1) pointer jumping on (next, smallest key so far) until every head knows the head with
//                   the smallest key of its cycle - the leader, whose state is the border's start state; (2) RETR_EXTERNAL: which
//                   outer borders are external; (3) rank of the returned borders in scan order (head order is scan order up to the
//                   order inside one 64-pixel word, which is settled by comparing keys) -> starts / terminal marks; (4) pointer
//                   jumping on (next, distance) until every head points at its leader; (5) per contour: length, exclusive scan
//                   -> offsets
//   k_ct_seg<true>  follow each segment again, writing its points at offset[contour] + (length - distance to the leader)
// CHAIN_APPROX_SIMPLE (keep a point when the direction changes) is local to a state: s' != s ^ 4.
#define CT_TERM 0x80000000u
#define CT_NONE 0xffffffffu
#define CT_UNSEL 0xfffffffeu
#define CT_FRAME 0xffffffffu      // ext[]: nothing but background left of the first pixel / the border is external
#define CT_INSIDE 0xfffffffeu     // ext[]: inside a hole
#define CT_JUMP_ROUNDS 40
// Spacing of the cuts.  DENSE (a module's mask): W / E cracks are heads in rows y % 2 == 0, N / S cracks in columns x % 4 == 0 - short
// walks, what the follower passes of a clean mask or a batch wait for (8 / 4, 8 / 2, 4 / 2 for columns / rows: chain + contours of 128
// frames 0.75 / 0.715 / 0.69 ms).  SPARSE (a frame expected to be speckled - the caller's `many_heads`): rows y % 4, columns x % 8 -
// fewer heads, what the bookkeeping of 600 k of them pays for (10 % noise, one image: 1.57 against 1.75 ms).
template <bool SPARSE> struct ct_cut {
    static constexpr int row_mask = SPARSE ? 3 : 1;
    static constexpr int col_mask = SPARSE ? 7 : 3;
    static constexpr u64 el_ns = SPARSE ? 0x0101010101010101ull : 0x1111111111111111ull;
};

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

template <bool SPARSE>
__global__ __launch_bounds__(256) void k_ct_headmaps(const u64* __restrict__ bits, ccl_geom G, u64* __restrict__ hmaps, u32* __restrict__ partsum,
                                                     uint8_t* __restrict__ cnt8)
{
    const int nwords = G.h * G.ww;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    u32 cnt = 0;
    if (idx < nwords) {
        const u64* fb = bits + (size_t)f * nwords;
        const int y = idx / G.ww, j = idx - y * G.ww;
        u64 c, n[8];
        ct_neighbours(G, fb, y, j, c, n);
        const u64 el = ct_cut<SPARSE>::el_ns;
        u64 hw = 0, he = 0, hn = 0, hs = 0;
        if (c) {
            // eligible cracks: N / S in the cut columns; W / E in the cut rows (ct_cut) and wherever the crack could be the smallest of
            // its border (W: nothing above the pixel; E: the background pixel has foreground above it) - see ct_head_type
            const u64 rowel = (y & ct_cut<SPARSE>::row_mask) == 0 ? ~0ull : 0ull;
            const u64 elw = rowel | ~(n[1] | n[2] | n[3]), ele = rowel | n[1];
            hw = c & ~n[4] & elw & (n[3] | n[2] | (~el & (n[1] | n[0] | (~ele & (n[7] | n[6] | n[5])))));
            he = c & ~n[0] & ele & (n[7] | n[6] | (~el & (n[5] | n[4] | (~elw & (n[3] | n[2] | n[1])))));
            hn = c & ~n[2] & el & (n[1] | n[0] | (~ele & (n[7] | n[6])));
            hs = c & ~n[6] & el & (n[5] | n[4] | (~elw & (n[3] | n[2])));
            hw |= c & ~(n[0] | n[1] | n[2] | n[3] | n[4] | n[5] | n[6] | n[7]);      // a pixel on its own: one state, one head
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
                                                   u32* __restrict__ head_pix, size_t hcap, const uint8_t* __restrict__ cnt8, u32* __restrict__ hint_host)
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
    if (head_pix && cnt) {   // head list: head_pix[k] = pixel index | type << 29
        const int y = i / ww, j = i - y * ww;
        const u32 pix0 = (u32)(y * w + 64 * j);
        u32 k = carry + woff + inc - cnt;
        u32* hp = head_pix + (size_t)f * hcap;
        for (int t = 0; t < 4; t++) {
            u64 m = maps[((size_t)f * nwords + i) * 4 + t];
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                hp[k] = (pix0 + (u32)b) | ((u32)t << 29);
                k++;
            }
        }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        total_out[(size_t)f * tstride] = carry + wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (hint_host) hint_host[f & 63] = carry + wsum[0] + wsum[1] + wsum[2] + wsum[3];     // (pinned: the caller's guess for its next batch)
    }
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
// state (s, sweep length t = number of background neighbours swept before s') of pixel (y, x) with neighbour ring R: type of the
// head that owns it (0 W, 1 E, 2 N, 3 S) = the first ELIGIBLE crack it sweeps, or -1 (the state is not a head).  Eligible: N / S
// cracks in the cut columns, W / E cracks in the cut rows (ct_cut) - these cut every border into segments of a few pixels - and every
// W / E crack that could be the smallest of its border: a W crack of a pixel with nothing above it (a component's first pixel is such
// a one), an E crack whose background pixel has foreground above it (a hole's first pixel is such a one) - so that every border has
// a head at its start state.  (Every vertical crack as a head: more heads than the walks need - speckle pays for them in the
// bookkeeping: 10 % noise 1.89 -> 1.59 ms with rows thinned; the one-block form does not care, it waits for latency.)
template <bool SPARSE>
__device__ __forceinline__ int ct_head_type(int s, int t, int x, int y, u32 R)
{
    const int iw = (3 - s) & 7, ie = (7 - s) & 7, in = (1 - s) & 7, is = (5 - s) & 7;
    const bool ns = (x & ct_cut<SPARSE>::col_mask) == 0, rowel = (y & ct_cut<SPARSE>::row_mask) == 0;
    const bool elw = rowel || !(R & 0xeu), ele = rowel || (R & 2u);
    int best = 8, type = -1;
    if (elw && iw < t) { best = iw; type = 0; }
    if (ele && ie < t && ie < best) { best = ie; type = 1; }
    if (ns && in < t && in < best) { best = in; type = 2; }
    if (ns && is < t && is < best) { best = is; type = 3; }
    return type;
}

// scan-order key of a vertical crack: W crack of pixel (y, x) = 2 * (y * (w + 1) + x), E crack = 2 * (y * (w + 1) + x + 1) + 1 (two
// cracks never share a position, so the low bit - "hole border if this is the smallest of its cycle" - does not disturb the order)
__device__ __forceinline__ u32 ct_key_w(const ccl_geom& G, int y, int x) { return ((u32)y * (u32)(G.w + 1) + (u32)x) << 1; }
__device__ __forceinline__ u32 ct_key_e(const ccl_geom& G, int y, int x) { return (((u32)y * (u32)(G.w + 1) + (u32)x + 1u) << 1) | 1u; }

// RETR_EXTERNAL: what lies left of a possible first pixel (y, x) in its row: CT_FRAME, or a head of the border that owns the E crack
// of the first foreground pixel met (the first head at or after the state that sweeps that crack)
template <bool SPARSE>
__device__ __forceinline__ u32 ct_left_of(const ccl_geom& G, const u64* __restrict__ fb, const u64* __restrict__ hm, const u32* __restrict__ hb,
                                          int y, int x)
{
    const u64* row = fb + (size_t)y * G.ww;
    int j = x >> 6;
    u64 m = row[j] & ((1ull << (x & 63)) - 1ull);
    while (!m && j > 0) m = row[--j];
    if (!m) return CT_FRAME;
    x = 64 * j + 63 - __clzll((long long)m);
    ct_tile T;
    ct_tile_load(G, fb, y - 3, x - 3, T);
    u32 R = ct_ring(T, y, x);
    if (!R) return ct_head_index(hm, hb, y * G.ww + (x >> 6), x & 63, 0);      // a pixel on its own: its one head is listed with the W heads
    int s = ct_first_cw(R, 0);
    for (long long guard = 8ll * G.w * G.h + 16; guard > 0; guard--) {
        const int t = __ffs((int)((R | (R << 8)) >> (s + 1))) - 1;
        const int ht = ct_head_type<SPARSE>(s, t, x, y, R);
        if (ht >= 0) return ct_head_index(hm, hb, y * G.ww + (x >> 6), x & 63, ht);
        const int s2 = (s + 1 + t) & 7;
        x += dx8(s2); y += dy8(s2);
        s = (s2 + 4) & 7;
        if (!ct_tile_covers(T, y, x)) {
            const int dy = dy8(s2), dx = dx8(s2);
            ct_tile_load(G, fb, y - (dy > 0 ? 1 : (dy < 0 ? 6 : 3)), x - (dx > 0 ? 1 : (dx < 0 ? 6 : 3)), T);
        }
        R = ct_ring(T, y, x);
    }
    return CT_FRAME;       // (not reached: every border has a head)
}

// one thread per head: follow the border from the head's state to the next head.
//   !WRITE: node[k] = next head << 32 | points emitted; key[k]; ext[k] (mode 0)
//    WRITE: the points go to their final place (see ct_offsets_body)
template <bool WRITE, bool SPARSE>
__global__ __launch_bounds__(256) void k_ct_seg(const u64* __restrict__ bits, ccl_geom G, const u64* __restrict__ hmaps, const u32* __restrict__ hbase,
                                                const u32* __restrict__ head_pix, const u32* __restrict__ hrank, size_t hcap,
                                                const ct_aux* __restrict__ aux, unsigned long long* __restrict__ node, u32* __restrict__ hkey,
                                                u32* __restrict__ hext, int mode, int method, const int32_t* __restrict__ offsets,
                                                int32_t* __restrict__ points, int max_contours, long long max_points, int32_t* __restrict__ h_points,
                                                long long h_cap, u32 skip_above)
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
    if (skip_above && H > skip_above) return;            // (the caller will repeat the pass in the other form: nothing of this one is kept)
    // !WRITE: two threads per head, in different blocks: one follows the segment, one works out the key and the look to the left (a
    // chain of dependent loads as long as a short walk: behind the walk in the same thread it was a third of this kernel)
    const u32 Hp = (H + 255u) & ~255u;
    const u32 items = WRITE ? H : 2u * Hp;
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < items; i += gridDim.x * 256) {
        const bool keys = !WRITE && i >= Hp;
        const u32 k = keys ? i - Hp : i;
        if (k >= H) continue;
        int32_t* out = nullptr;
        int32_t* hout = nullptr;                         // (WRITE, single image) the same points in the caller's pinned buffer
        long long hroom = 0;
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
            if (off >= total) continue;                 // (tables that do not add up must not turn into a write outside the contour)
            out = points + 2 * ((size_t)f * max_points + base + off);
            if (h_points) { hout = h_points + 2 * (base + (long long)off); hroom = h_cap - (base + (long long)off); }
        }
        const u32 hp = head_pix[(size_t)f * hcap + k];
        const int pix = (int)(hp & 0x1fffffffu), type = (int)(hp >> 29);
        int y = pix / G.w, x = pix - y * G.w;
        ct_tile T;
        ct_tile_load(G, fb, y - 3, x - 3, T);
        u32 R = ct_ring(T, y, x);
        if (!R) {                                        // a pixel on its own: a border of one point, its own successor
            if (WRITE) {
                out[0] = x; out[1] = y;
                if (hroom > 0) { hout[0] = x; hout[1] = y; }
            } else if (keys) {
                hkey[(size_t)f * hcap + k] = ct_key_w(G, y, x);
                if (mode == 0) hext[(size_t)f * hcap + k] = ct_left_of<SPARSE>(G, fb, hm, hb, y, x);
            } else {
                nd[k] = ((unsigned long long)k << 32) | 1u;
            }
            continue;
        }
        int s = ct_first_cw(R, type == 0 ? 4 : (type == 1 ? 0 : (type == 2 ? 2 : 6)));
        if (keys) {
            const int t0 = __ffs((int)((R | (R << 8)) >> (s + 1))) - 1;
            const bool sw = ((3 - s) & 7) < t0, se = ((7 - s) & 7) < t0;        // the state sweeps the pixel's W / E crack
            hkey[(size_t)f * hcap + k] = sw ? ct_key_w(G, y, x) : (se ? ct_key_e(G, y, x) : CT_NONE);
            // a component's first pixel has nothing above it: only such a W crack can turn out to be the smallest of its cycle
            if (mode == 0) hext[(size_t)f * hcap + k] = (sw && !(R & 0xeu)) ? ct_left_of<SPARSE>(G, fb, hm, hb, y, x) : CT_INSIDE;
            continue;
        }
        u32 cnt = 0, succ = k;
        bool first = true;
        // a border visits a pixel at most once per incoming direction: bound the walk so that a corrupted image cannot
        // keep the wave alive forever
        for (long long guard = 8ll * G.w * G.h + 16; guard > 0; guard--) {
            const u32 q = (R | (R << 8)) >> (s + 1);
            const int t = __ffs((int)q) - 1;
            if (!first) {
                const int ht = ct_head_type<SPARSE>(s, t, x, y, R);
                if (ht >= 0) { succ = ct_head_index(hm, hb, y * G.ww + (x >> 6), x & 63, ht); break; }
            }
            first = false;
            const int s2 = (s + 1 + t) & 7;
            if (s2 != (s ^ 4) || method == 1) {
                if (WRITE) {
                    out[2 * cnt] = x; out[2 * cnt + 1] = y;
                    if ((long long)cnt < hroom) { hout[2 * cnt] = x; hout[2 * cnt + 1] = y; }
                }
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
        if (!WRITE) nd[k] = ((unsigned long long)succ << 32) | cnt;
    }
}

// A single-image call hands its results to the host: the kernels that produce them write them a second time, straight into the
// caller's pinned buffer (posted writes over the link), instead of a copy launched behind them.  All NULL: no mirror.
struct ct_mirror {
    int32_t* info;       // {n_contours, n_points, heads}
    int32_t* counts;
    int32_t* offsets;
    uint8_t* hole;
    int32_t* points;     // the first `cap` points
    long long cap;
};

// per frame: contour lengths from the leaders, exclusive scan -> offsets, total -> out[f].n_points.  `dist`: per head of the frame,
// low 32 bits = a leader's distance around its border = the length of its contour.  All NT threads of the block take part.
template <int NT, typename DT>
__device__ __forceinline__ void ct_offsets_body(int nsel, const u32* __restrict__ starts, const u32* __restrict__ shead,
                                                const DT* dist, int32_t* __restrict__ counts, uint8_t* __restrict__ is_hole_out,
                                                int32_t* __restrict__ offsets, ct_frame_out* __restrict__ out, int max_contours, const ct_mirror& M)
{
    __shared__ u32 wsum[NT / 64];
    __shared__ u32 carry;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K = min(nsel, max_contours);
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < K; base += NT) {
        const int i = base + tid;
        u32 v = 0;
        if (i < K) {
            v = (u32)dist[shead[(size_t)f * max_contours + i]];
            const uint8_t hole = (uint8_t)(starts[(size_t)f * max_contours + i] >> 31);
            counts[(size_t)f * max_contours + i] = (int32_t)v;
            is_hole_out[(size_t)f * max_contours + i] = hole;
            if (M.counts) { M.counts[i] = (int32_t)v; M.hole[i] = hole; }
        }
        u32 inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        u32 woff = 0;
        for (int k = 0; k < wv; k++) woff += wsum[k];
        if (i < K) {
            offsets[(size_t)f * max_contours + i] = (int32_t)(carry + woff + inc - v);
            if (M.offsets) M.offsets[i] = (int32_t)(carry + woff + inc - v);
        }
        __syncthreads();
        if (tid == NT - 1) carry += woff + inc;
        __syncthreads();
    }
    if (tid == 0) {
        out[f].n_contours = nsel; out[f].n_points = (int32_t)carry;
        if (M.info) { M.info[0] = nsel; M.info[1] = (int32_t)carry; }
    }
}

// ---- the cycle bookkeeping between the two follower passes ----
// Steps (1)-(5) of the list above, written once as per-head functions and run in two forms:
//   k_ct_jump      one block of 1024 threads per frame, barriers between the steps, the tables in LDS when the frame's heads fit
//                  (`lds_heads`), otherwise in global memory: ONE launch - what a module's mask or a batch of them needs;
//   k_ctm_*        the same steps as launches over the whole chip, for a single image with very many heads (a speckled mask: one
//                  block would take milliseconds over 600 k heads).  The host cannot know the rounds a frame needs, so every
//                  round that could be needed is launched and a launch whose predecessor changed nothing returns at once.
// Tables in global memory are read and written with relaxed agent-scope accesses (program order, barriers and kernel boundaries
// do the rest).
template <bool L> __device__ __forceinline__ unsigned long long ctj_ld(const unsigned long long* t, u32 k)
{
    if (L) return t[k];
    return __hip_atomic_load(t + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool L> __device__ __forceinline__ void ctj_st(unsigned long long* t, u32 k, unsigned long long v)
{
    if (L) t[k] = v;
    else __hip_atomic_store(t + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u32 ctj_gld(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ctj_gst(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool L> __device__ __forceinline__ u32 ctj_hld(const u32* p) { return L ? *p : ctj_gld(p); }
template <bool L> __device__ __forceinline__ void ctj_hst(u32* p, u32 v) { if (L) *p = v; else ctj_gst(p, v); }

struct ctj_frame {
    u32 H;
    unsigned long long* nd;      // in: (next head, points) of k_ct_seg<false>; out: (leader | CT_TERM, distance to it)
    unsigned long long* tab;     // (next, smallest) pairs of step (1)
    const u32* key;
    u32* ext;                    // in: k_ct_seg<false>'s look to the left; then step (2)'s answers; then the ranks of step (3)
    u32* hr;                     // step (3)'s counts, then: rank of the border a leader starts | CT_UNSEL | CT_NONE (not a leader)
    u32* hrg;                    // the same in global memory for k_ct_seg<true> (which asks for leaders only) when hr is in LDS
    const u32* hp;
    const u32* hb;
    const uint8_t* c8;
};

// (1) leaders.  tab[k] = (J, m): m = the head with the smallest key among the heads from k up to, not including, J along the border.
// A pair read in one 64-bit access is always a true statement of that kind (older or newer), so the rounds work in place and need no
// order among the heads.  A round in which no m changed anywhere proves every m is its cycle's minimum: m(k) <= m(J(k)) for every k,
// the chain k, J(k), J(J(k)), ... closes on itself, its windows tile the whole cycle, so m is constant on it and equal to the minimum.
template <bool L> __device__ __forceinline__ void ctj_lead_init(const ctj_frame& F, u32 k) { ctj_st<L>(F.tab, k, (F.nd[k] & 0xffffffff00000000ull) | k); }
template <bool L> __device__ __forceinline__ bool ctj_lead_step(const ctj_frame& F, u32 k)
{
    const unsigned long long v = ctj_ld<L>(F.tab, k);
    const u32 J = (u32)(v >> 32), m = (u32)v;
    if (J == k) return false;
    const unsigned long long v2 = ctj_ld<L>(F.tab, J);
    const u32 m2 = (u32)v2;
    const bool better = F.key[m2] < F.key[m];
    ctj_st<L>(F.tab, k, (v2 & 0xffffffff00000000ull) | (better ? m2 : m));
    return better;
}
template <bool L> __device__ __forceinline__ bool ctj_leads(const ctj_frame& F, u32 k) { return (u32)ctj_ld<L>(F.tab, k) == k; }

// (2) RETR_EXTERNAL: ext[leader of an outer border] = CT_FRAME (external) | CT_INSIDE | the leader whose answer is also this one's
template <bool L> __device__ __forceinline__ void ctj_ext_init(const ctj_frame& F, u32 k)
{
    if (!ctj_leads<L>(F, k) || (F.key[k] & 1u)) return;
    const u32 e = F.ext[k];                              // k_ct_seg<false>: the head owning the E crack met on the way left
    if (e == CT_FRAME) return;
    const u32 c2 = (u32)ctj_ld<L>(F.tab, e);
    ctj_gst(F.ext + k, (F.key[c2] & 1u) ? CT_INSIDE : c2);
}
template <bool L> __device__ __forceinline__ bool ctj_ext_step(const ctj_frame& F, u32 k)
{
    if (!ctj_leads<L>(F, k) || (F.key[k] & 1u)) return false;
    const u32 e = ctj_gld(F.ext + k);
    if (e >= CT_INSIDE) return false;
    ctj_gst(F.ext + k, ctj_gld(F.ext + e));
    return true;
}
// does head k lead a border the mode returns
template <bool L> __device__ __forceinline__ u32 ctj_selected(const ctj_frame& F, u32 k, int mode)
{
    if (!ctj_leads<L>(F, k)) return 0u;
    return mode == 1 ? 1u : ((!(F.key[k] & 1u) && ctj_gld(F.ext + k) == CT_FRAME) ? 1u : 0u);
}
// (3) with hr[k] = 2 * (returned borders before head k in head order) + (k leads one): head order = scan order of the words; inside a
// word the heads are listed by type, so there the keys decide
template <bool L>
__device__ __forceinline__ void ctj_rank(const ctj_frame& F, u32 k, const ccl_geom& G, u32* __restrict__ starts, u32* __restrict__ shead, int max_contours)
{
    if (!(ctj_hld<L>(F.hr + k) & 1u)) return;
    const u32 pix = F.hp[k] & 0x1fffffffu;
    const int y = (int)(pix / (u32)G.w), x = (int)(pix - (u32)y * (u32)G.w);
    const int wi = y * G.ww + (x >> 6);
    const u32 b = F.hb[wi], e = b + F.c8[wi];
    u32 r = ctj_hld<L>(F.hr + b) >> 1;
    const u32 mine = F.key[k];
    for (u32 q = b; q < e; q++) r += (q != k && (ctj_hld<L>(F.hr + q) & 1u) && F.key[q] < mine) ? 1u : 0u;
    ctj_gst(F.ext + k, r);
    if (r < (u32)max_contours) {
        starts[r] = pix | ((mine & 1u) << 31);
        shead[r] = k;
    }
}
template <bool L> __device__ __forceinline__ void ctj_mark(const ctj_frame& F, u32 k)
{
    const u32 v = ctj_hld<L>(F.hr + k);
    const bool leads = ctj_leads<L>(F, k);
    const u32 mark = (v & 1u) ? ctj_gld(F.ext + k) : (leads ? CT_UNSEL : CT_NONE);
    ctj_hst<L>(F.hr + k, mark);
    if (L && leads) F.hrg[k] = mark;
}
// (4) distances.  t[k] = (J, D): D points lie between this head and head J along the border, until J is the cycle's leader.
template <bool L> __device__ __forceinline__ void ctj_dist_init(const ctj_frame& F, unsigned long long* t, u32 k)
{
    const unsigned long long v = F.nd[k];
    const u32 J = (u32)(v >> 32);
    ctj_st<L>(t, k, ((unsigned long long)(J | (ctj_hld<L>(F.hr + J) != CT_NONE ? CT_TERM : 0u)) << 32) | (u32)v);
}
template <bool L> __device__ __forceinline__ bool ctj_dist_step(unsigned long long* t, u32 k)
{
    const unsigned long long v = ctj_ld<L>(t, k);
    const u32 J = (u32)(v >> 32);
    if (J & CT_TERM) return false;
    const unsigned long long v2 = ctj_ld<L>(t, J);
    ctj_st<L>(t, k, (v2 & 0xffffffff00000000ull) | (u32)((u32)v + (u32)v2));
    return true;
}

template <bool L>
__device__ __forceinline__ void ctj_body(const ccl_geom& G, ctj_frame F, u32* keyl, u32* hrl, const unsigned long long* __restrict__ node, size_t hcap,
                                         u32* __restrict__ starts, u32* __restrict__ shead, int32_t* __restrict__ counts,
                                         uint8_t* __restrict__ is_hole_out, int32_t* __restrict__ offsets, int32_t* __restrict__ points,
                                         ct_frame_out* __restrict__ out, int max_contours, long long max_points, int mode, const ct_mirror& M)
{
    __shared__ u32 s_wsum[16];
    __shared__ u32 s_carry;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 H = F.H;
    const int max_rounds = min(CT_JUMP_ROUNDS, 34 - __clz((int)(H | 1u)));
    if (L) {
        for (u32 k = tid; k < H; k += 1024) keyl[k] = F.key[k];
        F.key = keyl;
        F.hr = hrl;
    }
    for (u32 k = tid; k < H; k += 1024) ctj_lead_init<L>(F, k);
    __syncthreads();
    // (two steps between barriers: the statements stay true in any order, and a barrier costs about what a step does)
    for (int round = 0; round < max_rounds; round++) {
        int changed = 0;
        for (int hop = 0; hop < 2; hop++)
            for (u32 k = tid; k < H; k += 1024) changed |= ctj_lead_step<L>(F, k) ? 1 : 0;
        if (!__syncthreads_or(changed)) break;
    }
    if (mode == 0) {
        for (u32 k = tid; k < H; k += 1024) ctj_ext_init<L>(F, k);
        __syncthreads();
        for (int round = 0; round < CT_JUMP_ROUNDS; round++) {
            int changed = 0;
            for (u32 k = tid; k < H; k += 1024) changed |= ctj_ext_step<L>(F, k) ? 1 : 0;
            if (!__syncthreads_or(changed)) break;
        }
    }
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (u32 base = 0; base < H; base += 1024) {
        const u32 k = base + tid;
        const u32 sel = k < H ? ctj_selected<L>(F, k, mode) : 0u;
        u32 inc = sel;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) s_wsum[wv] = inc;
        __syncthreads();
        u32 woff = 0;
        for (int q = 0; q < wv; q++) woff += s_wsum[q];
        if (k < H) ctj_hst<L>(F.hr + k, ((s_carry + woff + inc - sel) << 1) | sel);
        __syncthreads();
        if (tid == 1023) s_carry += woff + inc;
        __syncthreads();
    }
    const u32 nsel = s_carry;
    for (u32 k = tid; k < H; k += 1024) ctj_rank<L>(F, k, G, starts + (size_t)f * max_contours, shead + (size_t)f * max_contours, max_contours);
    __syncthreads();
    for (u32 k = tid; k < H; k += 1024) ctj_mark<L>(F, k);
    __threadfence_block();
    __syncthreads();
    unsigned long long* t = L ? F.tab : F.nd;          // (in global memory the distances replace the (next, points) pairs in place)
    for (u32 k = tid; k < H; k += 1024) ctj_dist_init<L>(F, t, k);
    __syncthreads();
    for (int round = 0; round < max_rounds; round++) {
        int changed = 0;
        for (int hop = 0; hop < 2; hop++)
            for (u32 k = tid; k < H; k += 1024) changed |= ctj_dist_step<L>(t, k) ? 1 : 0;
        if (!__syncthreads_or(changed)) break;
    }
    if (L)
        for (u32 k = tid; k < H; k += 1024) F.nd[k] = t[k];
    __threadfence_block();
    __syncthreads();               // the block's own stores to node[] are visible to all its threads from here on
    ct_offsets_body<1024>((int)nsel, starts, shead, F.nd, counts, is_hole_out, offsets, out, max_contours, M);
}

struct ctj_args {
    ccl_geom G;
    ct_aux* aux;
    unsigned long long *node, *node2;
    const u32* hkey;
    u32 *hext, *hrank;
    const u32* hbase;
    const uint8_t* cnt8;
    const u32* head_pix;
    size_t hcap;
    u32 *starts, *shead;
    int32_t* counts;
    uint8_t* is_hole;
    int32_t *offsets, *points;
    ct_frame_out* out;
    int max_contours;
    long long max_points;
    int mode;
    u32 lds_heads;
    int defer_big;               // a frame with more heads than the LDS tables hold: report -1 contours instead of working in global memory
    u32* nheads_out;             // nullable: the frame's head count, for the caller's next call (vpk_find_contours `many_heads`)
    ct_mirror mirror;            // single image: the results written a second time, into the caller's pinned buffer
    u32 *flags, *csum;           // k_ctm_* only: per frame CTM_NFLAGS round flags; partial sums of the two scans
    int hops;
};
__device__ __forceinline__ ctj_frame ctj_make_frame(const ctj_args& A, int f, unsigned long long* tab)
{
    ctj_frame F;
    const int nwords = A.G.h * A.G.ww;
    F.H = A.aux[f].nheads;
    F.nd = A.node + (size_t)f * A.hcap;
    F.tab = tab;
    F.key = A.hkey + (size_t)f * A.hcap;
    F.ext = A.hext + (size_t)f * A.hcap;
    F.hr = A.hrank + (size_t)f * A.hcap;
    F.hrg = F.hr;
    F.hp = A.head_pix + (size_t)f * A.hcap;
    F.hb = A.hbase + (size_t)f * nwords;
    F.c8 = A.cnt8 + (size_t)f * nwords;
    return F;
}


// The one-block form with every table in LDS (frames of up to CTJ_LDS_HEADS heads: any mask a module makes).  Written out on its own,
// because what it waits for is not bandwidth: on a nearly idle chip (one image per call) an LDS round trip takes 0.3-0.4 us and a
// barrier with a vote about as long, so the form is built to need few of them:
//   * step (1)'s pairs carry the smallest KEY, not the head that has it (a jump then reads one pair, not a pair and two keys); a
//     head leads iff the smallest key of its border is its own, and the few places that need a leader's index find it among the
//     heads of the key's word;
//   * only its owner writes a head's pair, so the owner keeps it in registers: a jump is one 8-byte read at a random place and one
//     write; each thread takes its (up to 8) heads through a jump together, four jumps between barriers;
//   * the (next, points) pairs stay in registers from the first load to step (4), and in between only the few leaders touch global
//     memory.
// (Steps (1) and (4) as ONE sequence on 16-byte entries (J, D, m, dm, leader) was built and measured: a 16-byte jump costs twice an
// 8-byte one, 12.8 us against 8.1 + 5.6 - not worth entries whose halves another wave might see apart.)
// 16 B of LDS per head: tab (8): the pairs of step (1), then of step (4); a1 (4): keys, then (after the ranks are out) the marks;
// a2 (4): the look to the left / step (2)'s answers, then step (3)'s counts.
#define CTJ_LDS_HEADS 8192
#define CTJ_ITEMS (CTJ_LDS_HEADS / 1024)
#ifndef CTJ_HOPS
#define CTJ_HOPS 4
#endif
template <int ITEMS>
__device__ __forceinline__ void ctj_body_lds(const ctj_args& A, int f, u32 H, unsigned long long* tab, u32* a1, u32* a2)
{
    __shared__ u32 s_wsum[16];
    __shared__ u32 s_carry;
    const ccl_geom& G = A.G;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int mode = A.mode;
    const unsigned long long HI = 0xffffffff00000000ull;
    unsigned long long* nd = A.node + (size_t)f * A.hcap;
    const u32* gkey = A.hkey + (size_t)f * A.hcap;
    const u32* gext = A.hext + (size_t)f * A.hcap;
    u32* ghr = A.hrank + (size_t)f * A.hcap;
    const int nwords = G.h * G.ww;
    const u32* hb = A.hbase + (size_t)f * nwords;
    const uint8_t* c8 = A.cnt8 + (size_t)f * nwords;
    u32* starts = A.starts + (size_t)f * A.max_contours;
    u32* shead = A.shead + (size_t)f * A.max_contours;
    const int max_rounds = min(CT_JUMP_ROUNDS, 34 - __clz((int)(H | 1u)));
#ifdef VP_CT_PROBE
    long long pt[12]; int pn = 0, pr1 = 0, pr2 = 0, pr3 = 0;
#define CT_STAMP() do { __syncthreads(); pt[pn++] = wall_clock64(); } while (0)
#else
#define CT_STAMP() do { } while (0)
#endif
    CT_STAMP();
    unsigned long long ndv[ITEMS], v[ITEMS];
    u32 own[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const u32 k = tid + 1024u * i;
        ndv[i] = 0; own[i] = CT_NONE;
        if (k < H) {
            ndv[i] = nd[k];
            own[i] = gkey[k];
            if (mode == 0) a2[k] = gext[k];
        }
    }
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const u32 k = tid + 1024u * i;
        v[i] = (unsigned long long)k << 32;
        if (k < H) { v[i] = (ndv[i] & HI) | own[i]; tab[k] = v[i]; a1[k] = own[i]; }
    }
    __syncthreads();
    CT_STAMP();
    // (1) leaders: a round in which no smallest key changed anywhere proves every one is its border's (see ctj_lead_step)
    for (int round = 0; round < max_rounds; round++) {
        int changed = 0;
        for (int hop = 0; hop < CTJ_HOPS; hop++) {
            unsigned long long v2[ITEMS];
#pragma unroll
            for (int i = 0; i < ITEMS; i++) { const u32 k = tid + 1024u * i; const u32 J = (u32)(v[i] >> 32); v2[i] = (k < H && J != k) ? tab[J] : v[i]; }
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const u32 k = tid + 1024u * i;
                if (k < H && (u32)(v[i] >> 32) != k) {
                    const bool better = (u32)v2[i] < (u32)v[i];
                    v[i] = (v2[i] & HI) | (better ? (u32)v2[i] : (u32)v[i]);
                    tab[k] = v[i];
                    changed |= better ? 1 : 0;
                }
            }
        }
#ifdef VP_CT_PROBE
        pr1++;
#endif
        if (!__syncthreads_or(changed)) break;
    }
    CT_STAMP();
    // (2) RETR_EXTERNAL: a2[leader of an outer border] = CT_FRAME (external) | CT_INSIDE | the leader whose answer is also this one's
    if (mode == 0) {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const u32 k = tid + 1024u * i;
            if (k >= H || (u32)v[i] != own[i] || (own[i] & 1u)) continue;
            const u32 e = a2[k];
            if (e == CT_FRAME) continue;
            const u32 km = (u32)tab[e];                  // the smallest key of the border that owns the crack met on the way left
            u32 c2 = CT_INSIDE;
            if (!(km & 1u)) {                            // an outer border: its leader is one of the heads of that key's word
                const u32 pos = km >> 1;
                const int y = (int)(pos / (u32)(G.w + 1));
                const int wi = y * G.ww + ((int)(pos - (u32)y * (u32)(G.w + 1)) >> 6);
                const u32 b = hb[wi], e2 = b + c8[wi];
                for (u32 q = b; q < e2; q++) c2 = a1[q] == km ? q : c2;
            }
            a2[k] = c2;
        }
        __syncthreads();
        for (int round = 0; round < CT_JUMP_ROUNDS; round++) {
            int changed = 0;
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const u32 k = tid + 1024u * i;
                if (k >= H || (u32)v[i] != own[i] || (own[i] & 1u)) continue;
                const u32 e = a2[k];
                if (e >= CT_INSIDE) continue;
                a2[k] = a2[e];
                changed = 1;
            }
#ifdef VP_CT_PROBE
            pr2++;
#endif
            if (!__syncthreads_or(changed)) break;
        }
    }
    CT_STAMP();
    // (3) a2[k] = 2 * (returned borders before head k in head order) + (k leads one): every thread takes a run of consecutive heads
    const u32 per = (H + 1023u) / 1024u;
    u32 selbits = 0;
    for (u32 i = 0; i < per; i++) {
        const u32 k = tid * per + i;
        if (k >= H) break;
        const u32 mine = a1[k];
        if ((u32)tab[k] == mine && (mode == 1 || (!(mine & 1u) && a2[k] == CT_FRAME))) selbits |= 1u << i;
    }
    {
        const u32 cnt = (u32)__popc(selbits);
        u32 inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) s_wsum[wv] = inc;
        __syncthreads();
        u32 woff = 0, tot = 0;
        for (int q = 0; q < 16; q++) { woff += q < wv ? s_wsum[q] : 0u; tot += s_wsum[q]; }
        const u32 before = woff + inc - cnt;
        for (u32 i = 0; i < per; i++) {
            const u32 k = tid * per + i;
            if (k >= H) break;
            a2[k] = ((before + (u32)__popc(selbits & ((1u << i) - 1u))) << 1) | ((selbits >> i) & 1u);
        }
        if (tid == 0) s_carry = tot;
        __syncthreads();
    }
    const u32 nsel = s_carry;
    CT_STAMP();
    // head order = scan order of the words; inside a word the heads are listed by type, so there the keys decide.  (a key says where
    // its crack is: no look at the head list)
    u32 rk[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const u32 k = tid + 1024u * i;
        rk[i] = CT_NONE;
        if (k >= H || (u32)v[i] != own[i]) continue;
        rk[i] = CT_UNSEL;
        if (!(a2[k] & 1u)) continue;
        const u32 mine = own[i];
        const u32 pos = mine >> 1;
        const int y = (int)(pos / (u32)(G.w + 1));
        const int x = (int)(pos - (u32)y * (u32)(G.w + 1)) - (int)(mine & 1u);
        const int wi = y * G.ww + (x >> 6);
        const u32 b = hb[wi], e = b + c8[wi];
        u32 r = a2[b] >> 1;
        for (u32 q = b; q < e; q++) r += (q != k && (a2[q] & 1u) && a1[q] < mine) ? 1u : 0u;
        rk[i] = r;
        if (r < (u32)A.max_contours) {
            starts[r] = (u32)(y * G.w + x) | ((mine & 1u) << 31);
            shead[r] = k;
        }
    }
    __syncthreads();
    CT_STAMP();
    // marks (k_ct_seg<true> asks for the marks of leaders only)
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const u32 k = tid + 1024u * i;
        if (k >= H) continue;
        a1[k] = rk[i];
        if (rk[i] != CT_NONE) ghr[k] = rk[i];
    }
    __syncthreads();
    CT_STAMP();
    // (4) distances
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const u32 k = tid + 1024u * i;
        v[i] = (unsigned long long)CT_TERM << 32;
        if (k < H) {
            const u32 J = (u32)(ndv[i] >> 32);
            v[i] = ((unsigned long long)(J | (a1[J] != CT_NONE ? CT_TERM : 0u)) << 32) | (u32)ndv[i];
            tab[k] = v[i];
        }
    }
    __syncthreads();
    for (int round = 0; round < max_rounds; round++) {
        int waiting = 0;                                  // some head of this thread has not reached its leader yet
        for (int hop = 0; hop < CTJ_HOPS; hop++) {
            unsigned long long v2[ITEMS];
#pragma unroll
            for (int i = 0; i < ITEMS; i++) { const u32 J = (u32)(v[i] >> 32); v2[i] = (J & CT_TERM) ? 0ull : tab[J]; }
            waiting = 0;
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const u32 k = tid + 1024u * i;
                if (!((u32)(v[i] >> 32) & CT_TERM)) {
                    v[i] = (v2[i] & HI) | (u32)((u32)v[i] + (u32)v2[i]);
                    tab[k] = v[i];
                    waiting |= ((u32)(v[i] >> 32) & CT_TERM) ? 0 : 1;
                }
            }
        }
#ifdef VP_CT_PROBE
        pr3++;
#endif
        if (!__syncthreads_or(waiting)) break;
    }
    CT_STAMP();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const u32 k = tid + 1024u * i;
        if (k < H) nd[k] = v[i];
    }
    ct_offsets_body<1024>((int)nsel, A.starts, A.shead, tab, A.counts, A.is_hole, A.offsets, A.out, A.max_contours, A.mirror);
#ifdef VP_CT_PROBE
    CT_STAMP();
    if (tid == 0 && f == 0) {
        printf("jump H=%u nsel=%u rounds %d %d %d | x10ns:", H, nsel, pr1, pr2, pr3);
        for (int i = 1; i < pn; i++) printf(" %lld", pt[i] - pt[i - 1]);
        printf(" | total %lld\n", pt[pn - 1] - pt[0]);
    }
#endif
}

extern __shared__ __attribute__((aligned(16))) unsigned long long ctj_dyn[];
__global__ __launch_bounds__(1024) void k_ct_jump(ctj_args A)
{
    const int f = blockIdx.x;
    const u32 H = A.aux[f].nheads;
    if (A.nheads_out && threadIdx.x == 0) A.nheads_out[f] = H;
    if (A.mirror.info && threadIdx.x == 0) A.mirror.info[2] = (int32_t)H;
    if (H > (u32)CTJ_LDS_HEADS && A.defer_big) {          // the caller repeats the pass in the launches form (k_ct_seg<true> finds no leader marks: writes nothing)
        if (threadIdx.x == 0) {
            A.out[f].n_contours = -1; A.out[f].n_points = 0;
            if (A.mirror.info) { A.mirror.info[0] = -1; A.mirror.info[1] = 0; }
        }
        return;
    }
    if (H <= (u32)CTJ_LDS_HEADS)
    {
        // (every thread takes ceil(H / 1024) heads through a jump together; the loops over them are unrolled, and what a small frame does
        // not need would still be issued: an instantiation per count - a module's mask has 1,400-2,800 heads)
        unsigned long long* tab = ctj_dyn;
        u32* a1 = reinterpret_cast<u32*>(ctj_dyn + CTJ_LDS_HEADS);
        u32* a2 = a1 + CTJ_LDS_HEADS;
        if (H <= 1024u) ctj_body_lds<1>(A, f, H, tab, a1, a2);
        else if (H <= 2048u) ctj_body_lds<2>(A, f, H, tab, a1, a2);
        else if (H <= 3072u) ctj_body_lds<3>(A, f, H, tab, a1, a2);
        else if (H <= 4096u) ctj_body_lds<4>(A, f, H, tab, a1, a2);
        else if (H <= 6144u) ctj_body_lds<6>(A, f, H, tab, a1, a2);
        else ctj_body_lds<CTJ_ITEMS>(A, f, H, tab, a1, a2);
    }
    else
        ctj_body<false>(A.G, ctj_make_frame(A, f, A.node2 + (size_t)f * A.hcap), nullptr, nullptr, A.node, A.hcap, A.starts, A.shead, A.counts, A.is_hole,
                        A.offsets, A.points, A.out, A.max_contours, A.max_points, A.mode, A.mirror);
}

// ---- the same steps as launches over the chip (grid (blocks, n) x 256) ----
// flags[f][i]: launch i of a round sequence changed something.  Slot 0 of each sequence is set by the launch before it.
#define CTM_SEQ 32
#define CTM_NFLAGS (3 * CTM_SEQ)
enum { CTM_LEAD_INIT, CTM_LEAD, CTM_EXT_INIT, CTM_EXT, CTM_RANK, CTM_MARK, CTM_DIST };

template <int PH>
__global__ __launch_bounds__(256) void k_ctm(ctj_args A, int slot)
{
    const int f = blockIdx.y;
    u32* fl = A.flags + (size_t)f * CTM_NFLAGS;
    constexpr bool is_round = PH == CTM_LEAD || PH == CTM_EXT || PH == CTM_DIST;
    if (is_round && !ctj_gld(fl + slot - 1)) return;       // the launch before this one changed nothing: the sequence is through
    const ctj_frame F = ctj_make_frame(A, f, A.node2 + (size_t)f * A.hcap);
    if (PH == CTM_LEAD_INIT && blockIdx.x == 0) {
        for (int i = threadIdx.x; i < CTM_NFLAGS; i += 256) ctj_gst(fl + i, (i % CTM_SEQ) == 0 ? 1u : 0u);   // every sequence runs its first round
        if (A.nheads_out && threadIdx.x == 0) A.nheads_out[f] = F.H;
        if (A.mirror.info && threadIdx.x == 0) A.mirror.info[2] = (int32_t)F.H;
    }
    int changed = 0;
    for (u32 k = blockIdx.x * 256 + threadIdx.x; k < F.H; k += gridDim.x * 256) {
        if (PH == CTM_LEAD_INIT) ctj_lead_init<false>(F, k);
        if (PH == CTM_LEAD)
            for (int h = 0; h < A.hops; h++) changed |= ctj_lead_step<false>(F, k) ? 1 : 0;
        if (PH == CTM_EXT_INIT) ctj_ext_init<false>(F, k);
        if (PH == CTM_EXT)
            for (int h = 0; h < A.hops; h++) changed |= ctj_ext_step<false>(F, k) ? 1 : 0;
        if (PH == CTM_RANK) ctj_rank<false>(F, k, A.G, A.starts + (size_t)f * A.max_contours, A.shead + (size_t)f * A.max_contours, A.max_contours);
        if (PH == CTM_MARK) ctj_mark<false>(F, k);
        if (PH == CTM_DIST)
            for (int h = 0; h < A.hops; h++) changed |= ctj_dist_step<false>(F.nd, k) ? 1 : 0;
    }
    if (is_round && changed) ctj_gst(fl + slot, 1u);
}
// (the distances' start is a launch of its own, after the marks: it needs the mark of every head's successor)
__global__ __launch_bounds__(256) void k_ctm_dist_init(ctj_args A)
{
    const int f = blockIdx.y;
    const ctj_frame F = ctj_make_frame(A, f, A.node2 + (size_t)f * A.hcap);
    for (u32 k = blockIdx.x * 256 + threadIdx.x; k < F.H; k += gridDim.x * 256) ctj_dist_init<false>(F, F.nd, k);
}

// the two scans (returned borders before each head; points before each contour), two launches each: sums of chunks of 1024 items,
// then every chunk adds up the sums before it and scans itself.  WHAT 0: heads -> hr[k]; 1: contours -> offsets, counts, is_hole
template <int WHAT>
__device__ __forceinline__ u32 ctm_item(const ctj_args& A, const ctj_frame& F, int f, u32 i, u32 n_items)
{
    if (i >= n_items) return 0u;
    if (WHAT == 0) return ctj_selected<false>(F, i, A.mode);
    const u32 sh = A.shead[(size_t)f * A.max_contours + i];
    return (u32)ctj_ld<false>(F.nd, sh);                  // the leader's distance around its cycle = the contour's length
}
template <int WHAT>
__global__ __launch_bounds__(1024) void k_ctm_sums(ctj_args A)
{
    __shared__ u32 s_w[16];
    const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const ctj_frame F = ctj_make_frame(A, f, A.node2 + (size_t)f * A.hcap);
    const u32 n_items = WHAT == 0 ? F.H : min(A.aux[f].nsel, (u32)A.max_contours);
    const u32 nchunks = (n_items + 1023u) / 1024u;
    u32* cs = A.csum + (size_t)f * (A.hcap / 1024 + 2);
    for (u32 c = blockIdx.x; c < nchunks; c += gridDim.x) {
        u32 v = ctm_item<WHAT>(A, F, f, c * 1024u + tid, n_items);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        if (lane == 0) s_w[wv] = v;
        __syncthreads();
        if (tid == 0) { u32 t = 0; for (int q = 0; q < 16; q++) t += s_w[q]; cs[c] = t; }
        __syncthreads();
    }
}
template <int WHAT>
__global__ __launch_bounds__(1024) void k_ctm_scan(ctj_args A)
{
    __shared__ u32 s_w[16], s_t[16];
    const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const ctj_frame F = ctj_make_frame(A, f, A.node2 + (size_t)f * A.hcap);
    const u32 nsel = WHAT == 0 ? 0u : A.aux[f].nsel;
    const u32 n_items = WHAT == 0 ? F.H : min(nsel, (u32)A.max_contours);
    const u32 nchunks = (n_items + 1023u) / 1024u;
    const u32* cs = A.csum + (size_t)f * (A.hcap / 1024 + 2);
    if (nchunks == 0 && blockIdx.x == 0 && tid == 0) {
        if (WHAT == 0) A.aux[f].nsel = 0u;
        else {
            A.out[f].n_contours = (int32_t)nsel; A.out[f].n_points = 0;
            if (A.mirror.info) { A.mirror.info[0] = (int32_t)nsel; A.mirror.info[1] = 0; }
        }
    }
    for (u32 c = blockIdx.x; c < nchunks; c += gridDim.x) {
        u32 pre = 0;
        for (u32 q = tid; q < c; q += 1024) pre += cs[q];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) pre += __shfl_xor(pre, d);
        if (lane == 0) s_t[wv] = pre;
        const u32 i = c * 1024u + tid;
        const u32 v = ctm_item<WHAT>(A, F, f, i, n_items);
        u32 inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        u32 carry = 0, woff = 0;
        for (int q = 0; q < 16; q++) carry += s_t[q];
        for (int q = 0; q < wv; q++) woff += s_w[q];
        const u32 excl = carry + woff + inc - v;
        if (i < n_items) {
            if (WHAT == 0) ctj_gst(F.hr + i, (excl << 1) | v);
            else {
                const size_t o = (size_t)f * A.max_contours + i;
                A.counts[o] = (int32_t)v;
                A.is_hole[o] = (uint8_t)(A.starts[o] >> 31);
                A.offsets[o] = (int32_t)excl;
                if (A.mirror.counts) { A.mirror.counts[i] = (int32_t)v; A.mirror.hole[i] = (uint8_t)(A.starts[o] >> 31); A.mirror.offsets[i] = (int32_t)excl; }
            }
        }
        if (c == nchunks - 1 && tid == 1023) {
            if (WHAT == 0) A.aux[f].nsel = excl + v;
            else {
                A.out[f].n_contours = (int32_t)nsel; A.out[f].n_points = (int32_t)(excl + v);
                if (A.mirror.info) { A.mirror.info[0] = (int32_t)nsel; A.mirror.info[1] = (int32_t)(excl + v); }
            }
        }
        __syncthreads();
    }
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
    const size_t words = (size_t)n * h * vp_ww(w);
    const size_t hcap = ct_hcap(w, h) * n;
    return vp_align(words * 32) + vp_align(words * 4) + vp_align(words) + 4 * vp_align(hcap * 4) + 2 * vp_align(hcap * 8) +
           2 * vp_align((size_t)n * max_contours * 4) + vp_align(sizeof(ct_aux) * n) + vp_align(words / 64 + 4 * n) +
           vp_align((size_t)n * CTM_NFLAGS * 4) + vp_align((ct_hcap(w, h) / 1024 + 2) * 4 * n) + 8192;
}


// Head counts of the frames of the last batched pass (up to 64 of them), written by its prefix kernel into a pinned array of the
// context's: read WITHOUT synchronising by the next call that has to choose a form of the bookkeeping (vp_ct_batch_hint) - possibly
// a call late, possibly of another mask: a guess, which the results do not depend on.
static u32* vp_ct_hint_slots(vp_ctx* ctx, int n)
{
    if (!ctx->ct_hint_host) {
        void* p = nullptr;
        if (hipHostMalloc(&p, 64 * sizeof(u32), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        memset(p, 0, 64 * sizeof(u32));
        ctx->ct_hint_host = (uint32_t*)p;
    }
    ctx->ct_hint_n = n < 64 ? n : 64;
    return ctx->ct_hint_host;
}
uint32_t vp_ct_batch_hint(vp_ctx* ctx)
{
    uint32_t m = 0;
    if (ctx->ct_hint_host)
        for (int i = 0; i < ctx->ct_hint_n; i++) { const uint32_t v = *(volatile uint32_t*)(ctx->ct_hint_host + i); m = v > m ? v : m; }
    return m;
}

// d_counts / d_is_hole / d_offsets: [n][max_contours]; d_points: [n][max_points][2]; d_info: [n] {n_contours, n_points}.  Contours are
// stored in discovery order (raster order of the start pixel); cv2 returns them reversed - the caller reverses.
// Five launches on the context's stream.  `many_heads` (one image whose last pass counted very many heads - the caller's guess, the
// results do not depend on it): the bookkeeping between the follower passes as launches over the chip instead of one block.
// d_nheads_out (nullable, [n]): the frames' head counts.
int vpk_find_contours(vp_ctx* ctx, const u64* d_bits, int w, int h, int n, int mode, int method, int32_t* d_counts, uint8_t* d_is_hole,
                      int32_t* d_offsets, int32_t* d_points, int max_contours, long long max_points, int32_t* d_info, bool many_heads,
                      uint32_t* d_nheads_out, const vp_contour_mirror* host, bool defer_big)
{
    if (mode != 0 && mode != 1) return vp_fail(ctx, VP_ERR_INVALID, "contour mode");
    if (method != 1 && method != 2) return vp_fail(ctx, VP_ERR_INVALID, "contour approximation");
    if ((size_t)w * h >= (1u << 29)) return vp_fail(ctx, VP_ERR_INVALID, "contours: image too large");
    ctj_args A;
    ccl_make_geom(A.G, w, h, VP_CCL_PIXEL, 0, 0);
    const ccl_geom& G = A.G;
    const int nwords = h * G.ww;
    const size_t words = (size_t)n * nwords;
    const size_t hcap = ct_hcap(w, h);
    u64* hmaps = (u64*)vp_ws_take(ctx, words * 32);
    u32* hbase = (u32*)vp_ws_take(ctx, words * 4);
    uint8_t* cnt8 = (uint8_t*)vp_ws_take(ctx, words);
    u32* head_pix = (u32*)vp_ws_take(ctx, hcap * n * 4);
    A.hrank = (u32*)vp_ws_take(ctx, hcap * n * 4);
    u32* hkey = (u32*)vp_ws_take(ctx, hcap * n * 4);
    A.hext = (u32*)vp_ws_take(ctx, hcap * n * 4);
    A.node = (unsigned long long*)vp_ws_take(ctx, hcap * n * 8);
    A.node2 = (unsigned long long*)vp_ws_take(ctx, hcap * n * 8);
    A.starts = (u32*)vp_ws_take(ctx, (size_t)n * max_contours * 4);
    A.shead = (u32*)vp_ws_take(ctx, (size_t)n * max_contours * 4);
    A.aux = (ct_aux*)vp_ws_take(ctx, sizeof(ct_aux) * n);
    const size_t nparts = (size_t)(nwords + 255) / 256;
    u32* partsum = (u32*)vp_ws_take(ctx, nparts * n * 4);
    A.flags = (u32*)vp_ws_take(ctx, (size_t)n * CTM_NFLAGS * 4);
    A.csum = (u32*)vp_ws_take(ctx, (hcap / 1024 + 2) * 4 * n);
    if (!hmaps || !hbase || !cnt8 || !head_pix || !A.hrank || !hkey || !A.hext || !A.node || !A.node2 || !A.starts || !A.shead || !A.aux || !partsum ||
        !A.flags || !A.csum)
        return vp_fail(ctx, VP_ERR_NOMEM, "contour workspace");
    const size_t jump_lds = (size_t)CTJ_LDS_HEADS * 16;
    if (!ctx->ct_lds_set) {                               // (per context = per device: the attribute belongs to the device's copy of the kernel)
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ct_jump), hipFuncAttributeMaxDynamicSharedMemorySize, (int)jump_lds));
        ctx->ct_lds_set = 1;
    }
    A.hkey = hkey; A.hbase = hbase; A.cnt8 = cnt8; A.head_pix = head_pix; A.hcap = hcap;
    A.counts = d_counts; A.is_hole = d_is_hole; A.offsets = d_offsets; A.points = d_points;
    A.out = reinterpret_cast<ct_frame_out*>(d_info);
    A.max_contours = max_contours; A.max_points = max_points; A.mode = mode; A.lds_heads = CTJ_LDS_HEADS; A.nheads_out = d_nheads_out;
    A.hops = 3;
    A.defer_big = (defer_big && n == 1 && !many_heads) ? 1 : 0;
    A.mirror = ct_mirror{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (host && n == 1) A.mirror = ct_mirror{host->info, host->counts, host->offsets, host->is_hole, host->points, host->points_cap};
    hipStream_t s = ctx->stream;
    vp_prof_scope ps(ctx, VPK_OTHER);
    const dim3 wgrid((unsigned)((nwords + 255) / 256), (unsigned)n);
    // heads per frame are not known on the host: a fixed number of blocks per frame walks the head list (a real mask has a few
    // thousand heads; empty blocks of a grid sized for the worst case would cost more than the work)
    // (one image: the head count of the context's last single-image pass sizes the grid - two items per head, half as much again -
    // instead of a thousand blocks of which thirty find work; the loops stride over the grid, so a wrong guess only costs time)
    size_t hblocks = (size_t)std::max(32, std::min(1024, 8192 / n));
    if (n == 1 && ctx->ct_heads_hint) hblocks = std::min<size_t>(hblocks, std::max<size_t>(32, ((size_t)ctx->ct_heads_hint * 3 + 255) / 256));
    const dim3 hgrid((unsigned)std::min<size_t>((hcap + 255) / 256, hblocks), (unsigned)n);
    if (many_heads) hipLaunchKernelGGL(k_ct_headmaps<true>, wgrid, dim3(256), 0, s, d_bits, G, hmaps, partsum, cnt8);
    else hipLaunchKernelGGL(k_ct_headmaps<false>, wgrid, dim3(256), 0, s, d_bits, G, hmaps, partsum, cnt8);
    hipLaunchKernelGGL(k_ct_prefix, wgrid, dim3(256), 0, s, hmaps, 4, nwords, partsum, hbase, &A.aux->nheads, 2, w, G.ww, head_pix, hcap, cnt8,
                       (n > 1 || !host) ? vp_ct_hint_slots(ctx, n) : nullptr);
#define CT_SEG_ARGS d_bits, G, hmaps, hbase, head_pix, A.hrank, hcap, A.aux, A.node, hkey, A.hext, mode, method, d_offsets, d_points, max_contours, max_points
    const u32 skip_above = A.defer_big ? (u32)CTJ_LDS_HEADS : 0u;
    if (many_heads) hipLaunchKernelGGL((k_ct_seg<false, true>), hgrid, dim3(256), 0, s, CT_SEG_ARGS, (int32_t*)nullptr, 0ll, skip_above);
    else hipLaunchKernelGGL((k_ct_seg<false, false>), hgrid, dim3(256), 0, s, CT_SEG_ARGS, (int32_t*)nullptr, 0ll, skip_above);
    if (!many_heads) {
        hipLaunchKernelGGL(k_ct_jump, dim3((unsigned)n), dim3(1024), jump_lds, s, A);
    } else {
        // a launch of `hops` steps multiplies the shortest window (distance jumped) by hops + 1 at least: rounds that always suffice
        int rounds = 2;
        for (size_t reach = 1; reach < hcap; reach *= (size_t)(A.hops + 1)) rounds++;
        if (rounds >= CTM_SEQ) return vp_fail(ctx, VP_ERR_INVALID, "contours: image too large");
        const dim3 mg((unsigned)std::max(32, std::min(1024, 8192 / n)), (unsigned)n), sg((unsigned)std::max(8, std::min(256, 2048 / n)), (unsigned)n);
        hipLaunchKernelGGL((k_ctm<CTM_LEAD_INIT>), mg, dim3(256), 0, s, A, 0);
        for (int i = 1; i <= rounds; i++) hipLaunchKernelGGL((k_ctm<CTM_LEAD>), mg, dim3(256), 0, s, A, i);
        if (mode == 0) {
            hipLaunchKernelGGL((k_ctm<CTM_EXT_INIT>), mg, dim3(256), 0, s, A, 0);
            for (int i = 1; i <= rounds; i++) hipLaunchKernelGGL((k_ctm<CTM_EXT>), mg, dim3(256), 0, s, A, CTM_SEQ + i);
        }
        hipLaunchKernelGGL((k_ctm_sums<0>), sg, dim3(1024), 0, s, A);
        hipLaunchKernelGGL((k_ctm_scan<0>), sg, dim3(1024), 0, s, A);
        hipLaunchKernelGGL((k_ctm<CTM_RANK>), mg, dim3(256), 0, s, A, 0);
        hipLaunchKernelGGL((k_ctm<CTM_MARK>), mg, dim3(256), 0, s, A, 0);
        hipLaunchKernelGGL(k_ctm_dist_init, mg, dim3(256), 0, s, A);
        for (int i = 1; i <= rounds; i++) hipLaunchKernelGGL((k_ctm<CTM_DIST>), mg, dim3(256), 0, s, A, 2 * CTM_SEQ + i);
        hipLaunchKernelGGL((k_ctm_sums<1>), sg, dim3(1024), 0, s, A);
        hipLaunchKernelGGL((k_ctm_scan<1>), sg, dim3(1024), 0, s, A);
    }
    if (many_heads) hipLaunchKernelGGL((k_ct_seg<true, true>), hgrid, dim3(256), 0, s, CT_SEG_ARGS, A.mirror.points, A.mirror.cap, skip_above);
    else hipLaunchKernelGGL((k_ct_seg<true, false>), hgrid, dim3(256), 0, s, CT_SEG_ARGS, A.mirror.points, A.mirror.cap, skip_above);
#undef CT_SEG_ARGS
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}
