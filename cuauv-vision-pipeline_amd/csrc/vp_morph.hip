// Morphology kernels for gfx950.
//
// Replaces (reference, via cv2): utils/transform.py:80-164 erode / dilate / morph_remove_noise /
// morph_close_holes / morph_borders, modules/preprocessor.py:120-129.
//
// Binary masks with all-ones (rect) kernels — the red_buoy / bins chains — run on bit-packed
// images: 64 px per u64, a row of 1920 px is 30 words, a whole 1080p mask is 259 KB.  A
// workgroup stages a strip of rows plus the halo of *all* chained stages in LDS and runs every
// erode/dilate stage there (shift/AND/OR on words, one barrier between the horizontal and the
// vertical half of a stage), so OPEN followed by CLOSE (4 stencils) costs one read of the bit
// image and one write of the 0/255 mask: 1.25 B/px.  Border rule = cv2's default for morphology:
// samples outside the image never win (treated as 1 for erode, 0 for dilate) at every stage.
//
// Everything else (grey-level images, ellipse/cross kernels, multi-channel) goes through the
// generic kernel: brute-force min/max over the structuring element's offsets.
//
// Measured and dropped (git history has them): a row-sweep variant (lane = word column, rings in LDS, neighbours by
// shuffle: 237 us vs 60 us for 64 x 1080p — one wave per SIMD running serial LDS/shuffle chains) and fusing the
// strip-local labelling into this kernel (140.6 us vs 81.3 + 58.1 us: both parts are bound by per-block latency).
#include "vp_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

#define MB_THREADS 512
#define MB_STRIP 32


// 64-bit funnel shifts built from v_alignbit_b32 (full rate) instead of 64-bit variable shifts (quarter rate):
// fsr(next, cur, d) = (cur >> d) | (next << (64 - d)),  fsl(cur, prev, d) = (cur << d) | (prev >> (64 - d)),  1 <= d <= 31
__device__ __forceinline__ u64 fsr(u64 next, u64 cur, int d)
{
    const u32 lo = __builtin_amdgcn_alignbit((u32)(cur >> 32), (u32)cur, (u32)d);
    const u32 hi = __builtin_amdgcn_alignbit((u32)next, (u32)(cur >> 32), (u32)d);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 fsl(u64 cur, u64 prev, int d)
{
    const u32 hi = __builtin_amdgcn_alignbit((u32)(cur >> 32), (u32)cur, (u32)(32 - d));
    const u32 lo = __builtin_amdgcn_alignbit((u32)cur, (u32)(prev >> 32), (u32)(32 - d));
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u32 expand4m(u32 nib) { return (((nib & 0xfu) * 0x00204081u) & 0x01010101u) * 0xffu; }

// ---- pack / unpack ---------------------------------------------------------------------------

// grid.x = n*h rows; thread = 16-px group (4 per word)
__global__ __launch_bounds__(256) void k_pack_bits(const uint8_t* __restrict__ src, size_t stride, int w, int ww,
                                                   u64* __restrict__ bits, int* __restrict__ flags)
{
    const size_t row = blockIdx.x;
    const int grp = blockIdx.y * 256 + threadIdx.x;
    const bool live = grp < ww * 4;
    const uint8_t* p = src + row * stride;
    u32 m = 0;
    bool odd = false;
    if (live) {
        const int x0 = grp * 16;
        if (x0 + 16 <= w && (((uintptr_t)(p + x0)) & 15) == 0) {
            const uint4 v = *reinterpret_cast<const uint4*>(p + x0);
            const u32 in[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const u32 b = (in[k >> 2] >> (8 * (k & 3))) & 0xff;
                m |= (u32)(b != 0) << k;
                odd |= (b != 0) & (b != 255);
            }
        } else {
            for (int k = 0; k < 16; k++) {
                const int x = x0 + k;
                if (x < w) {
                    const u32 b = p[x];
                    m |= (u32)(b != 0) << k;
                    odd |= (b != 0) & (b != 255);
                }
            }
        }
    }
    u64 wv = (u64)m << (16 * (threadIdx.x & 3));
    wv |= __shfl_xor(wv, 1);
    wv |= __shfl_xor(wv, 2);
    if (live && (threadIdx.x & 3) == 0) bits[row * (size_t)ww + (grp >> 2)] = wv;
    if (flags && __any(odd)) {
        if ((threadIdx.x & 63) == 0) atomicOr(flags, 1);
    }
}

__device__ __forceinline__ void store_mask16(uint8_t* __restrict__ dstrow, int x0, int w, u32 m16, bool vec_ok)
{
    if (vec_ok && x0 + 16 <= w) {
        vp_store16(dstrow + x0, expand4m(m16), expand4m(m16 >> 4), expand4m(m16 >> 8), expand4m(m16 >> 12));
    } else {
        for (int k = 0; k < 16; k++)
            if (x0 + k < w) dstrow[x0 + k] = ((m16 >> k) & 1) ? 255 : 0;
    }
}

__global__ __launch_bounds__(256) void k_unpack_bits(const u64* __restrict__ bits, int w, int ww, uint8_t* __restrict__ dst)
{
    const size_t row = blockIdx.x;
    const int grp = blockIdx.y * 256 + threadIdx.x;
    if (grp >= ww * 4) return;
    const u64 wv = bits[row * (size_t)ww + (grp >> 2)];
    uint8_t* drow = dst + row * (size_t)w;
    store_mask16(drow, grp * 16, w, (u32)(wv >> (16 * (grp & 3))) & 0xffffu, (((uintptr_t)drow) & 15) == 0);
}

// the batch as one run of 16-pixel groups (rows follow each other without a gap when w % 16 == 0): four groups per thread, a wave
// stores 1 KB per instruction and no lane idles at the end of a row
__global__ __launch_bounds__(256) void k_unpack_bits_flat(const u64* __restrict__ bits, u32 gpr, int ww, u32 ngroups, uint8_t* __restrict__ dst)
{
    for (u32 gb = blockIdx.x * 1024u; gb < ngroups; gb += gridDim.x * 1024u) {
        const u32 g0 = gb + threadIdx.x;
        u32 m[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 g = g0 + 256u * k;
            m[k] = 0;
            if (g < ngroups) {
                const u32 row = g / gpr, c = g - row * gpr;
                m[k] = (u32)(bits[(size_t)row * ww + (c >> 2)] >> (16 * (c & 3))) & 0xffffu;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 g = g0 + 256u * k;
            if (g < ngroups) vp_store16(dst + (size_t)g * 16, expand4m(m[k]), expand4m(m[k] >> 4), expand4m(m[k] >> 8), expand4m(m[k] >> 12));
        }
    }
}

int vpk_pack_bits(vp_ctx* ctx, const uint8_t* d_src, size_t stride, int w, int h, int n, u64* d_bits, int* d_flags)
{
    const int ww = vp_ww(w);
    dim3 grid((unsigned)((size_t)n * h), (unsigned)((ww * 4 + 255) / 256));
    hipLaunchKernelGGL(k_pack_bits, grid, dim3(256), 0, ctx->stream, d_src, stride, w, ww, d_bits, d_flags);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

int vpk_unpack_bits(vp_ctx* ctx, const u64* d_bits, int w, int h, int n, uint8_t* d_dst)
{
    const int ww = vp_ww(w);
    const size_t ngroups = (size_t)n * h * (size_t)(w / 16);
    if (w % 16 == 0 && (((uintptr_t)d_dst) & 15) == 0 && ngroups > 0 && ngroups < ((size_t)1 << 32) - 2048) {
        hipLaunchKernelGGL(k_unpack_bits_flat, dim3((unsigned)((ngroups + 1023) / 1024)), dim3(256), 0, ctx->stream, d_bits, (u32)(w / 16), ww, (u32)ngroups, d_dst);
        VP_HIP(ctx, hipGetLastError());
        return VP_OK;
    }
    dim3 grid((unsigned)((size_t)n * h), (unsigned)((ww * 4 + 255) / 256));
    hipLaunchKernelGGL(k_unpack_bits, grid, dim3(256), 0, ctx->stream, d_bits, w, ww, d_dst);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// ---- bit-plane morphology, LDS strips -----------------------------------------------------------

struct mb_params {
    int w, h, ww;
    int halo_top, halo_bot;   // sum of vertical extents over all stages
    int rows;                 // MB_STRIP + halo_top + halo_bot (LDS rows)
    int strips;               // strips per frame
    vp_bitplan plan;
};

// iterate the staged words with a division-free (row, column) mapping: 8 rows x 32 columns per pass of the block
#define MB_FOR_WORDS(r, j, i)                                   \
    for (int r = threadIdx.x >> 5; r < rows; r += MB_THREADS / 32) \
        for (int j = threadIdx.x & 31, i = r * ww + j; j < ww; j += 32, i += 32)

// dynamic LDS: two buffers of rows*ww u64
__global__ __launch_bounds__(MB_THREADS) void k_morph_bits(const u64* __restrict__ in, mb_params P, u64* __restrict__ out_bits,
                                                           uint8_t* __restrict__ out_mask)
{
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int ww = P.ww, rows = P.rows;
    u64* A = lds;
    u64* B = lds + (size_t)rows * ww;
    const int frame = blockIdx.x / P.strips;
    const int strip = blockIdx.x - frame * P.strips;
    const int y0 = strip * MB_STRIP;          // first output row of this strip
    const int ybase = y0 - P.halo_top;        // image row of LDS row 0
    const u64* fin = in + (size_t)frame * P.h * ww;
    const u64 lastmask = (P.w & 63) ? ((1ull << (P.w & 63)) - 1ull) : ~0ull;  // valid bits of word ww-1

    MB_FOR_WORDS(r, j, i) {
        const int y = ybase + r;
        A[i] = (y >= 0 && y < P.h) ? fin[(size_t)y * ww + j] : 0ull;
    }
    __syncthreads();

    for (int si = 0; si < P.plan.n; si++) {
        const vp_bitstage st = P.plan.s[si];
        const u64 neutral = st.dilate ? 0ull : ~0ull;
        // horizontal half: A -> B
        MB_FOR_WORDS(r, j, i) {
            const int y = ybase + r;
            if (y < 0 || y >= P.h) continue;
            u64 cur = A[i];
            u64 prev = j > 0 ? A[i - 1] : neutral;
            u64 next = j + 1 < ww ? A[i + 1] : neutral;
            if (!st.dilate) {  // out-of-image columns inside the last word count as 1 for erosion
                if (j == ww - 1) cur |= ~lastmask;
                if (j + 1 == ww - 1) next |= ~lastmask;
            }
            u64 acc = cur;
            if (st.dilate) {
                for (int d = 1; d <= st.r; d++) acc |= fsr(next, cur, d);
                for (int d = 1; d <= st.l; d++) acc |= fsl(cur, prev, d);
            } else {
                for (int d = 1; d <= st.r; d++) acc &= fsr(next, cur, d);
                for (int d = 1; d <= st.l; d++) acc &= fsl(cur, prev, d);
            }
            if (j == ww - 1) acc &= lastmask;
            B[i] = acc;
        }
        __syncthreads();
        // vertical half: B -> A.  Rows outside the image never win; rows outside the staged range only feed halo
        // rows that are no longer needed.
        MB_FOR_WORDS(r, j, i) {
            const int y = ybase + r;
            if (y < 0 || y >= P.h) continue;
            int lo = max(max(r - st.u, 0), -ybase);
            int hi = min(min(r + st.d, rows - 1), P.h - 1 - ybase);
            u64 acc = B[i];
            const u64* col = B + j;
            if (st.dilate) {
#pragma unroll 4
                for (int rr = lo; rr <= hi; rr++) acc |= col[rr * ww];
            } else {
#pragma unroll 4
                for (int rr = lo; rr <= hi; rr++) acc &= col[rr * ww];
            }
            A[i] = acc;
        }
        __syncthreads();
    }

    // epilogue: rows [halo_top, halo_top + MB_STRIP) of A are final
    const int nout_rows = min(MB_STRIP, P.h - y0);
    if (out_bits) {
        u64* fo = out_bits + (size_t)frame * P.h * ww;
        for (int r = threadIdx.x >> 5; r < nout_rows; r += MB_THREADS / 32)
            for (int j = threadIdx.x & 31; j < ww; j += 32) fo[(size_t)(y0 + r) * ww + j] = A[(P.halo_top + r) * ww + j];
    }
    if (out_mask) {
        uint8_t* fm = out_mask + (size_t)frame * P.h * P.w;
        const int gpr = ww * 4;  // 16-px groups per row
        for (int r = threadIdx.x >> 7; r < nout_rows; r += MB_THREADS / 128) {
            uint8_t* drow = fm + (size_t)(y0 + r) * P.w;
            const bool vec_ok = (((uintptr_t)drow) & 15) == 0;
            const u64* arow = A + (P.halo_top + r) * ww;
            for (int g = threadIdx.x & 127; g < gpr; g += 128)
                store_mask16(drow, g * 16, P.w, (u32)(arow[g >> 2] >> (16 * (g & 3))) & 0xffffu, vec_ok);
        }
    }
}

// Vertical half of a stage, B -> A: a thread takes one word column and a run of K consecutive rows, loads the K + 2R rows its window
// covers ONCE and slides over them in registers (windows of 2, 4, 8 rows by doubling, the odd row last): (K + 2R) / K LDS reads and
// about log2(2R) operations per output instead of 2R + 1 reads with two comparisons each.  Rows outside the image never win: min and
// max are idempotent, so reading the nearest image row again is the same as leaving them out - the row index is clamped to
// [lo_r, hi_r] and no predicate is left (rows of the image that are not staged only feed halo rows nobody needs any more).
template <int R, bool DIL, int K>
__device__ __forceinline__ void mb_vpass(const u64* __restrict__ B, u64* __restrict__ A, int ww, int rows, int lo_r, int hi_r)
{
    static_assert(R == 1 || R == 2 || R == 4, "window sizes built by doubling");
    const int r0 = (int)(threadIdx.x >> 5) * K;
    if (r0 >= rows) return;
    auto op = [](u64 a, u64 b) -> u64 { return DIL ? (a | b) : (a & b); };
    for (int j = threadIdx.x & 31; j < ww; j += 32) {
        u64 v[K + 2 * R];
#pragma unroll
        for (int k = 0; k < K + 2 * R; k++) v[k] = B[min(max(r0 - R + k, lo_r), hi_r) * ww + j];
        u64 out[K];
        if constexpr (R == 1) {
#pragma unroll
            for (int i = 0; i < K; i++) out[i] = op(op(v[i], v[i + 1]), v[i + 2]);
        } else {
            u64 p[K + 2 * R - 1];
#pragma unroll
            for (int i = 0; i < K + 2 * R - 1; i++) p[i] = op(v[i], v[i + 1]);                   // 2 rows
            u64 q[K + 2 * R - 3];
#pragma unroll
            for (int i = 0; i < K + 2 * R - 3; i++) q[i] = op(p[i], p[i + 2]);                   // 4 rows
            if constexpr (R == 2) {
#pragma unroll
                for (int i = 0; i < K; i++) out[i] = op(q[i], v[i + 4]);
            } else {
#pragma unroll
                for (int i = 0; i < K; i++) out[i] = op(op(q[i], q[i + 4]), v[i + 8]);           // 8 rows and the ninth
            }
        }
#pragma unroll
        for (int i = 0; i < K; i++)
            if (r0 + i < rows) A[(r0 + i) * ww + j] = out[i];
    }
}

// one erode / dilate of radius R on the staged rows: horizontal half A -> B, vertical half B -> A
template <int R, bool dil, int rows>
__device__ __forceinline__ void mb_stage(u64* __restrict__ A, u64* __restrict__ B, int ww, int ybase, int h, u64 lastmask, int lo_r, int hi_r)
{
    // horizontal half: all reads of a thread first (its word column in every 16th row, neighbours at clamped indices and replaced
    // by the neutral word afterwards - no branch), then straight-line shifts.  Rows outside the image are computed like any other:
    // nothing reads them (mb_vpass clamps its rows to the image).
    constexpr u64 neutral = dil ? 0ull : ~0ull;
    constexpr int RQ = MB_THREADS / 32, NR = (rows + RQ - 1) / RQ;
    const int rq = threadIdx.x >> 5;
    for (int j = threadIdx.x & 31; j < ww; j += 32) {
        const bool first = j == 0, last = j == ww - 1;
        const int il = first ? 0 : -1, ir = last ? 0 : 1;
        u64 cur[NR], prv[NR], nxt[NR];
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const int r = rq + RQ * k;
            if (RQ * k + RQ <= rows || r < rows) {
                const int i = r * ww + j;
                cur[k] = A[i]; prv[k] = A[i + il]; nxt[k] = A[i + ir];
            }
        }
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const int r = rq + RQ * k;
            if (RQ * k + RQ <= rows || r < rows) {
                u64 c = cur[k], pv = first ? neutral : prv[k], nx = last ? neutral : nxt[k];
                if (!dil) {                                   // columns past the image inside the last word count as set
                    if (last) c |= ~lastmask;
                    if (j + 1 == ww - 1) nx |= ~lastmask;
                }
                u64 acc = c;
#pragma unroll
                for (int d = 1; d <= R; d++) {
                    if (dil) acc |= fsr(nx, c, d) | fsl(c, pv, d);
                    else acc &= fsr(nx, c, d) & fsl(c, pv, d);
                }
                if (last) acc &= lastmask;
                B[r * ww + j] = acc;
            }
        }
    }
    __syncthreads();
    mb_vpass<R, dil, (rows + MB_THREADS / 32 - 1) / (MB_THREADS / 32)>(B, A, ww, rows, lo_r, hi_r);
    __syncthreads();
}

// Compile-time specialisation for the plans the modules actually use (square kernels, centre anchor): radii and
// kinds are template constants, so the shift loops unroll into immediate funnel shifts and the vertical windows
// into straight-line LDS reads.  KIND bit k = stage k dilates.
template <int NS, int R0, int R1, int R2, int KIND, int STRIP>
__global__ __launch_bounds__(MB_THREADS, 8) void k_morph_bits_sym(const u64* __restrict__ in, int w, int h, int ww, int strips,
                                                               u64* __restrict__ out_bits, uint8_t* __restrict__ out_mask, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int HALO = R0 + (NS > 1 ? R1 : 0) + (NS > 2 ? R2 : 0);
    constexpr int rows = STRIP + 2 * HALO;
    u64* A = lds;
    u64* B = lds + (size_t)rows * ww;
    const int frame = blockIdx.x / strips;
    const int strip = blockIdx.x - frame * strips;
    const int y0 = strip * STRIP;
    const int ybase = y0 - HALO;
    const u64* fin = in + (size_t)frame * h * ww;
    const u64 lastmask = (w & 63) ? ((1ull << (w & 63)) - 1ull) : ~0ull;
    const int lo_r = max(0, -ybase), hi_r = min(rows - 1, h - 1 - ybase);      // the staged rows that are rows of the image
    MB_FOR_WORDS(r, j, i) {
        const int y = ybase + r;
        A[i] = (y >= 0 && y < h && !(dbg & 2)) ? fin[(size_t)y * ww + j] : 0ull;
    }
    __syncthreads();
    if (!(dbg & 1)) {
        mb_stage<R0, (KIND & 1) != 0, rows>(A, B, ww, ybase, h, lastmask, lo_r, hi_r);
        if constexpr (NS > 1) mb_stage<R1, ((KIND >> 1) & 1) != 0, rows>(A, B, ww, ybase, h, lastmask, lo_r, hi_r);
        if constexpr (NS > 2) mb_stage<R2, ((KIND >> 2) & 1) != 0, rows>(A, B, ww, ybase, h, lastmask, lo_r, hi_r);
    }
    const int nout_rows = min(STRIP, h - y0);
    if (out_bits && !(dbg & 4)) {
        u64* fo = out_bits + (size_t)frame * h * ww;
        for (int r = threadIdx.x >> 5; r < nout_rows; r += MB_THREADS / 32)
            for (int j = threadIdx.x & 31; j < ww; j += 32) fo[(size_t)(y0 + r) * ww + j] = A[(HALO + r) * ww + j];
    }
    if (out_mask && !(dbg & 8)) {
        uint8_t* fm = out_mask + (size_t)frame * h * w;
        const int gpr = ww * 4;
        for (int r = threadIdx.x >> 7; r < nout_rows; r += MB_THREADS / 128) {
            uint8_t* drow = fm + (size_t)(y0 + r) * w;
            const bool vec_ok = (((uintptr_t)drow) & 15) == 0;
            const u64* arow = A + (HALO + r) * ww;
            for (int g = threadIdx.x & 127; g < gpr; g += 128)
                store_mask16(drow, g * 16, w, (u32)(arow[g >> 2] >> (16 * (g & 3))) & 0xffffu, vec_ok);
        }
    }
}

template <int NS, int R0, int R1, int R2, int KIND, int STRIP>
static int launch_sym_strip(vp_ctx* ctx, const u64* d_in, int w, int h, int n, u64* d_out_bits, uint8_t* d_out_mask)
{
    constexpr int HALO = R0 + (NS > 1 ? R1 : 0) + (NS > 2 ? R2 : 0);
    const int ww = vp_ww(w), strips = (h + STRIP - 1) / STRIP;
    const size_t lds = (size_t)2 * (STRIP + 2 * HALO) * ww * sizeof(u64);
    if (lds > 64 * 1024) return VP_ERR_UNSUPPORTED;
    vp_prof_scope prof(ctx, VPK_MORPH);
    hipLaunchKernelGGL((k_morph_bits_sym<NS, R0, R1, R2, KIND, STRIP>), dim3((unsigned)((size_t)n * strips)), dim3(MB_THREADS), lds, ctx->stream, d_in, w, h,
                       ww, strips, d_out_bits, d_out_mask, getenv("VP_MORPH_DBG") ? atoi(getenv("VP_MORPH_DBG")) : 0);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// 64-row strips when they fit 40 KB of LDS (four blocks per CU; 1080p: 38 KB): the halo rows every stage recomputes are then 20 % of the
// strip instead of 33 % (74 instead of 78 us per 128 frames); wider frames keep 32 rows
template <int NS, int R0, int R1, int R2, int KIND>
static int launch_sym(vp_ctx* ctx, const u64* d_in, int w, int h, int n, u64* d_out_bits, uint8_t* d_out_mask)
{
    constexpr int HALO = R0 + (NS > 1 ? R1 : 0) + (NS > 2 ? R2 : 0);
    static const bool tall_ok = getenv("VP_MORPH_STRIP32") == nullptr;
    if (tall_ok && h > 64 && (size_t)2 * (64 + 2 * HALO) * vp_ww(w) * sizeof(u64) <= 40 * 1024)
        return launch_sym_strip<NS, R0, R1, R2, KIND, 64>(ctx, d_in, w, h, n, d_out_bits, d_out_mask);
    return launch_sym_strip<NS, R0, R1, R2, KIND, MB_STRIP>(ctx, d_in, w, h, n, d_out_bits, d_out_mask);
}

// returns VP_ERR_UNSUPPORTED when the plan is not one of the specialised shapes
static int try_sym(vp_ctx* ctx, const vp_bitplan& plan, const u64* d_in, int w, int h, int n, u64* ob, uint8_t* om)
{
    if (plan.n < 1 || plan.n > 3) return VP_ERR_UNSUPPORTED;
    int rad[3] = {0, 0, 0}, kind = 0;
    for (int i = 0; i < plan.n; i++) {
        const vp_bitstage& s = plan.s[i];
        if (s.l != s.r || s.l != s.u || s.l != s.d) return VP_ERR_UNSUPPORTED;
        rad[i] = s.l;
        kind |= (s.dilate ? 1 : 0) << i;
    }
#define SYM(NS, A, B, C, K) if (plan.n == NS && rad[0] == A && rad[1] == B && rad[2] == C && kind == K) return launch_sym<NS, A, B, C, K>(ctx, d_in, w, h, n, ob, om)
    SYM(3, 2, 4, 2, 2);   // OPEN 5x5 + CLOSE 5x5  (erode, dilate x2 merged, erode)      — modules/red_buoy.py:31-33
    SYM(3, 2, 4, 2, 5);   // CLOSE 5x5 + OPEN 5x5
    SYM(2, 2, 2, 0, 2);   // OPEN 5x5                                                   — modules/bins.py:23-24
    SYM(2, 2, 2, 0, 1);   // CLOSE 5x5
    SYM(3, 1, 2, 1, 2);   // OPEN 3x3 + CLOSE 3x3
    SYM(3, 1, 2, 1, 5);
    SYM(2, 1, 1, 0, 2);   // OPEN 3x3
    SYM(2, 1, 1, 0, 1);   // CLOSE 3x3
    SYM(1, 1, 0, 0, 0); SYM(1, 1, 0, 0, 1);   // erode / dilate 3x3
    SYM(1, 2, 0, 0, 0); SYM(1, 2, 0, 0, 1);   // erode / dilate 5x5
#undef SYM
    return VP_ERR_UNSUPPORTED;
}

int vpk_morph_bits(vp_ctx* ctx, const vp_bitplan& plan, const u64* d_in, int w, int h, int n, u64* d_out_bits,
                   uint8_t* d_out_mask)
{
    for (int i = 0; i < plan.n; i++)
        if (plan.s[i].l > 31 || plan.s[i].r > 31 || plan.s[i].l < 0 || plan.s[i].r < 0 || plan.s[i].u < 0 || plan.s[i].d < 0)
            return vp_fail(ctx, VP_ERR_INVALID, "bit stage extent");
    {
        static const bool no_sym = getenv("VP_NO_SYM") != nullptr;
        const int rc = no_sym ? VP_ERR_UNSUPPORTED : try_sym(ctx, plan, d_in, w, h, n, d_out_bits, d_out_mask);
        if (rc != VP_ERR_UNSUPPORTED) return rc;
    }
    mb_params P;
    P.w = w;
    P.h = h;
    P.ww = vp_ww(w);
    P.plan = plan;
    P.halo_top = P.halo_bot = 0;
    for (int i = 0; i < plan.n; i++) {
        if (plan.s[i].l > 31 || plan.s[i].r > 31 || plan.s[i].l < 0 || plan.s[i].r < 0 || plan.s[i].u < 0 || plan.s[i].d < 0)
            return vp_fail(ctx, VP_ERR_INVALID, "bit stage extent");
        P.halo_top += plan.s[i].u;
        P.halo_bot += plan.s[i].d;
    }
    P.rows = MB_STRIP + P.halo_top + P.halo_bot;
    P.strips = (h + MB_STRIP - 1) / MB_STRIP;
    const size_t lds = (size_t)2 * P.rows * P.ww * sizeof(u64);
    if (lds > 160 * 1024) return VP_ERR_UNSUPPORTED;  // caller splits the plan
    if (lds > 64 * 1024)
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_morph_bits), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    vp_prof_scope prof(ctx, VPK_MORPH);
    hipLaunchKernelGGL(k_morph_bits, dim3((unsigned)((size_t)n * P.strips)), dim3(MB_THREADS), lds, ctx->stream, d_in, P,
                       d_out_bits, d_out_mask);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// ---- generic grey-level morphology: arbitrary structuring element ---------------------------------

// offs: noffs pairs (dx, dy) relative to the anchor.  One thread per (x, y, channel).
__global__ __launch_bounds__(256) void k_morph_generic(int dilate, const uint8_t* __restrict__ src, int w, int h, int cn,
                                                       const int16_t* __restrict__ offs, int noffs, uint8_t* __restrict__ dst)
{
    const int xc = blockIdx.x * 256 + threadIdx.x;  // x*cn + c
    const int y = blockIdx.y;
    if (xc >= w * cn) return;
    const int x = xc / cn, c = xc - x * cn;
    int best = dilate ? 0 : 255;
    for (int k = 0; k < noffs; k++) {
        const int xx = x + offs[2 * k], yy = y + offs[2 * k + 1];
        if (xx < 0 || xx >= w || yy < 0 || yy >= h) continue;
        const int v = src[((size_t)yy * w + xx) * cn + c];
        best = dilate ? max(best, v) : min(best, v);
    }
    dst[((size_t)y * w + x) * cn + c] = (uint8_t)best;
}

int vpk_morph_generic(vp_ctx* ctx, int dilate, const uint8_t* d_src, int w, int h, int cn, const int16_t* d_offs, int noffs,
                      uint8_t* d_dst)
{
    dim3 grid((unsigned)((w * cn + 255) / 256), (unsigned)h);
    hipLaunchKernelGGL(k_morph_generic, grid, dim3(256), 0, ctx->stream, dilate, d_src, w, h, cn, d_offs, noffs, d_dst);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// ---- span form of the same operator ------------------------------------------------------------------------------------------
// A structuring element is a list of horizontal spans (dy, x0..x1), one per run of members in each of its rows (an ellipse
// has one span per row).  With running min/max tables over windows of 1, 2, 4, ... 128 pixels of every row
// (T_j[x] = op over [x, x + 2^j - 1], clipped at the row end), the op over any span is the op of two table entries, so a
// pixel costs two reads per span instead of one per member: 202 instead of 7,845 for the 101 x 101 ellipse of
// modules/preprocessor.py:120-129.  Out-of-image pixels never win (cv2's default border for morphology) because spans are
// clipped to the row and rows outside the image are skipped.
#define MS_MAX_LEVELS 7          // windows up to 255 pixels
#define MS_MAX_ROWBYTES 16384    // one row in LDS, twice
#define MS_MAX_SPANS 2048

// one block per row: tab[j-1][y][p] for j = 1..levels
__global__ __launch_bounds__(256) void k_morph_table(int dilate, const uint8_t* __restrict__ src, int rowbytes, int cn, int levels, size_t plane,
                                                     uint8_t* __restrict__ tab)
{
    extern __shared__ uint8_t ms_lds[];
    uint8_t* A = ms_lds;
    uint8_t* B = ms_lds + rowbytes;
    const int y = blockIdx.x;
    const uint8_t* row = src + (size_t)y * rowbytes;
    for (int p = threadIdx.x; p < rowbytes; p += 256) A[p] = row[p];
    __syncthreads();
    for (int j = 1; j <= levels; j++) {
        const int s = (1 << (j - 1)) * cn;
        uint8_t* out = tab + (size_t)(j - 1) * plane + (size_t)y * rowbytes;
        for (int p = threadIdx.x; p < rowbytes; p += 256) {
            const int a = A[p], b = p + s < rowbytes ? A[p + s] : a;
            const uint8_t v = (uint8_t)(dilate ? max(a, b) : min(a, b));
            B[p] = v;
            out[p] = v;
        }
        __syncthreads();
        uint8_t* t = A; A = B; B = t;
    }
}

// spans: nspans triples (dy, x0, x1) relative to the anchor.  One thread per (x, channel) of row blockIdx.y.
__global__ __launch_bounds__(256) void k_morph_spans(int dilate, const uint8_t* __restrict__ src, const uint8_t* __restrict__ tab, size_t plane, int w,
                                                     int h, int cn, const int16_t* __restrict__ spans, int nspans, uint8_t* __restrict__ dst)
{
    __shared__ int16_t sp[3 * MS_MAX_SPANS];
    for (int i = threadIdx.x; i < 3 * nspans; i += 256) sp[i] = spans[i];
    __syncthreads();
    const int xc = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (xc >= w * cn) return;
    const int x = xc / cn, c = xc - x * cn;
    const size_t rowbytes = (size_t)w * cn;
    int best = dilate ? 0 : 255;
    for (int k = 0; k < nspans; k++) {
        const int yy = y + sp[3 * k];
        if (yy < 0 || yy >= h) continue;
        const int l = max(x + sp[3 * k + 1], 0), r = min(x + sp[3 * k + 2], w - 1);
        if (l > r) continue;
        const int j = 31 - __clz(r - l + 1);
        const uint8_t* T = (j == 0 ? src : tab + (size_t)(j - 1) * plane) + (size_t)yy * rowbytes + c;
        const int a = T[(size_t)l * cn], b = T[(size_t)(r - (1 << j) + 1) * cn];
        best = dilate ? max(best, max(a, b)) : min(best, min(a, b));
    }
    dst[(size_t)y * rowbytes + xc] = (uint8_t)best;
}

int vpk_morph_spans(vp_ctx* ctx, int dilate, const uint8_t* d_src, int w, int h, int cn, const int16_t* d_spans, int nspans, int max_len,
                    uint8_t* d_tab, uint8_t* d_dst)
{
    const int rowbytes = w * cn;
    int levels = 0;
    while ((2 << levels) <= max_len) levels++;   // largest j with 2^j <= max_len
    if (levels > MS_MAX_LEVELS || rowbytes > MS_MAX_ROWBYTES || nspans > MS_MAX_SPANS) return vp_fail(ctx, VP_ERR_UNSUPPORTED, "span morphology limits");
    const size_t plane = (size_t)rowbytes * h;
    if (levels > 0)
        hipLaunchKernelGGL(k_morph_table, dim3((unsigned)h), dim3(256), 2 * (size_t)rowbytes, ctx->stream, dilate, d_src, rowbytes, cn, levels, plane, d_tab);
    dim3 grid((unsigned)((rowbytes + 255) / 256), (unsigned)h);
    hipLaunchKernelGGL(k_morph_spans, grid, dim3(256), 0, ctx->stream, dilate, d_src, d_tab, plane, w, h, cn, d_spans, nspans, d_dst);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

__global__ __launch_bounds__(256) void k_sub_sat_u8(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, size_t n,
                                                    uint8_t* __restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int d = (int)a[i] - (int)b[i];
    dst[i] = (uint8_t)(d < 0 ? 0 : d);
}

// dst = saturate(round-half-even(a * alpha + b * beta + gamma)), every operation a correctly rounded double (no contraction): the
// statement the cv2 stand-in makes in numpy float64 (vision/cv2_facade.py addWeighted)
__global__ __launch_bounds__(256) void k_add_weighted_u8(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, size_t n, double alpha,
                                                         double beta, double gamma, uint8_t* __restrict__ dst)
{
    // lane = 16 bytes = one 16-B load per input and one 16-B store (aligned pointers; the last n % 16 bytes and unaligned images byte-wise)
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i0 >= n) return;
    auto one = [&](int x, int y) -> unsigned {
        const double v = __dadd_rn(__dadd_rn(__dmul_rn((double)x, alpha), __dmul_rn((double)y, beta)), gamma);
        const double r = rint(v);
        return (unsigned)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
    };
    if (i0 + 16 <= n && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)dst) & 15u) == 0) {
        const uint4 qa = *reinterpret_cast<const uint4*>(a + i0), qb = *reinterpret_cast<const uint4*>(b + i0);
        const unsigned va[4] = {qa.x, qa.y, qa.z, qa.w}, vb[4] = {qb.x, qb.y, qb.z, qb.w};
        unsigned out[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; k++) out[k >> 2] |= one((int)((va[k >> 2] >> (8 * (k & 3))) & 255u), (int)((vb[k >> 2] >> (8 * (k & 3))) & 255u)) << (8 * (k & 3));
        vp_store16(dst + i0, out[0], out[1], out[2], out[3]);
    } else {
        for (size_t i = i0; i < n && i < i0 + 16; i++) dst[i] = (uint8_t)one(a[i], b[i]);
    }
}

// Overlay drawing into a device image (vp_draw_polylines_dev): one wave per segment.  The host rasteriser's Bresenham loop
// (vp_draw_polylines_u8 / vision/utils/draw.py _line: err = dx + dy; e2 = 2 err; x steps when e2 >= dy, y steps when e2 <= dx) always
// advances the longer axis, and after i steps the shorter one stands at floor((2 i m + M) / (2 M)) (m, M = the shorter and the longer
// extent; dx >= |dy| counts as x-major) - tests/test_draw.py checks that against the loop - so the steps of a segment are independent:
// lane = step.  nxt[g] = index of the point that point g is joined to (itself: a single point; -1: the open end of a polyline).
__device__ __forceinline__ void draw_segment_steps(uint8_t* __restrict__ img, int w, int h, int cn, int2 a, int2 b, int thickness, uchar4 color, int lane)
{
    const long long dx = llabs((long long)b.x - a.x), ady = llabs((long long)b.y - a.y);
    const int sx = a.x < b.x ? 1 : -1, sy = a.y < b.y ? 1 : -1;
    const long long n = dx > ady ? dx : ady;
    const int r0 = (thickness - 1) / 2;
    for (long long i = lane; i <= n; i += 64) {
        long long x = a.x, y = a.y;
        if (n > 0) {
            if (dx >= ady) { x += sx * i; y += sy * ((2 * i * ady + dx) / (2 * dx)); }
            else { y += sy * i; x += sx * ((2 * i * dx + ady) / (2 * ady)); }
        }
        if (x + thickness <= 0 || x - thickness >= w || y + thickness <= 0 || y - thickness >= h) continue;
        for (int by = 0; by < thickness; by++) {
            const long long py = y - r0 + by;
            if (py < 0 || py >= h) continue;
            for (int bx = 0; bx < thickness; bx++) {
                const long long px = x - r0 + bx;
                if (px < 0 || px >= w) continue;
                uint8_t* q = img + ((size_t)py * w + (size_t)px) * cn;
                q[0] = color.x;
                if (cn > 1) q[1] = color.y;
                if (cn > 2) q[2] = color.z;
                if (cn > 3) q[3] = color.w;
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_draw_segments(uint8_t* __restrict__ img, int w, int h, int cn, const int2* __restrict__ pts,
                                                       const int32_t* __restrict__ nxt, int npts, int thickness, uchar4 color)
{
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= npts) return;
    const int j = nxt[g];
    if (j < 0) return;
    draw_segment_steps(img, w, h, cn, pts[g], pts[j], thickness, color, lane);
}

// a few vertices (a box of bins.py: four) travel as kernel arguments: no copy, no staging
#define VP_DRAW_SMALL 48
struct draw_small { int2 pts[VP_DRAW_SMALL]; int16_t nxt[VP_DRAW_SMALL]; };
__global__ __launch_bounds__(256) void k_draw_small(uint8_t* __restrict__ img, int w, int h, int cn, draw_small D, int npts, int thickness, uchar4 color)
{
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= npts) return;
    const int j = D.nxt[g];
    if (j < 0) return;
    draw_segment_steps(img, w, h, cn, D.pts[g], D.pts[j], thickness, color, threadIdx.x & 63);
}
int vpk_draw_small(vp_ctx* ctx, uint8_t* d_img, int w, int h, int cn, const int32_t* pts, const int32_t* nxt, int npts, int thickness, const uint8_t* color)
{
    if (npts <= 0) return VP_OK;
    if (npts > VP_DRAW_SMALL) return vp_fail(ctx, VP_ERR_INVALID, "vpk_draw_small: too many points");
    draw_small D;
    for (int i = 0; i < npts; i++) { D.pts[i] = make_int2(pts[2 * i], pts[2 * i + 1]); D.nxt[i] = (int16_t)nxt[i]; }
    const uchar4 c = make_uchar4(color[0], cn > 1 ? color[1] : 0, cn > 2 ? color[2] : 0, cn > 3 ? color[3] : 0);
    hipLaunchKernelGGL(k_draw_small, dim3((unsigned)((npts + 3) / 4)), dim3(256), 0, ctx->stream, d_img, w, h, cn, D, npts, thickness, c);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

int vpk_draw_segments(vp_ctx* ctx, uint8_t* d_img, int w, int h, int cn, const int32_t* d_pts, const int32_t* d_nxt, int npts, int thickness,
                      const uint8_t* color)
{
    if (npts <= 0) return VP_OK;
    const uchar4 c = make_uchar4(color[0], cn > 1 ? color[1] : 0, cn > 2 ? color[2] : 0, cn > 3 ? color[3] : 0);
    hipLaunchKernelGGL(k_draw_segments, dim3((unsigned)((npts + 3) / 4)), dim3(256), 0, ctx->stream, d_img, w, h, cn, reinterpret_cast<const int2*>(d_pts), d_nxt,
                       npts, thickness, c);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

int vpk_add_weighted_u8(vp_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, double alpha, double beta, double gamma, uint8_t* dst)
{
    hipLaunchKernelGGL(k_add_weighted_u8, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, ctx->stream, a, b, n, alpha, beta, gamma, dst);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

int vpk_absdiff_sub_u8(vp_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* dst)
{
    hipLaunchKernelGGL(k_sub_sat_u8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, b, n, dst);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}
