// Colour conversion + inRange kernels for gfx950.
//
// Replaces (reference, via cv2): utils/color.py:11-32 (cvtColor + split), :105-121 (inRange),
// :66-103 (colour distance); modules/bins.py:13-16; modules/red_buoy.py:21-28.
//
// The chain kernel k_color_thresh_* fuses cvtColor + split + inRange: it reads packed BGR once
// (3 B/px), evaluates only the converted channels whose range is not trivially [0,255], and
// writes the 0/255 mask (1 B/px) plus a bit-packed copy (1/8 B/px) that the morphology and CCL
// kernels consume.  The arithmetic is OpenCV's 8-bit fixed point; the
// LUTs live in LDS.  HBM-bound: 4.125 B/px.
#include "vp_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

#define LAB_LSHIFT (-1336934)  // -((16*255*32768 + 50)/100)

struct LabLds { uint16_t gamma[256]; uint16_t cbrt[2048]; };
struct HsvLds { int32_t sdiv[256]; int32_t hdiv[256]; };

__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
// signed 24-bit multiply, low 32 bits of the product (full rate; v_mul_lo_u32 runs at a quarter of it).  Spelled as the instruction:
// __mul24 becomes it only where the compiler can bound both operands itself.
__device__ __forceinline__ int mul_i24(int a, int b)
{
    int r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// NEED bit0 = L, bit1 = a, bit2 = b
template <int NEED>
__device__ __forceinline__ void lab_px(const LabLds& t, int b, int g, int r, int& L, int& A, int& Bc)
{
    const int R = t.gamma[r], G = t.gamma[g], B = t.gamma[b];
    const int fY = t.cbrt[(R * 871 + G * 2929 + B * 296 + 2048) >> 12];
    if (NEED & 1) L = clamp255((296 * fY + LAB_LSHIFT + 16384) >> 15);
    if (NEED & 2) {
        const int fX = t.cbrt[(R * 1777 + G * 1541 + B * 778 + 2048) >> 12];
        A = clamp255((500 * (fX - fY) + (128 << 15) + 16384) >> 15);
    }
    if (NEED & 4) {
        const int fZ = t.cbrt[(R * 73 + G * 448 + B * 3575 + 2048) >> 12];
        Bc = clamp255((200 * (fY - fZ) + (128 << 15) + 16384) >> 15);
    }
}

__device__ __forceinline__ void hsv_px(const HsvLds& t, int b, int g, int r, int& H, int& S, int& V)
{
    int v = max(max(b, g), r);
    int vmin = min(min(b, g), r);
    int diff = v - vmin;
    S = (mul_i24(diff, t.sdiv[v]) + 2048) >> 12;   // (operands of 8 x 20 and 12 x 17 bits: the 24-bit multiply is exact)
    int hh = (v == r) ? (g - b) : ((v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff));
    hh = (mul_i24(hh, t.hdiv[diff]) + 2048) >> 12;
    hh += hh < 0 ? 180 : 0;
    H = hh;            // in [0, 180): the three branches give [-30, 30], [30, 90], [90, 150] before the wrap, so no clamp is needed
    V = v;
}

// ---- BGR -> Lab inside the threshold kernels ------------------------------------------------------------------------------------------
// The threshold kernels do not need L, a, b themselves, only "lo <= channel <= hi".  Every channel is a clamped, monotonic function
// of one integer - L of fY, a of fX - fY, b of fY - fZ (lab_px above) - so the host turns each range into the interval of that integer
// (lab_interval) and the kernel compares there: no multiply-shift-clamp per pixel.  And the three gamma look-ups + nine multiply-adds
// of the X / Y / Z sums become three look-ups of pre-multiplied (X, Y) pairs + adds: coefficient * gamma[value] per input channel,
// the rounding constant folded into the blue entries.  Same integers as OpenCV's statement sequence, ≈ half the VALU instructions
// (the kernel was bound by them, not by HBM: it took the same time without its mask stores).
struct LabTLds { uint2 xy[3][256]; u32 z[3][256]; uint16_t cbrt[2048]; };   // [0] blue, [1] green, [2] red

__device__ __forceinline__ bool in_span(int v, int lo, int hi)             // lo <= v <= hi as one unsigned comparison; lo > hi: never
{
    const bool empty = lo > hi;                                            // wave-uniform (kernel arguments)
    const u32 l = empty ? 0x7fffffffu : (u32)lo, span = empty ? 0u : (u32)hi - (u32)lo;
    return (u32)v - l <= span;
}

// q holds the intervals of fY, fX - fY, fY - fZ (vpk_color_thresh)
template <int NEED>
__device__ __forceinline__ bool lab_test(const LabTLds& t, const vp_range3& q, int b, int g, int r)
{
    const uint2 pb = t.xy[0][b], pg = t.xy[1][g], pr = t.xy[2][r];
    const int fY = t.cbrt[(pb.y + pg.y + pr.y) >> 12];
    bool ok = true;
    if (NEED & 1) ok = ok & in_span(fY, q.lo[0], q.hi[0]);
    if (NEED & 2) {
        const int fX = t.cbrt[(pb.x + pg.x + pr.x) >> 12];
        ok = ok & in_span(fX - fY, q.lo[1], q.hi[1]);
    }
    if (NEED & 4) {
        const int fZ = t.cbrt[(t.z[0][b] + t.z[1][g] + t.z[2][r]) >> 12];
        ok = ok & in_span(fY - fZ, q.lo[2], q.hi[2]);
    }
    return ok;
}

// BGR -> HSV inside the threshold kernels, the same idea: S and H are floor((product + 2048) / 4096) of a product the kernel has
// anyway, so their ranges are compared as intervals of the products (hsv_intervals); the hue's "+ 180 when negative" makes its range two
// intervals, one for each sign.  q: [0] and lo2 / hi2 = the two intervals of hh * hdiv[diff], [1] = interval of diff * sdiv[v], [2] = V.
__device__ __forceinline__ bool hsv_test(const HsvLds& t, const vp_range3& q, int b, int g, int r)
{
    const int v = max(max(b, g), r);
    const int diff = v - min(min(b, g), r);
    const int ps = mul_i24(diff, t.sdiv[v]);          // 8 x 20 bits, and 12 x 17 bits below: the full-rate 24-bit multiply is exact (v_mul_lo_u32 runs at a quarter of the rate)
    const int hh = (v == r) ? (g - b) : ((v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff));
    const int ph = mul_i24(hh, t.hdiv[diff]);
    return (in_span(ph, q.lo[0], q.hi[0]) | in_span(ph, q.lo2, q.hi2)) & in_span(ps, q.lo[1], q.hi[1]) & in_span(v, q.lo[2], q.hi[2]);
}

__device__ __forceinline__ int gray_px(int b, int g, int r) { return (b * 1868 + g * 9617 + r * 4899 + 8192) >> 14; }

// lo <= c <= hi per channel as one unsigned comparison each: (unsigned)(c - lo) <= (unsigned)(hi - lo).  For an empty range (lo > hi) the
// span is taken as 0 and lo as INT_MAX, so that c - lo wraps to something large and the test fails for every c.
__device__ __forceinline__ bool in3(const vp_range3& q, int c0, int c1, int c2)
{
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int c = k == 0 ? c0 : (k == 1 ? c1 : c2);
        const bool empty = q.lo[k] > q.hi[k];                              // wave-uniform (kernel arguments)
        const u32 lo = empty ? 0x7fffffffu : (u32)q.lo[k], span = empty ? 0u : (u32)(q.hi[k] - q.lo[k]);
        ok = ok & ((u32)c - lo <= span);
    }
    return ok;
}

template <int MODE>
struct ModeLds;
template <>
struct ModeLds<VP_BGR2LAB> { typedef LabLds type; };
template <>
struct ModeLds<VP_BGR2HSV> { typedef HsvLds type; };
template <>
struct ModeLds<VP_BGR2GRAY> { typedef int type; };
template <>
struct ModeLds<VP_BGR2YCRCB> { typedef int type; };
template <>
struct ModeLds<VP_BGR2HLS> { typedef int type; };

template <int MODE>
__device__ __forceinline__ void load_lds(typename ModeLds<MODE>::type& s, const vp_tables& tab)
{
    if constexpr (MODE == VP_BGR2LAB) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) s.gamma[i] = tab.gamma[i];
        for (int i = threadIdx.x; i < 2048; i += blockDim.x) s.cbrt[i] = tab.cbrt[i];
    } else if constexpr (MODE == VP_BGR2HSV) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) { s.sdiv[i] = tab.sdiv[i]; s.hdiv[i] = tab.hdiv[i]; }
    }
    __syncthreads();
}

// LDS of the threshold kernels: the pre-multiplied Lab tables, otherwise what the conversions use
template <int MODE>
struct ThreshLds { typedef typename ModeLds<MODE>::type type; };
template <>
struct ThreshLds<VP_BGR2LAB> { typedef LabTLds type; };

template <int MODE>
__device__ __forceinline__ void load_tlds(typename ThreshLds<MODE>::type& s, const vp_tables& tab)
{
    if constexpr (MODE == VP_BGR2LAB) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) {
            const u32 G = tab.gamma[i];
            s.xy[0][i] = make_uint2(778u * G + 2048u, 296u * G + 2048u);  s.z[0][i] = 3575u * G + 2048u;   // blue (+ the rounding constant)
            s.xy[1][i] = make_uint2(1541u * G, 2929u * G);                s.z[1][i] = 448u * G;            // green
            s.xy[2][i] = make_uint2(1777u * G, 871u * G);                 s.z[2][i] = 73u * G;             // red
        }
        for (int i = threadIdx.x; i < 2048; i += blockDim.x) s.cbrt[i] = tab.cbrt[i];
        __syncthreads();
    } else {
        load_lds<MODE>(s, tab);
    }
}

// predicate for one pixel
template <int MODE, int NEED>
__device__ __forceinline__ bool px_pred(const typename ThreshLds<MODE>::type& s, const vp_range3& q, int b, int g, int r)
{
    if constexpr (MODE == VP_BGR2LAB) {
        return lab_test<NEED>(s, q, b, g, r);
    } else if constexpr (MODE == VP_BGR2HSV) {
        return hsv_test(s, q, b, g, r);
    } else {
        int y = gray_px(b, g, r);
        return (y >= q.lo[0]) & (y <= q.hi[0]);
    }
}

#define BYTE_OF(arr, i) (((arr)[(i) >> 2] >> (8 * ((i)&3))) & 0xffu)

__device__ __forceinline__ u32 expand4(u32 nib)  // 4 bits -> 4 bytes of 0x00/0xFF
{
    return (((nib & 0xfu) * 0x00204081u) & 0x01010101u) * 0xffu;
}

// Flat fast path: frames contiguous, w % 64 == 0, base 16-B aligned.  One lane = 16 px = 48 B.
template <int MODE, int NEED, bool WMASK, bool WBITS>
__global__ __launch_bounds__(256, 8) void k_color_thresh_flat(const uint8_t* __restrict__ src, size_t ngroups,
                                                           vp_tables tab, vp_range3 q, uint8_t* __restrict__ mask,
                                                           u64* __restrict__ bits)
{
    __shared__ typename ThreshLds<MODE>::type s;
    load_tlds<MODE>(s, tab);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= ngroups) return;   // ngroups % 4 == 0 and g is quad-aligned: the 4 lanes of a word leave together
    const uint4* p = reinterpret_cast<const uint4*>(src + g * 48);
    uint4 v0 = p[0], v1 = p[1], v2 = p[2];
    for (;;) {
        // prefetch the next group before this one's stores are issued: on CDNA4 vmcnt counts stores too, so
        // a load issued after a store would wait for the store's completion
        const size_t gn = g + stride;
        const bool more = gn < ngroups;
        uint4 n0 = v0, n1 = v1, n2 = v2;
        if (more) {
            const uint4* pn = reinterpret_cast<const uint4*>(src + gn * 48);
            n0 = pn[0]; n1 = pn[1]; n2 = pn[2];
        }
        const u32 in[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
        u32 m = 0;
        if constexpr (MODE == VP_BGR2HSV) {
            // four pixels at a time: fully unrolled, the HSV arithmetic of 16 pixels keeps 118 VGPRs live (4 waves per SIMD)
#pragma unroll
            for (int k4 = 0; k4 < 4; k4++) {
                const u32 w0 = in[3 * k4], w1 = in[3 * k4 + 1], w2 = in[3 * k4 + 2];   // 12 bytes = 4 pixels
                const u32 three[3] = {w0, w1, w2};
                u32 mm = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int b = BYTE_OF(three, 3 * k), gg = BYTE_OF(three, 3 * k + 1), r = BYTE_OF(three, 3 * k + 2);
                    mm |= (u32)px_pred<MODE, NEED>(s, q, b, gg, r) << k;
                }
                m |= mm << (4 * k4);
                asm volatile("" : "+v"(m));   // keep the groups apart: no interleaving of their live ranges
            }
        } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int b = BYTE_OF(in, 3 * k), gg = BYTE_OF(in, 3 * k + 1), r = BYTE_OF(in, 3 * k + 2);
            m |= (u32)px_pred<MODE, NEED>(s, q, b, gg, r) << k;
        }
        }
        if (WMASK) {
            vp_store16(mask + g * 16, expand4(m), expand4(m >> 4), expand4(m >> 8), expand4(m >> 12));
        }
        if (WBITS) {
            u64 wv = (u64)m << (16 * (threadIdx.x & 3));
            wv |= __shfl_xor(wv, 1);
            wv |= __shfl_xor(wv, 2);
            if ((threadIdx.x & 3) == 0) bits[g >> 2] = wv;
        }
        if (!more) break;
        v0 = n0; v1 = n1; v2 = n2;
        g = gn;
    }
}

// Generic path: any w / stride / alignment.  grid.x = n*h rows; thread = 16-px group in the row.
template <int MODE, int NEED>
__global__ __launch_bounds__(256) void k_color_thresh_rows(const uint8_t* __restrict__ src, size_t stride, int w, int ww,
                                                           vp_tables tab, vp_range3 q, uint8_t* __restrict__ mask,
                                                           u64* __restrict__ bits)
{
    __shared__ typename ThreshLds<MODE>::type s;
    load_tlds<MODE>(s, tab);
    const size_t row = blockIdx.x;
    const int grp = blockIdx.y * 256 + threadIdx.x;  // 16-px group, 4 per word
    const bool live = grp < ww * 4;
    const uint8_t* p = src + row * stride;
    u32 m = 0;
    if (live) {
#pragma unroll 4
        for (int k = 0; k < 16; k++) {
            const int x = grp * 16 + k;
            if (x < w) {
                const int b = p[3 * x], gg = p[3 * x + 1], r = p[3 * x + 2];
                const bool ok = px_pred<MODE, NEED>(s, q, b, gg, r);
                m |= (u32)ok << k;
                if (mask) mask[row * (size_t)w + x] = ok ? 255 : 0;
            }
        }
    }
    if (bits) {
        u64 wv = (u64)m << (16 * (threadIdx.x & 3));
        wv |= __shfl_xor(wv, 1);
        wv |= __shfl_xor(wv, 2);
        if (live && (threadIdx.x & 3) == 0) bits[row * (size_t)ww + (grp >> 2)] = wv;
    }
}

template <int MODE, int NEED>
static int launch_thresh(vp_ctx* ctx, const uint8_t* d_bgr, size_t stride, int w, int h, int n, const vp_range3& q,
                         uint8_t* d_mask, u64* d_bits)
{
    const int ww = vp_ww(w);
    vp_prof_scope prof(ctx, VPK_COLOR);
    const bool flat = (w % 64 == 0) && stride == (size_t)w * 3 && ((uintptr_t)d_bgr % 16 == 0) &&
                      (d_mask == nullptr || (uintptr_t)d_mask % 16 == 0);
    if (flat) {
        const size_t ngroups = (size_t)n * h * w / 16;
        size_t blocks = (ngroups + 255) / 256;
        static const int bpc = getenv("VP_COLOR_BPC") ? atoi(getenv("VP_COLOR_BPC")) : 64;
        const size_t cap = (size_t)ctx->num_cu * (bpc > 0 ? bpc : 64);
        if (blocks > cap) blocks = cap;
        dim3 grid((unsigned)blocks);
        if (d_mask && d_bits)
            hipLaunchKernelGGL((k_color_thresh_flat<MODE, NEED, true, true>), grid, dim3(256), 0, ctx->stream, d_bgr, ngroups, ctx->tab, q, d_mask, d_bits);
        else if (d_mask)
            hipLaunchKernelGGL((k_color_thresh_flat<MODE, NEED, true, false>), grid, dim3(256), 0, ctx->stream, d_bgr, ngroups, ctx->tab, q, d_mask, d_bits);
        else
            hipLaunchKernelGGL((k_color_thresh_flat<MODE, NEED, false, true>), grid, dim3(256), 0, ctx->stream, d_bgr, ngroups, ctx->tab, q, d_mask, d_bits);
    } else {
        dim3 grid((unsigned)((size_t)n * h), (unsigned)((ww * 4 + 255) / 256));
        hipLaunchKernelGGL((k_color_thresh_rows<MODE, NEED>), grid, dim3(256), 0, ctx->stream, d_bgr, stride, w, ww, ctx->tab, q, d_mask, d_bits);
    }
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// lo <= clamp255((mult * v + add) >> 15) <= hi  <=>  *vlo <= v <= *vhi (mult > 0; >> is the arithmetic shift = floor).  A bound the
// clamp satisfies for every v becomes +-2^24 (v itself stays within +-2^16); an impossible range comes back as vlo > vhi.
static void lab_interval(int lo, int hi, long long mult, long long add, int* vlo, int* vhi)
{
    const long long BIG = 1ll << 24;
    auto floor_div = [](long long a, long long b) { long long q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) q--; return q; };
    if (lo > hi || lo > 255 || hi < 0) { *vlo = 1; *vhi = 0; return; }
    long long a = -BIG, b = BIG;
    if (lo > 0) a = -floor_div(-((long long)lo * 32768 - add), mult);                 // ceil((lo * 2^15 - add) / mult)
    if (hi < 255) b = floor_div(((long long)hi + 1) * 32768 - 1 - add, mult);
    if (a < -BIG) a = -BIG;
    if (b > BIG) b = BIG;
    *vlo = (int)a; *vhi = (int)b;
}

// lo <= floor((p + 2048) / 4096) <= hi  <=>  *plo <= p <= *phi (bounds clamped so that the products stay inside int)
static void shift12_interval(long long lo, long long hi, int* plo, int* phi)
{
    if (lo > hi) { *plo = 1; *phi = 0; return; }
    lo = std::max(lo, -100000ll); hi = std::min(hi, 100000ll);
    *plo = (int)(lo * 4096 - 2048);
    *phi = (int)((hi + 1) * 4096 - 2049);
}
// the H / S / V ranges of an HSV threshold as what hsv_test compares: H = t (t >= 0) or t + 180 (t < 0) with t = floor((ph + 2048) / 4096)
static void hsv_intervals(const vp_range3& q, vp_range3* o)
{
    const long long hlo = q.lo[0], hhi = q.hi[0];
    shift12_interval(std::max(hlo, 0ll), hhi, &o->lo[0], &o->hi[0]);                      // t >= 0
    shift12_interval(hlo - 180, std::min(hhi - 180, -1ll), &o->lo2, &o->hi2);             // t < 0
    shift12_interval(std::max((long long)q.lo[1], 0ll), q.hi[1], &o->lo[1], &o->hi[1]);  // S >= 0 always
    o->lo[2] = q.lo[2]; o->hi[2] = q.hi[2];
    if (q.lo[0] > q.hi[0]) { o->lo[0] = o->lo2 = 1; o->hi[0] = o->hi2 = 0; }
}

int vpk_color_thresh(vp_ctx* ctx, int mode, const uint8_t* d_bgr, size_t stride, int w, int h, int n, const vp_range3& q,
                     uint8_t* d_mask, u64* d_bits)
{
    if (!d_mask && !d_bits) return VP_OK;
    if ((size_t)n * h > 0x7fffffffULL) return vp_fail(ctx, VP_ERR_INVALID, "batch too large");
    if (mode == VP_BGR2LAB) {
        int need = 0;
        for (int c = 0; c < 3; c++)
            if (!(q.lo[c] <= 0 && q.hi[c] >= 255)) need |= 1 << c;
        // the ranges of L, a, b as intervals of fY, fX - fY, fY - fZ (lab_px: channel = clamp255((mult * v + add) >> 15))
        vp_range3 qi;
        lab_interval(q.lo[0], q.hi[0], 296, (long long)LAB_LSHIFT + 16384, &qi.lo[0], &qi.hi[0]);
        lab_interval(q.lo[1], q.hi[1], 500, (128ll << 15) + 16384, &qi.lo[1], &qi.hi[1]);
        lab_interval(q.lo[2], q.hi[2], 200, (128ll << 15) + 16384, &qi.lo[2], &qi.hi[2]);
        switch (need) {
            case 0: need = 1;  // everything passes; evaluate L so that the kernel stays generic
            case 1: return launch_thresh<VP_BGR2LAB, 1>(ctx, d_bgr, stride, w, h, n, qi, d_mask, d_bits);
            case 2: return launch_thresh<VP_BGR2LAB, 2>(ctx, d_bgr, stride, w, h, n, qi, d_mask, d_bits);
            case 4: return launch_thresh<VP_BGR2LAB, 4>(ctx, d_bgr, stride, w, h, n, qi, d_mask, d_bits);
            default: return launch_thresh<VP_BGR2LAB, 7>(ctx, d_bgr, stride, w, h, n, qi, d_mask, d_bits);
        }
    }
    if (mode == VP_BGR2HSV) {
        vp_range3 qi;
        hsv_intervals(q, &qi);
        return launch_thresh<VP_BGR2HSV, 7>(ctx, d_bgr, stride, w, h, n, qi, d_mask, d_bits);
    }
    if (mode == VP_BGR2GRAY) return launch_thresh<VP_BGR2GRAY, 1>(ctx, d_bgr, stride, w, h, n, q, d_mask, d_bits);
    return vp_fail(ctx, VP_ERR_INVALID, "color mode");
}

// ---- standalone conversions (operator API) ----------------------------------------------------

// RGB2YCrCb_i<uchar>: Q14 integers, Y as in BGR2GRAY; stored Y, Cr, Cb
__device__ __forceinline__ void ycrcb_px(int b, int g, int r, int& c0, int& c1, int& c2)
{
    const int Y = (b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14;
    c0 = Y;
    c1 = min(max(((r - Y) * 11682 + (128 << 14) + (1 << 13)) >> 14, 0), 255);
    c2 = min(max(((b - Y) * 9241 + (128 << 14) + (1 << 13)) >> 14, 0), 255);
}
// RGB2HLS_b: float32 statement sequence of RGB2HLS_f (hrange 180) on src * (1/255); single correctly rounded operations, no contraction
__device__ __forceinline__ int hls_sat(float v) { return min(max((int)rintf(v), 0), 255); }
__device__ __forceinline__ void hls_px(int bi, int gi, int ri, int& c0, int& c1, int& c2)
{
    const float scale = 1.f / 255.f;
    const float b = __fmul_rn((float)bi, scale), g = __fmul_rn((float)gi, scale), r = __fmul_rn((float)ri, scale);
    const float vmax = fmaxf(r, fmaxf(g, b)), vmin = fminf(r, fminf(g, b));
    float diff = __fsub_rn(vmax, vmin);
    const float sum = __fadd_rn(vmax, vmin);
    const float l = __fmul_rn(sum, 0.5f);
    float hh = 0.f, sat = 0.f;
    if (diff > 1.1920928955078125e-7f) {
        sat = l < 0.5f ? __fdiv_rn(diff, sum) : __fdiv_rn(diff, __fsub_rn(__fsub_rn(2.f, vmax), vmin));
        diff = __fdiv_rn(60.f, diff);
        if (vmax == r) hh = __fmul_rn(__fsub_rn(g, b), diff);
        else if (vmax == g) hh = __fadd_rn(__fmul_rn(__fsub_rn(b, r), diff), 120.f);
        else hh = __fadd_rn(__fmul_rn(__fsub_rn(r, g), diff), 240.f);
        if (hh < 0.f) hh = __fadd_rn(hh, 360.f);
    }
    c0 = hls_sat(__fmul_rn(hh, 0.5f));
    c1 = hls_sat(__fmul_rn(l, 255.f));
    c2 = hls_sat(__fmul_rn(sat, 255.f));
}

template <int CODE>
__global__ __launch_bounds__(256) void k_cvt_color(const uint8_t* __restrict__ src, size_t stride, int w, int h, vp_tables tab,
                                                   uint8_t* __restrict__ dst, uint8_t* __restrict__ p0,
                                                   uint8_t* __restrict__ p1, uint8_t* __restrict__ p2)
{
    __shared__ typename ModeLds<CODE>::type s;
    load_lds<CODE>(s, tab);
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* p = src + (size_t)y * stride + 3 * (size_t)x;
    const int b = p[0], g = p[1], r = p[2];
    const size_t o = (size_t)y * w + x;
    if constexpr (CODE == VP_BGR2GRAY) {
        const uint8_t v = (uint8_t)gray_px(b, g, r);
        if (dst) dst[o] = v;
        if (p0) p0[o] = v;
    } else {
        int c0 = 0, c1 = 0, c2 = 0;
        if constexpr (CODE == VP_BGR2LAB) lab_px<7>(s, b, g, r, c0, c1, c2);
        else if constexpr (CODE == VP_BGR2YCRCB) ycrcb_px(b, g, r, c0, c1, c2);
        else if constexpr (CODE == VP_BGR2HLS) hls_px(b, g, r, c0, c1, c2);
        else hsv_px(s, b, g, r, c0, c1, c2);
        if (dst) { dst[3 * o] = (uint8_t)c0; dst[3 * o + 1] = (uint8_t)c1; dst[3 * o + 2] = (uint8_t)c2; }
        if (p0) p0[o] = (uint8_t)c0;
        if (p1) p1[o] = (uint8_t)c1;
        if (p2) p2[o] = (uint8_t)c2;
    }
}

__global__ __launch_bounds__(256) void k_gray2bgr(const uint8_t* __restrict__ src, size_t stride, int w, int h,
                                                  uint8_t* __restrict__ dst, uint8_t* p0, uint8_t* p1, uint8_t* p2)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const uint8_t v = src[(size_t)y * stride + x];
    const size_t o = (size_t)y * w + x;
    if (dst) { dst[3 * o] = v; dst[3 * o + 1] = v; dst[3 * o + 2] = v; }
    if (p0) p0[o] = v;
    if (p1) p1[o] = v;
    if (p2) p2[o] = v;
}

// Flat forms of the standalone conversions, on the pattern of k_color_thresh_flat: packed rows (stride == 3 w, so the image is one run
// of pixels), every pointer 16-B aligned; one lane = 16 px = three 16-B loads, and per requested output 16-B stores (interleaved
// image: three of them; a plane: one).  Only what is asked for is computed (NEED: bit c = channel c of the converted pixel is stored
// somewhere) and stored.  The one-pixel-per-thread kernels above stay for strided views, unaligned planes and the last npx % 16 px.
template <int CODE, int NEED>
__global__ __launch_bounds__(256) void k_cvt_color_flat(const uint8_t* __restrict__ src, size_t ngroups, vp_tables tab, uint8_t* __restrict__ dst,
                                                        uint8_t* __restrict__ p0, uint8_t* __restrict__ p1, uint8_t* __restrict__ p2)
{
    __shared__ typename ModeLds<CODE>::type s;
    load_lds<CODE>(s, tab);
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += step) {
        const uint4* p = reinterpret_cast<const uint4*>(src + g * 48);
        const uint4 v0 = p[0], v1 = p[1], v2 = p[2];
        const u32 in[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
        u32 a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0}, c[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int bb = BYTE_OF(in, 3 * k), gg = BYTE_OF(in, 3 * k + 1), rr = BYTE_OF(in, 3 * k + 2);
            int c0 = 0, c1 = 0, c2 = 0;
            if constexpr (CODE == VP_BGR2GRAY) c0 = gray_px(bb, gg, rr);
            else if constexpr (CODE == VP_BGR2LAB) lab_px<NEED>(s, bb, gg, rr, c0, c1, c2);
            else if constexpr (CODE == VP_BGR2YCRCB) ycrcb_px(bb, gg, rr, c0, c1, c2);
            else if constexpr (CODE == VP_BGR2HLS) hls_px(bb, gg, rr, c0, c1, c2);
            else hsv_px(s, bb, gg, rr, c0, c1, c2);
            a[k >> 2] |= (u32)c0 << (8 * (k & 3));
            b[k >> 2] |= (u32)c1 << (8 * (k & 3));
            c[k >> 2] |= (u32)c2 << (8 * (k & 3));
        }
        if constexpr (CODE == VP_BGR2GRAY) {
            if (dst) vp_store16(dst + g * 16, a[0], a[1], a[2], a[3]);
            if (p0) vp_store16(p0 + g * 16, a[0], a[1], a[2], a[3]);
        } else {
            if (dst) {
                u32 d[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    d[(3 * k) >> 2] |= BYTE_OF(a, k) << (8 * ((3 * k) & 3));
                    d[(3 * k + 1) >> 2] |= BYTE_OF(b, k) << (8 * ((3 * k + 1) & 3));
                    d[(3 * k + 2) >> 2] |= BYTE_OF(c, k) << (8 * ((3 * k + 2) & 3));
                }
                vp_store16(dst + g * 48, d[0], d[1], d[2], d[3]);
                vp_store16(dst + g * 48 + 16, d[4], d[5], d[6], d[7]);
                vp_store16(dst + g * 48 + 32, d[8], d[9], d[10], d[11]);
            }
            if (p0) vp_store16(p0 + g * 16, a[0], a[1], a[2], a[3]);
            if (p1) vp_store16(p1 + g * 16, b[0], b[1], b[2], b[3]);
            if (p2) vp_store16(p2 + g * 16, c[0], c[1], c[2], c[3]);
        }
    }
}

// lane = 16 px = one 16-B load; the interleaved image takes three 16-B stores of byte triples, the planes are copies
__global__ __launch_bounds__(256) void k_gray2bgr_flat(const uint8_t* __restrict__ src, size_t ngroups, uint8_t* __restrict__ dst, uint8_t* p0,
                                                       uint8_t* p1, uint8_t* p2)
{
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= ngroups) return;
    const uint4 v = reinterpret_cast<const uint4*>(src)[g];
    const u32 a[4] = {v.x, v.y, v.z, v.w};
    if (dst) {
        u32 d[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; k++) {
#pragma unroll
            for (int c = 0; c < 3; c++) d[(3 * k + c) >> 2] |= BYTE_OF(a, k) << (8 * ((3 * k + c) & 3));
        }
        vp_store16(dst + g * 48, d[0], d[1], d[2], d[3]);
        vp_store16(dst + g * 48 + 16, d[4], d[5], d[6], d[7]);
        vp_store16(dst + g * 48 + 32, d[8], d[9], d[10], d[11]);
    }
    if (p0) vp_store16(p0 + g * 16, a[0], a[1], a[2], a[3]);
    if (p1) vp_store16(p1 + g * 16, a[0], a[1], a[2], a[3]);
    if (p2) vp_store16(p2 + g * 16, a[0], a[1], a[2], a[3]);
}

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

template <int CODE>
static void launch_cvt_generic(vp_ctx* ctx, const uint8_t* d_src, size_t stride, int w, int h, uint8_t* d_dst, uint8_t* d_p0, uint8_t* d_p1, uint8_t* d_p2)
{
    hipLaunchKernelGGL((k_cvt_color<CODE>), dim3((unsigned)((w + 255) / 256), (unsigned)h), dim3(256), 0, ctx->stream, d_src, stride, w, h, ctx->tab, d_dst,
                       d_p0, d_p1, d_p2);
}

template <int CODE>
static void launch_cvt(vp_ctx* ctx, const uint8_t* d_src, size_t stride, int w, int h, uint8_t* d_dst, uint8_t* d_p0, uint8_t* d_p1, uint8_t* d_p2)
{
    const size_t npx = (size_t)w * h, ngroups = npx / 16;
    const bool flat = ctx->flat_ops && (stride == (size_t)w * 3 || h == 1) && ngroups > 0 && aligned16(d_src) && aligned16(d_dst) && aligned16(d_p0) &&
                      aligned16(d_p1) && aligned16(d_p2);
    if (!flat) { launch_cvt_generic<CODE>(ctx, d_src, stride, w, h, d_dst, d_p0, d_p1, d_p2); return; }
    const dim3 grid((unsigned)std::min<size_t>((ngroups + 255) / 256, (size_t)ctx->num_cu * 32));
    int need = d_dst ? 7 : ((d_p0 ? 1 : 0) | (d_p1 ? 2 : 0) | (d_p2 ? 4 : 0));
    if constexpr (CODE == VP_BGR2LAB) {
        switch (need) {       // every cube-root look-up that is not needed is a dependent LDS read less per pixel
            case 1: hipLaunchKernelGGL((k_cvt_color_flat<CODE, 1>), grid, dim3(256), 0, ctx->stream, d_src, ngroups, ctx->tab, d_dst, d_p0, d_p1, d_p2); break;
            case 2: hipLaunchKernelGGL((k_cvt_color_flat<CODE, 2>), grid, dim3(256), 0, ctx->stream, d_src, ngroups, ctx->tab, d_dst, d_p0, d_p1, d_p2); break;
            case 4: hipLaunchKernelGGL((k_cvt_color_flat<CODE, 4>), grid, dim3(256), 0, ctx->stream, d_src, ngroups, ctx->tab, d_dst, d_p0, d_p1, d_p2); break;
            default: hipLaunchKernelGGL((k_cvt_color_flat<CODE, 7>), grid, dim3(256), 0, ctx->stream, d_src, ngroups, ctx->tab, d_dst, d_p0, d_p1, d_p2); break;
        }
    } else {
        hipLaunchKernelGGL((k_cvt_color_flat<CODE, 7>), grid, dim3(256), 0, ctx->stream, d_src, ngroups, ctx->tab, d_dst, d_p0, d_p1, d_p2);
    }
    const size_t done = ngroups * 16;
    if (done < npx) {   // the last npx % 16 pixels, as a one-row image
        const int dcn = CODE == VP_BGR2GRAY ? 1 : 3;
        launch_cvt_generic<CODE>(ctx, d_src + done * 3, 0, (int)(npx - done), 1, d_dst ? d_dst + done * dcn : nullptr, d_p0 ? d_p0 + done : nullptr,
                                 d_p1 ? d_p1 + done : nullptr, d_p2 ? d_p2 + done : nullptr);
    }
}

int vpk_cvt_color(vp_ctx* ctx, int code, const uint8_t* d_src, size_t stride, int w, int h, uint8_t* d_dst, uint8_t* d_p0,
                  uint8_t* d_p1, uint8_t* d_p2)
{
    switch (code) {
        case VP_BGR2LAB: launch_cvt<VP_BGR2LAB>(ctx, d_src, stride, w, h, d_dst, d_p0, d_p1, d_p2); break;
        case VP_BGR2HSV: launch_cvt<VP_BGR2HSV>(ctx, d_src, stride, w, h, d_dst, d_p0, d_p1, d_p2); break;
        case VP_BGR2GRAY: launch_cvt<VP_BGR2GRAY>(ctx, d_src, stride, w, h, d_dst, d_p0, d_p1, d_p2); break;
        case VP_BGR2YCRCB: launch_cvt<VP_BGR2YCRCB>(ctx, d_src, stride, w, h, d_dst, d_p0, d_p1, d_p2); break;
        case VP_BGR2HLS: launch_cvt<VP_BGR2HLS>(ctx, d_src, stride, w, h, d_dst, d_p0, d_p1, d_p2); break;
        case VP_GRAY2BGR: {
            const size_t npx = (size_t)w * h, ngroups = npx / 16;
            const bool flat = ctx->flat_ops && (stride == (size_t)w || h == 1) && ngroups > 0 && aligned16(d_src) && aligned16(d_dst) && aligned16(d_p0) &&
                              aligned16(d_p1) && aligned16(d_p2);
            if (!flat) {
                hipLaunchKernelGGL(k_gray2bgr, dim3((unsigned)((w + 255) / 256), (unsigned)h), dim3(256), 0, ctx->stream, d_src, stride, w, h, d_dst, d_p0, d_p1, d_p2);
                break;
            }
            hipLaunchKernelGGL(k_gray2bgr_flat, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, ctx->stream, d_src, ngroups, d_dst, d_p0, d_p1, d_p2);
            const size_t done = ngroups * 16;
            if (done < npx)
                hipLaunchKernelGGL(k_gray2bgr, dim3(1, 1), dim3(256), 0, ctx->stream, d_src + done, (size_t)0, (int)(npx - done), 1, d_dst ? d_dst + done * 3 : nullptr,
                                   d_p0 ? d_p0 + done : nullptr, d_p1 ? d_p1 + done : nullptr, d_p2 ? d_p2 + done : nullptr);
            break;
        }
        default: return vp_fail(ctx, VP_ERR_INVALID, "conversion code");
    }
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// ---- standalone inRange -----------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_inrange_u8(const uint8_t* __restrict__ src, size_t stride, int w, int h, int cn,
                                                    vp_range3 q, uint8_t* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* p = src + (size_t)y * stride + (size_t)x * cn;
    bool ok = true;
    for (int c = 0; c < cn; c++) ok = ok & (p[c] >= q.lo[c]) & (p[c] <= q.hi[c]);
    dst[(size_t)y * w + x] = ok ? 255 : 0;
}

__global__ __launch_bounds__(256) void k_inrange_f32(const float* __restrict__ src, size_t stride_bytes, int w, int h, float lo,
                                                     float hi, uint8_t* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const float v = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(src) + (size_t)y * stride_bytes)[x];
    dst[(size_t)y * w + x] = (v >= lo && v <= hi) ? 255 : 0;
}

// Flat form: packed rows, 16-B aligned; lane = 16 px = CN 16-B loads and one 16-B store.  Per channel one unsigned comparison
// (in_span); ranges are kernel arguments, so an empty one costs nothing per pixel.
// WBITS: also the bit-packed mask (w % 64 == 0, so the four 16-px groups of a word are four neighbouring lanes that leave together):
// what the contour pass and the bit-plane morphology would otherwise make with a launch of their own (k_pack_bits).
template <int CN, bool WBITS>
__global__ __launch_bounds__(256) void k_inrange_u8_flat(const uint8_t* __restrict__ src, size_t ngroups, vp_range3 q, uint8_t* __restrict__ dst,
                                                         u64* __restrict__ bits)
{
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= ngroups) return;
    u32 in[4 * CN];
#pragma unroll
    for (int j = 0; j < CN; j++) {
        const uint4 v = reinterpret_cast<const uint4*>(src + g * 16 * CN)[j];
        in[4 * j] = v.x; in[4 * j + 1] = v.y; in[4 * j + 2] = v.z; in[4 * j + 3] = v.w;
    }
    u32 out[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        bool ok = true;
#pragma unroll
        for (int c = 0; c < CN; c++) ok = ok & in_span((int)BYTE_OF(in, CN * k + c), q.lo[c], q.hi[c]);
        out[k >> 2] |= (ok ? 0xffu : 0u) << (8 * (k & 3));
    }
    vp_store16(dst + g * 16, out[0], out[1], out[2], out[3]);
    if constexpr (WBITS) {
        u32 m = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) m |= ((out[k >> 2] >> (8 * (k & 3))) & 1u) << k;
        u64 wv = (u64)m << (16 * (threadIdx.x & 3));
        wv |= __shfl_xor(wv, 1);
        wv |= __shfl_xor(wv, 2);
        if ((threadIdx.x & 3) == 0) bits[g >> 2] = wv;
    }
}

// d_bits (nullable): the bit-packed mask as well, when the flat form runs and w % 64 == 0; *made_bits tells whether it was written
int vpk_inrange_u8(vp_ctx* ctx, const uint8_t* d_src, size_t stride, int w, int h, int cn, const vp_range3& q, uint8_t* d_dst, u64* d_bits, int* made_bits)
{
    const size_t npx = (size_t)w * h, ngroups = npx / 16;
    const bool flat = ctx->flat_ops && (cn == 1 || cn == 3) && (stride == (size_t)w * cn || h == 1) && ngroups > 0 && aligned16(d_src) && aligned16(d_dst);
    const bool wbits = flat && d_bits && (w % 64) == 0;
    if (made_bits) *made_bits = wbits ? 1 : 0;
    if (flat) {
        const dim3 grid((unsigned)((ngroups + 255) / 256));
        if (cn == 1 && wbits) hipLaunchKernelGGL((k_inrange_u8_flat<1, true>), grid, dim3(256), 0, ctx->stream, d_src, ngroups, q, d_dst, d_bits);
        else if (cn == 1) hipLaunchKernelGGL((k_inrange_u8_flat<1, false>), grid, dim3(256), 0, ctx->stream, d_src, ngroups, q, d_dst, (u64*)nullptr);
        else if (wbits) hipLaunchKernelGGL((k_inrange_u8_flat<3, true>), grid, dim3(256), 0, ctx->stream, d_src, ngroups, q, d_dst, d_bits);
        else hipLaunchKernelGGL((k_inrange_u8_flat<3, false>), grid, dim3(256), 0, ctx->stream, d_src, ngroups, q, d_dst, (u64*)nullptr);
        const size_t done = ngroups * 16;
        if (done < npx)
            hipLaunchKernelGGL(k_inrange_u8, dim3(1, 1), dim3(256), 0, ctx->stream, d_src + done * cn, (size_t)0, (int)(npx - done), 1, cn, q, d_dst + done);
    } else {
        dim3 grid((unsigned)((w + 255) / 256), (unsigned)h);
        hipLaunchKernelGGL(k_inrange_u8, grid, dim3(256), 0, ctx->stream, d_src, stride, w, h, cn, q, d_dst);
    }
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

int vpk_inrange_f32(vp_ctx* ctx, const float* d_src, size_t stride_bytes, int w, int h, float lo, float hi, uint8_t* d_dst)
{
    dim3 grid((unsigned)((w + 255) / 256), (unsigned)h);
    hipLaunchKernelGGL(k_inrange_f32, grid, dim3(256), 0, ctx->stream, d_src, stride_bytes, w, h, lo, hi, d_dst);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// ---- colour distance (float32, numpy order of operations; built with -ffp-contract=off) ---------

struct cd_params { float color[3]; float wts[3]; int skipmask; };

__global__ __launch_bounds__(256) void k_color_distance(const uint8_t* __restrict__ p0, const uint8_t* __restrict__ p1,
                                                        const uint8_t* __restrict__ p2, size_t npx, cd_params prm,
                                                        float* __restrict__ d2, uint8_t* __restrict__ sq)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx) return;
    const uint8_t* p[3] = {p0, p1, p2};
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (prm.skipmask & (1 << c)) continue;
        const float t = __fsub_rn((float)p[c][i], prm.color[c]);
        const float s2 = __fmul_rn(t, t);
        const float term = __fmul_rn(prm.wts[c], s2);
        acc = __fadd_rn(acc, term);
    }
    if (d2) d2[i] = acc;
    if (sq) sq[i] = (uint8_t)(int)__fsqrt_rn(acc);
}

int vpk_color_distance(vp_ctx* ctx, const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, size_t npx, const float* color,
                       const float* wts, int skipmask, float* d2, uint8_t* sq)
{
    cd_params prm;
    for (int c = 0; c < 3; c++) { prm.color[c] = color[c]; prm.wts[c] = wts[c]; }
    prm.skipmask = skipmask;
    // a skipped channel may carry a NULL plane; point it somewhere valid (never dereferenced)
    const uint8_t* any = p0 ? p0 : (p1 ? p1 : p2);
    hipLaunchKernelGGL(k_color_distance, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, ctx->stream, p0 ? p0 : any,
                       p1 ? p1 : any, p2 ? p2 : any, npx, prm, d2, sq);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}


// ---- float32 CIE L*a*b* (north star: "LAB floats match within 1e-4") -------------------------------------------------
// Analytic conversion of BGR float32 in [0,1] (sRGB gamma, D65, the same matrix/white point as the 8-bit path):
// L in [0,100], a/b in about [-127,127].  The reference never converts float images (every LAB call site is on u8,
// SURVEY A1), so this is an extension; it is checked against float64 arithmetic at 1e-4, not against cv2's
// interpolated-LUT float path.
__global__ __launch_bounds__(256) void k_bgr2lab_f32(const float* __restrict__ src, size_t npx, float* __restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx) return;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float v = src[3 * i + k];
        c[k] = v <= 0.04045f ? v * (1.0f / 12.92f) : powf((v + 0.055f) * (1.0f / 1.055f), 2.4f);
    }
    const float B = c[0], G = c[1], R = c[2];
    float X = (0.412453f * R + 0.357580f * G + 0.180423f * B) * (1.0f / 0.950456f);
    float Y = 0.212671f * R + 0.715160f * G + 0.072169f * B;
    float Z = (0.019334f * R + 0.119193f * G + 0.950227f * B) * (1.0f / 1.088754f);
    const float thr = 216.0f / 24389.0f, sl = 841.0f / 108.0f, bi = 16.0f / 116.0f;
    const float fx = X > thr ? cbrtf(X) : sl * X + bi;
    const float fy = Y > thr ? cbrtf(Y) : sl * Y + bi;
    const float fz = Z > thr ? cbrtf(Z) : sl * Z + bi;
    dst[3 * i] = 116.0f * fy - 16.0f;
    dst[3 * i + 1] = 500.0f * (fx - fy);
    dst[3 * i + 2] = 200.0f * (fy - fz);
}

int vpk_bgr2lab_f32(vp_ctx* ctx, const float* d_src, size_t npx, float* d_dst)
{
    hipLaunchKernelGGL(k_bgr2lab_f32, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, ctx->stream, d_src, npx, d_dst);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}


// ---- order statistics of a float32 image (np.percentile in thresh_color_distance, utils/color.py:98) ------------------
// Exact k-th smallest by 4 rounds of 8-bit radix selection on order-preserving keys; each round is one histogram
// kernel (LDS bins per block, then 256 global atomics) and a 1 KB read-back.
__device__ __forceinline__ u32 f32_key(float f)
{
    const u32 b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__global__ __launch_bounds__(256) void k_radix_hist(const float* __restrict__ src, size_t n, u32 prefix, u32 prefix_mask, int shift,
                                                    u32* __restrict__ hist)
{
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const u32 k = f32_key(src[i]);
        if ((k & prefix_mask) == prefix) atomicAdd(&h[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(hist + threadIdx.x, h[threadIdx.x]);
}

int vpk_kth_f32(vp_ctx* ctx, const float* d_src, size_t n, size_t k, u32* d_hist, float* out)
{
    if (k >= n) return vp_fail(ctx, VP_ERR_INVALID, "order statistic index");
    u32 prefix = 0, mask = 0;
    size_t rank = k;
    u32 host_hist[256];
    size_t blocks = (n + 255) / 256;
    if (blocks > (size_t)ctx->num_cu * 8) blocks = (size_t)ctx->num_cu * 8;
    for (int shift = 24; shift >= 0; shift -= 8) {
        VP_HIP(ctx, hipMemsetAsync(d_hist, 0, 1024, ctx->stream));
        hipLaunchKernelGGL(k_radix_hist, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_src, n, prefix, mask, shift, d_hist);
        VP_HIP(ctx, hipMemcpyAsync(host_hist, d_hist, 1024, hipMemcpyDeviceToHost, ctx->stream));
        VP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        int b = 0;
        for (; b < 256; b++) {
            if (rank < host_hist[b]) break;
            rank -= host_hist[b];
        }
        if (b == 256) return vp_fail(ctx, VP_ERR_HIP, "radix select lost its element");
        prefix |= (u32)b << shift;
        mask |= 0xffu << shift;
    }
    const u32 bits = (prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix;
    memcpy(out, &bits, 4);
    return VP_OK;
}
