// cv2.GaussianBlur on 8-bit images (modules/preprocessor.py:110-114 `PPX_gaussian_blur`, utils/transform.py `simple_gaussian_blur`).
//
// OpenCV >= 4.0 filters CV_8U images on a bit-exact fixed-point path (imgproc/src/smooth.dispatch.cpp, GaussianBlurFixedPoint):
// taps in 8.8 fixed point that sum to exactly 256 (error diffusion towards the centre tap), a horizontal pass whose 8.8 sums are
// exact, a vertical pass whose 16.16 sums are rounded half up to 8 bits, BORDER_REFLECT_101.  The taps are made on the host
// (vp_gaussian_taps, double precision); the two passes below are integer arithmetic, so the result does not depend on the order
// of the additions.
#include "vp_internal.h"
#include <cmath>

#define GB_MAX_TAPS 511
#define GB_TILE 1024     // output bytes per block in the horizontal pass

__device__ __forceinline__ int gb_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// grid (ceil(w*cn / GB_TILE), h), 256 threads.  A block stages its piece of the row (+ halo, already reflected) in LDS.
__global__ __launch_bounds__(256) void k_gauss_h(const uint8_t* __restrict__ src, int w, int cn, const uint16_t* __restrict__ taps, int kw,
                                                 uint16_t* __restrict__ tmp)
{
    extern __shared__ uint8_t gb_lds[];
    __shared__ uint16_t tp[GB_MAX_TAPS + 1];
    const int y = blockIdx.y, rowbytes = w * cn, r = kw / 2;
    const int b0 = blockIdx.x * GB_TILE;                    // first output byte of this block
    const int nb = min(GB_TILE, rowbytes - b0);
    const int x0 = b0 / cn;                                 // first pixel touched
    const int x1 = (b0 + nb - 1) / cn;                      // last pixel touched
    const int npx = x1 - x0 + 1 + 2 * r;                    // staged pixels
    const uint8_t* row = src + (size_t)y * rowbytes;
    for (int i = threadIdx.x; i < kw; i += 256) tp[i] = taps[i];
    for (int i = threadIdx.x; i < npx * cn; i += 256) {
        const int px = i / cn, c = i - px * cn;
        gb_lds[i] = row[(size_t)gb_reflect101(x0 - r + px, w) * cn + c];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < nb; o += 256) {
        const int b = b0 + o, x = b / cn, c = b - x * cn;
        const uint8_t* p = gb_lds + (size_t)(x - x0) * cn + c;   // tap 0 sits at pixel x - r = staged pixel (x - x0)
        u32 s = 0;
        for (int k = 0; k < kw; k++) s += (u32)tp[k] * p[(size_t)k * cn];
        tmp[(size_t)y * rowbytes + b] = (uint16_t)s;
    }
}

// grid (ceil(w*cn / 256), h), 256 threads: thread = one output byte
__global__ __launch_bounds__(256) void k_gauss_v(const uint16_t* __restrict__ tmp, int rowbytes, int h, const uint16_t* __restrict__ taps, int kh,
                                                 uint8_t* __restrict__ dst)
{
    __shared__ uint16_t tp[GB_MAX_TAPS + 1];
    for (int i = threadIdx.x; i < kh; i += 256) tp[i] = taps[i];
    __syncthreads();
    const int b = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, r = kh / 2;
    if (b >= rowbytes) return;
    u32 s = 0;
    if (y - r >= 0 && y + r < h) {
        const uint16_t* p = tmp + (size_t)(y - r) * rowbytes + b;
        for (int k = 0; k < kh; k++) s += (u32)tp[k] * p[(size_t)k * rowbytes];
    } else {
        for (int k = 0; k < kh; k++) s += (u32)tp[k] * tmp[(size_t)gb_reflect101(y + k - r, h) * rowbytes + b];
    }
    const u32 v = (s + (1u << 15)) >> 16;
    dst[(size_t)y * rowbytes + b] = (uint8_t)(v > 255u ? 255u : v);
}

// getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED (8 fraction bits): n odd, 1 <= n <= GB_MAX_TAPS
void vp_gaussian_taps(int n, double sigma, uint16_t* out)
{
    static const double t3[] = {0.25, 0.5, 0.25}, t5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625},
                        t7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125},
                        t9[] = {4 / 256., 13 / 256., 30 / 256., 51 / 256., 60 / 256., 51 / 256., 30 / 256., 13 / 256., 4 / 256.};
    if (n == 1) { out[0] = 256; return; }
    double k[GB_MAX_TAPS + 1];
    const double* tab = nullptr;
    if (sigma <= 0) tab = n == 3 ? t3 : n == 5 ? t5 : n == 7 ? t7 : n == 9 ? t9 : nullptr;
    if (tab) {
        for (int i = 0; i < n; i++) k[i] = tab[i];
    } else {
        const double sx = sigma > 0 ? sigma : (double)n * 0.15 + 0.35;
        const double scale2x = -0.125 / (sx * sx);
        const int n2 = (n - 1) / 2;
        double sum = 0;
        for (int i = 0, x = 1 - n; i < n2; i++, x += 2) { k[i] = std::exp((double)(x * x) * scale2x); sum += k[i]; }
        sum = sum * 2 + 1;
        const double mul1 = 1.0 / sum;
        for (int i = 0; i < n2; i++) { k[i] *= mul1; k[n - 1 - i] = k[i]; }
        k[n2] = mul1;
    }
    const int n2 = n / 2;
    double err = 0;
    long long sum = 0;
    for (int i = 0; i < n2; i++) {
        const double adj = k[i] * 256.0 + err;
        const long long v0 = std::llrint(adj);
        err = adj - (double)v0;
        out[i] = out[n - 1 - i] = (uint16_t)v0;
        sum += 2 * v0;
    }
    out[n2] = (uint16_t)(256 - sum);
}

// d_taps: kw + kh uint16 (x taps then y taps); d_tmp: w*h*cn uint16
int vpk_gaussian_blur(vp_ctx* ctx, const uint8_t* d_src, int w, int h, int cn, const uint16_t* d_taps, int kw, int kh, uint16_t* d_tmp, uint8_t* d_dst)
{
    const int rowbytes = w * cn;
    vp_prof_scope ps(ctx, VPK_OTHER);
    const size_t lds = (size_t)(GB_TILE / cn + 2 + 2 * (kw / 2)) * cn + 16;
    hipLaunchKernelGGL(k_gauss_h, dim3((unsigned)((rowbytes + GB_TILE - 1) / GB_TILE), (unsigned)h), dim3(256), lds, ctx->stream, d_src, w, cn, d_taps, kw, d_tmp);
    hipLaunchKernelGGL(k_gauss_v, dim3((unsigned)((rowbytes + 255) / 256), (unsigned)h), dim3(256), 0, ctx->stream, d_tmp, rowbytes, h, d_taps + kw, kh, d_dst);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// cv2.threshold on 8-bit data (utils/color.py:124-199 binary_threshold / binary_threshold_inv / max_threshold / above_threshold /
// below_threshold): imgproc/src/thresh.cpp compares against ithresh = floor(thresh); imaxval = saturate(round(maxval)).
// type: 0 BINARY, 1 BINARY_INV, 2 TRUNC, 3 TOZERO, 4 TOZERO_INV
__global__ __launch_bounds__(256) void k_threshold_u8(const uint8_t* __restrict__ src, size_t n, int ithresh, int imaxval, int type, uint8_t* __restrict__ dst)
{
    const int tr = ithresh < 0 ? 0 : (ithresh > 255 ? 255 : ithresh);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int v = src[i];
        const bool above = v > ithresh;
        int o;
        switch (type) {
        case 0: o = above ? imaxval : 0; break;
        case 1: o = above ? 0 : imaxval; break;
        case 2: o = above ? tr : v; break;
        case 3: o = above ? v : 0; break;
        default: o = above ? 0 : v; break;
        }
        dst[i] = (uint8_t)o;
    }
}

int vpk_threshold_u8(vp_ctx* ctx, const uint8_t* d_src, size_t n, int ithresh, int imaxval, int type, uint8_t* d_dst)
{
    vp_prof_scope ps(ctx, VPK_OTHER);
    const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, (size_t)ctx->num_cu * 16));
    hipLaunchKernelGGL(k_threshold_u8, dim3(blocks), dim3(256), 0, ctx->stream, d_src, n, ithresh, imaxval, type, d_dst);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// 256-bin histogram of a byte plane (Otsu's threshold, imgproc/src/thresh.cpp getThreshVal_Otsu_8u, needs nothing else)
__global__ __launch_bounds__(256) void k_hist_u8(const uint8_t* __restrict__ src, size_t n, u32* __restrict__ hist)
{
    __shared__ u32 lh[256 * 16];   // 16 interleaved copies against same-bin atomics
    const int cp = threadIdx.x & 15;
    for (int i = threadIdx.x; i < 256 * 16; i += 256) lh[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) atomicAdd(&lh[src[i] * 16 + cp], 1u);
    __syncthreads();
    u32 v = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) v += lh[threadIdx.x * 16 + k];
    if (v) atomicAdd(hist + threadIdx.x, v);
}

int vpk_hist_u8(vp_ctx* ctx, const uint8_t* d_src, size_t n, u32* d_hist)
{
    vp_prof_scope ps(ctx, VPK_OTHER);
    VP_HIP(ctx, hipMemsetAsync(d_hist, 0, 1024, ctx->stream));
    const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, (size_t)ctx->num_cu * 8));
    hipLaunchKernelGGL(k_hist_u8, dim3(blocks), dim3(256), 0, ctx->stream, d_src, n, d_hist);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// ---- cv2.Canny(image, t1, t2), 3x3 aperture, L1 gradient (utils/feature.py:43-101) -----------------------------------------------
// canny.cpp in four data-parallel steps: (1) Sobel derivatives with replicated borders, the channel with the largest |dx| + |dy| wins
// (first on ties); (2) non-maximum suppression with OpenCV's integer direction test (TG22 = 13573 in Q15) against the neighbours'
// magnitudes, 0 outside the image, giving a bit plane of survivors and one of survivors above the high threshold; (3) hysteresis =
// 8-connected components of the survivor plane (the labelling kernels of vp_ccl.hip) that hold a pixel of the strong plane — the
// result of OpenCV's stack flood does not depend on its visiting order; (4) 255 where a survivor's component is marked.
__global__ __launch_bounds__(256) void k_canny_grad(const uint8_t* __restrict__ src, int w, int h, int cn, uint16_t* __restrict__ mag,
                                                    short2* __restrict__ grad)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int xm = max(x - 1, 0), xp = min(x + 1, w - 1), ym = max(y - 1, 0), yp = min(y + 1, h - 1);
    const uint8_t* r0 = src + (size_t)ym * w * cn;
    const uint8_t* r1 = src + (size_t)y * w * cn;
    const uint8_t* r2 = src + (size_t)yp * w * cn;
    int bm = -1, bdx = 0, bdy = 0;
    for (int c = 0; c < cn; c++) {
        const int a00 = r0[xm * cn + c], a01 = r0[x * cn + c], a02 = r0[xp * cn + c];
        const int a10 = r1[xm * cn + c], a12 = r1[xp * cn + c];
        const int a20 = r2[xm * cn + c], a21 = r2[x * cn + c], a22 = r2[xp * cn + c];
        const int dx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
        const int dy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
        const int m = abs(dx) + abs(dy);
        if (m > bm) { bm = m; bdx = dx; bdy = dy; }
    }
    mag[(size_t)y * w + x] = (uint16_t)bm;              // <= 2040
    grad[(size_t)y * w + x] = make_short2((short)bdx, (short)bdy);
}

// block = 256 consecutive pixels of a row; a wave's ballot is one 64-pixel word of each plane
__global__ __launch_bounds__(256) void k_canny_nms(const uint16_t* __restrict__ mag, const short2* __restrict__ grad, int w, int h, int ww, int low,
                                                   int high, u64* __restrict__ cand, u64* __restrict__ strong)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    bool keep = false, hi = false;
    if (x < w) {
        const uint16_t* a = mag + (size_t)y * w + x;
        const int m = a[0];
        if (m > low) {
            const short2 g = grad[(size_t)y * w + x];
            const int xs = g.x, ys = g.y;
            const int ax = abs(xs), ay = abs(ys) << 15;
            const int tg22x = ax * 13573;
            auto at = [&](int dy, int dx) -> int {
                const int xx = x + dx, yy = y + dy;
                return (xx >= 0 && xx < w && yy >= 0 && yy < h) ? (int)mag[(size_t)yy * w + xx] : 0;
            };
            if (ay < tg22x) keep = m > at(0, -1) && m >= at(0, 1);
            else if (ay > tg22x + (ax << 16)) keep = m > at(-1, 0) && m >= at(1, 0);
            else {
                const int s = (xs ^ ys) < 0 ? -1 : 1;
                keep = m > at(-1, -s) && m > at(1, s);
            }
            hi = keep && m > high;
        }
    }
    const u64 bc = __ballot(keep), bs = __ballot(hi);
    const int word = x >> 6;
    if ((threadIdx.x & 63) == 0 && word < ww) {
        cand[(size_t)y * ww + word] = bc;
        strong[(size_t)y * ww + word] = bs;
    }
}

// one thread per word of the strong plane: marks the components that hold an edge seed
__global__ __launch_bounds__(256) void k_canny_mark(const u64* __restrict__ strong, const int32_t* __restrict__ labels, int w, int h, int ww,
                                                    uint8_t* __restrict__ seen)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)h * ww) return;
    u64 s = strong[i];
    const int y = (int)(i / ww), j = (int)(i - (size_t)y * ww);
    while (s) {
        const int b = __ffsll((long long)s) - 1;
        s &= s - 1;
        seen[labels[(size_t)y * w + 64 * j + b]] = 1;
    }
}

__global__ __launch_bounds__(256) void k_canny_out(const int32_t* __restrict__ labels, const uint8_t* __restrict__ seen, size_t n, uint8_t* __restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int l = labels[i];
    dst[i] = (l > 0 && seen[l]) ? 255 : 0;
}

size_t vp_canny_ws_bytes(int w, int h)
{
    const size_t npx = (size_t)w * h, bitbytes = (size_t)h * vp_ww(w) * 8;
    return vp_align(npx * 2) + vp_align(npx * 4) + 2 * vp_align(bitbytes) + vp_align(npx * 4) + vp_align(vp_ccl_nids(w, h) + 2) + vp_align(4) +
           vp_ccl_ws_bytes(w, h, 1, 1) + 4096;
}

int vpk_canny_u8(vp_ctx* ctx, const uint8_t* d_src, int w, int h, int cn, int low, int high, uint8_t* d_dst)
{
    const size_t npx = (size_t)w * h;
    const int ww = vp_ww(w);
    const size_t bitbytes = (size_t)h * ww * 8;
    uint16_t* d_mag = (uint16_t*)vp_ws_take(ctx, npx * 2);
    short2* d_grad = (short2*)vp_ws_take(ctx, npx * 4);
    u64* d_cand = (u64*)vp_ws_take(ctx, bitbytes);
    u64* d_strong = (u64*)vp_ws_take(ctx, bitbytes);
    int32_t* d_labels = (int32_t*)vp_ws_take(ctx, npx * 4);
    const size_t nseen = vp_ccl_nids(w, h) + 2;          // labels are 1 .. number of components <= number of segments
    uint8_t* d_seen = (uint8_t*)vp_ws_take(ctx, nseen);
    int32_t* d_nl = (int32_t*)vp_ws_take(ctx, 4);
    vp_ccl_ws ws;
    vp_ccl_ws_carve(ctx, w, h, 1, 1, &ws);
    if (!d_mag || !d_grad || !d_cand || !d_strong || !d_labels || !d_seen || !d_nl || !vp_ccl_ws_ok(ws))
        return vp_fail(ctx, VP_ERR_NOMEM, "canny workspace");
    hipStream_t s = ctx->stream;
    const dim3 grid((unsigned)((w + 255) / 256), (unsigned)h);
    {
        vp_prof_scope ps(ctx, VPK_OTHER);
        hipLaunchKernelGGL(k_canny_grad, grid, dim3(256), 0, s, d_src, w, h, cn, d_mag, d_grad);
        hipLaunchKernelGGL(k_canny_nms, grid, dim3(256), 0, s, d_mag, d_grad, w, h, ww, low, high, d_cand, d_strong);
        VP_HIP(ctx, hipMemsetAsync(d_seen, 0, nseen, s));
    }
    const int rc = vpk_ccl(ctx, d_cand, w, h, 1, VP_CCL_PIXEL, ws, d_labels, nullptr, nullptr, 1, d_nl);
    if (rc != VP_OK) return rc;
    {
        vp_prof_scope ps(ctx, VPK_OTHER);
        hipLaunchKernelGGL(k_canny_mark, dim3((unsigned)(((size_t)h * ww + 255) / 256)), dim3(256), 0, s, d_strong, d_labels, w, h, ww, d_seen);
        hipLaunchKernelGGL(k_canny_out, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s, d_labels, d_seen, npx, d_dst);
    }
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// ---- cv2.adaptiveThreshold(src, maxValue, ADAPTIVE_THRESH_MEAN_C, type, blockSize, C) (utils/color.py:220-254) ------------------------
// mean = exact nearest integer of the block sum / blockSize^2 over a window with replicated borders (what all of OpenCV's box-filter
// roundings give for odd block sizes up to 151), then src - mean > -ceil(C) (BINARY) or src - mean <= -floor(C) (BINARY_INV).
#define AT_TILE 1024
__global__ __launch_bounds__(256) void k_box_h(const uint8_t* __restrict__ src, int w, int r, uint16_t* __restrict__ tmp)
{
    extern __shared__ uint8_t at_lds[];
    const int y = blockIdx.y, x0 = blockIdx.x * AT_TILE;
    const int nx = min(AT_TILE, w - x0);
    const uint8_t* row = src + (size_t)y * w;
    for (int i = threadIdx.x; i < nx + 2 * r; i += 256) at_lds[i] = row[min(max(x0 - r + i, 0), w - 1)];
    __syncthreads();
    for (int o = threadIdx.x; o < nx; o += 256) {
        u32 s = 0;
        for (int k = 0; k <= 2 * r; k++) s += at_lds[o + k];
        tmp[(size_t)y * w + x0 + o] = (uint16_t)s;
    }
}
__global__ __launch_bounds__(256) void k_box_v_adaptive(const uint8_t* __restrict__ src, const uint16_t* __restrict__ tmp, int w, int h, int r, int imax,
                                                        int idelta, int inv, uint8_t* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    u32 s = 0;
    for (int k = -r; k <= r; k++) s += tmp[(size_t)min(max(y + k, 0), h - 1) * w + x];
    const u32 d = (u32)(2 * r + 1) * (u32)(2 * r + 1);
    const int mean = (int)((2ull * s + d) / (2ull * d));
    const int diff = (int)src[(size_t)y * w + x] - mean;
    const bool on = inv ? diff <= -idelta : diff > -idelta;
    dst[(size_t)y * w + x] = (uint8_t)(on ? imax : 0);
}

// d_tmp: w*h uint16
int vpk_adaptive_threshold_mean(vp_ctx* ctx, const uint8_t* d_src, int w, int h, int imax, int idelta, int inv, int block, uint16_t* d_tmp, uint8_t* d_dst)
{
    const int r = block / 2;
    vp_prof_scope ps(ctx, VPK_OTHER);
    hipLaunchKernelGGL(k_box_h, dim3((unsigned)((w + AT_TILE - 1) / AT_TILE), (unsigned)h), dim3(256), (size_t)AT_TILE + 2 * r + 16, ctx->stream, d_src, w, r, d_tmp);
    hipLaunchKernelGGL(k_box_v_adaptive, dim3((unsigned)((w + 255) / 256), (unsigned)h), dim3(256), 0, ctx->stream, d_src, d_tmp, w, h, r, imax, idelta, inv, d_dst);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}
