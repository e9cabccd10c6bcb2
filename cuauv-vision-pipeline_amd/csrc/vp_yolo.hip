// Pre- and post-processing around a detector (BASELINE config 5): letterbox resize and non-maximum suppression.
//
// The reference's modules/yolo.py:112 calls `self.model.track(image)`; everything between the camera frame and the handler record
// (handlers/torpedoes.py:76-82 reads x1..y4 + confidence) happens inside ultralytics, which is neither in the reference tree nor
// installed here.  These kernels follow the published algorithms of that dependency (LetterBox: scale to fit, centre, pad 114,
// BGR->RGB, HWC->CHW, /255; NMS: greedy IoU suppression in score order; rotated boxes: the probabilistic IoU of Gaussian box
// models with the "worse than some higher-scored box" rule) and are checked against plain PyTorch fp32 restatements in
// tests/test_gpu_yolo.py - parity unpinned, like the rest of that row.
#include "vp_internal.h"

// cv2.resize(INTER_LINEAR) on 8-bit data as OpenCV's generic path computes it (imgproc/src/resize.cpp): 11-bit horizontal and
// vertical coefficients (cvRound(f * 2048)), half-pixel centres, edge replication, and the two-stage rounding
// (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
struct lb_params {
    int sw, sh;          // source size
    int dw, dh;          // destination (network input) size
    int nw, nh;          // resized content size
    int left, top;       // content offset inside the destination
    float scale_x, scale_y;   // source pixels per content pixel
    int pad;
};

__device__ __forceinline__ void lb_coef(int d, float scale, int ssize, int& s0, int& a0, int& a1)
{
    float f = (float)(((double)d + 0.5) * (double)scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= ssize - 1) { s = ssize - 1; f = 0.f; }
    s0 = s;
    a1 = (int)rintf(f * 2048.f);
    a0 = (int)rintf((1.f - f) * 2048.f);
}

// grid (ceil(dw/64), dh), 64 threads: thread = one destination pixel, three planes written (R, G, B order)
__global__ __launch_bounds__(64) void k_letterbox(const uint8_t* __restrict__ src, lb_params P, float* __restrict__ dst)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y;
    if (x >= P.dw) return;
    const size_t plane = (size_t)P.dw * P.dh;
    float* o = dst + (size_t)y * P.dw + x;
    const int cx = x - P.left, cy = y - P.top;
    if (cx < 0 || cx >= P.nw || cy < 0 || cy >= P.nh) {
        const float v = (float)P.pad / 255.f;
        o[0] = v; o[plane] = v; o[2 * plane] = v;
        return;
    }
    int bgr[3];
    if (P.nw == P.sw && P.nh == P.sh) {
        const uint8_t* p = src + ((size_t)cy * P.sw + cx) * 3;
        bgr[0] = p[0]; bgr[1] = p[1]; bgr[2] = p[2];
    } else if (P.sw == 2 * P.nw && P.sh == 2 * P.nh) {
        // cv2.resize turns INTER_LINEAR into the 2x2 box average at an exact halving (imgproc/src/resize.cpp)
        const uint8_t* r0 = src + ((size_t)(2 * cy) * P.sw + 2 * cx) * 3;
        const uint8_t* r1 = r0 + (size_t)P.sw * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) bgr[c] = (r0[c] + r0[3 + c] + r1[c] + r1[3 + c] + 2) >> 2;
    } else {
        int sx, ax0, ax1, sy, ay0, ay1;
        lb_coef(cx, P.scale_x, P.sw, sx, ax0, ax1);
        lb_coef(cy, P.scale_y, P.sh, sy, ay0, ay1);
        const int sx1 = min(sx + 1, P.sw - 1), sy1 = min(sy + 1, P.sh - 1);
        const uint8_t* r0 = src + (size_t)sy * P.sw * 3;
        const uint8_t* r1 = src + (size_t)sy1 * P.sw * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int S0 = r0[sx * 3 + c] * ax0 + r0[sx1 * 3 + c] * ax1;
            const int S1 = r1[sx * 3 + c] * ax0 + r1[sx1 * 3 + c] * ax1;
            bgr[c] = (((ay0 * (S0 >> 4)) >> 16) + ((ay1 * (S1 >> 4)) >> 16) + 2) >> 2;
            bgr[c] = min(max(bgr[c], 0), 255);
        }
    }
    o[0] = (float)bgr[2] / 255.f;
    o[plane] = (float)bgr[1] / 255.f;
    o[2 * plane] = (float)bgr[0] / 255.f;
}

// cv2.resize(src, (dw, dh), interpolation=INTER_LINEAR) for 8-bit images with cn interleaved channels: thread = one destination byte
__global__ __launch_bounds__(256) void k_resize_u8(const uint8_t* __restrict__ src, int sw, int sh, int cn, int dw, int dh, float scale_x, float scale_y,
                                                   uint8_t* __restrict__ dst)
{
    const int xc = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (xc >= dw * cn) return;
    const int x = xc / cn, c = xc - x * cn;
    int v;
    if (sw == 2 * dw && sh == 2 * dh) {
        const uint8_t* r0 = src + ((size_t)(2 * y) * sw + 2 * x) * cn + c;
        const uint8_t* r1 = r0 + (size_t)sw * cn;
        v = (r0[0] + r0[cn] + r1[0] + r1[cn] + 2) >> 2;
    } else {
        int sx, ax0, ax1, sy, ay0, ay1;
        lb_coef(x, scale_x, sw, sx, ax0, ax1);
        lb_coef(y, scale_y, sh, sy, ay0, ay1);
        const int sx1 = min(sx + 1, sw - 1), sy1 = min(sy + 1, sh - 1);
        const uint8_t* r0 = src + (size_t)sy * sw * cn + c;
        const uint8_t* r1 = src + (size_t)sy1 * sw * cn + c;
        const int S0 = r0[sx * cn] * ax0 + r0[sx1 * cn] * ax1;
        const int S1 = r1[sx * cn] * ax0 + r1[sx1 * cn] * ax1;
        v = (((ay0 * (S0 >> 4)) >> 16) + ((ay1 * (S1 >> 4)) >> 16) + 2) >> 2;
        v = min(max(v, 0), 255);
    }
    dst[((size_t)y * dw + x) * cn + c] = (uint8_t)v;
}

int vpk_resize_u8(vp_ctx* ctx, const uint8_t* d_src, int sw, int sh, int cn, int dw, int dh, uint8_t* d_dst)
{
    vp_prof_scope ps(ctx, VPK_OTHER);
    if (sw == dw && sh == dh) {
        VP_HIP(ctx, hipMemcpyAsync(d_dst, d_src, (size_t)sw * sh * cn, hipMemcpyDeviceToDevice, ctx->stream));
        return VP_OK;
    }
    hipLaunchKernelGGL(k_resize_u8, dim3((unsigned)((dw * cn + 255) / 256), (unsigned)dh), dim3(256), 0, ctx->stream, d_src, sw, sh, cn, dw, dh,
                       (float)((double)sw / dw), (float)((double)sh / dh), d_dst);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// cv2.warpAffine(src, M, (dw, dh), INTER_LINEAR, borderMode) on 8-bit images, OpenCV's classical fixed-point path: source coordinates in
// 22.10 fixed point from round-half-even products (no fused multiply-add: the reference arithmetic is plain IEEE double), 5 fractional
// bits kept, 15-bit bilinear weights, (sum + 2^14) >> 15.  M maps destination to source here (the host inverts).  thread = one pixel.
struct wa_params { double m[6]; int sw, sh, cn, dw, dh, border; uint8_t cval[4]; };
__device__ __forceinline__ int wa_round(double v)
{
    if (!(v > -2147483648.0)) return INT_MIN;
    if (v >= 2147483647.0) return INT_MAX;
    return (int)rint(v);
}
__global__ __launch_bounds__(256) void k_warp_affine_u8(const uint8_t* __restrict__ src, wa_params P, uint8_t* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= P.dw) return;
    const u32 ad = (u32)wa_round(__dmul_rn(__dmul_rn(P.m[0], (double)x), 1024.0));
    const u32 bd = (u32)wa_round(__dmul_rn(__dmul_rn(P.m[3], (double)x), 1024.0));
    const u32 X0 = (u32)wa_round(__dmul_rn(__dadd_rn(__dmul_rn(P.m[1], (double)y), P.m[2]), 1024.0)) + 16u;
    const u32 Y0 = (u32)wa_round(__dmul_rn(__dadd_rn(__dmul_rn(P.m[4], (double)y), P.m[5]), 1024.0)) + 16u;
    const int X = (int)(X0 + ad) >> 5, Y = (int)(Y0 + bd) >> 5;
    const int sx = min(max(X >> 5, -32768), 32767), sy = min(max(Y >> 5, -32768), 32767);
    const int fx = X & 31, fy = Y & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const int cn = P.cn, sw = P.sw, sh = P.sh;
    uint8_t* d = dst + ((size_t)y * P.dw + x) * cn;
    if (P.border == 0 && (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0)) {
        for (int c = 0; c < cn; c++) d[c] = P.cval[c];
        return;
    }
    int x0, x1, y0, y1;
    if (P.border == 1) {
        x0 = min(max(sx, 0), sw - 1); x1 = min(max(sx + 1, 0), sw - 1);
        y0 = min(max(sy, 0), sh - 1); y1 = min(max(sy + 1, 0), sh - 1);
    } else {
        x0 = (sx >= 0 && sx < sw) ? sx : -1; x1 = (sx + 1 >= 0 && sx + 1 < sw) ? sx + 1 : -1;
        y0 = (sy >= 0 && sy < sh) ? sy : -1; y1 = (sy + 1 >= 0 && sy + 1 < sh) ? sy + 1 : -1;
    }
    const bool i00 = x0 >= 0 && y0 >= 0, i01 = x1 >= 0 && y0 >= 0, i10 = x0 >= 0 && y1 >= 0, i11 = x1 >= 0 && y1 >= 0;
    const uint8_t* p00 = src + ((size_t)max(y0, 0) * sw + max(x0, 0)) * cn;
    const uint8_t* p01 = src + ((size_t)max(y0, 0) * sw + max(x1, 0)) * cn;
    const uint8_t* p10 = src + ((size_t)max(y1, 0) * sw + max(x0, 0)) * cn;
    const uint8_t* p11 = src + ((size_t)max(y1, 0) * sw + max(x1, 0)) * cn;
    for (int c = 0; c < cn; c++) {
        const int cv = P.cval[c];
        const int v00 = i00 ? p00[c] : cv, v01 = i01 ? p01[c] : cv, v10 = i10 ? p10[c] : cv, v11 = i11 ? p11[c] : cv;
        const int v = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
        d[c] = (uint8_t)min(max(v, 0), 255);
    }
}

int vpk_warp_affine_u8(vp_ctx* ctx, const uint8_t* d_src, int sw, int sh, int cn, const double* M23, int inverse_map, int border,
                       const uint8_t* cval, uint8_t* d_dst, int dw, int dh)
{
#pragma clang fp contract(off)   // the inversion is plain IEEE double in the reference arithmetic
    wa_params P;
    double M[6];
    for (int i = 0; i < 6; i++) M[i] = M23[i];
    if (!inverse_map) {
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1. / D : 0;
        const double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11; M[1] *= -D;
        M[3] *= -D; M[4] = A22;
        const double b1 = -M[0] * M[2] - M[1] * M[5];
        const double b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1; M[5] = b2;
    }
    for (int i = 0; i < 6; i++) P.m[i] = M[i];
    P.sw = sw; P.sh = sh; P.cn = cn; P.dw = dw; P.dh = dh; P.border = border;
    for (int c = 0; c < 4; c++) P.cval[c] = cval ? cval[c] : 0;
    vp_prof_scope ps(ctx, VPK_OTHER);
    hipLaunchKernelGGL(k_warp_affine_u8, dim3((unsigned)((dw + 255) / 256), (unsigned)dh), dim3(256), 0, ctx->stream, d_src, P, d_dst);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// LetterBox geometry (scale-up allowed, centred): r = min(dh/sh, dw/sw); content = round(size * r); the odd padding pixel goes
// right / bottom (round(d - 0.1), round(d + 0.1)).  geom_out: {r, left, top}
int vpk_letterbox(vp_ctx* ctx, const uint8_t* d_src, int sw, int sh, int dw, int dh, int pad, float* d_dst, float* geom_out)
{
    const double r = std::min((double)dh / sh, (double)dw / sw);
    lb_params P;
    P.sw = sw; P.sh = sh; P.dw = dw; P.dh = dh; P.pad = pad;
    P.nw = std::max(1, (int)nearbyint(sw * r));
    P.nh = std::max(1, (int)nearbyint(sh * r));
    P.nw = std::min(P.nw, dw); P.nh = std::min(P.nh, dh);
    const double px = (dw - P.nw) / 2.0, py = (dh - P.nh) / 2.0;
    P.left = (int)nearbyint(px - 0.1);
    P.top = (int)nearbyint(py - 0.1);
    P.scale_x = (float)((double)sw / P.nw);
    P.scale_y = (float)((double)sh / P.nh);
    if (geom_out) { geom_out[0] = (float)r; geom_out[1] = (float)P.left; geom_out[2] = (float)P.top; }
    vp_prof_scope ps(ctx, VPK_OTHER);
    hipLaunchKernelGGL(k_letterbox, dim3((unsigned)((dw + 63) / 64), (unsigned)dh), dim3(64), 0, ctx->stream, d_src, P, d_dst);
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}

// ---- NMS ----------------------------------------------------------------------------------------------------------------------
// rank[i] = position of box i in descending score order (ties: lower index first); order[rank[i]] = i
__global__ __launch_bounds__(256) void k_nms_rank(const float* __restrict__ scores, int n, int* __restrict__ order)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float s = scores[i];
    int r = 0;
    for (int j = 0; j < n; j++) {
        const float t = scores[j];
        r += (t > s || (t == s && j < i)) ? 1 : 0;
    }
    order[r] = i;
}

__device__ __forceinline__ float nms_iou(const float* a, const float* b)   // xyxy
{
    const float ix = fmaxf(0.f, fminf(a[2], b[2]) - fmaxf(a[0], b[0]));
    const float iy = fmaxf(0.f, fminf(a[3], b[3]) - fmaxf(a[1], b[1]));
    const float inter = ix * iy;
    const float ua = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter;
    return inter / ua;
}
// probabilistic IoU of two rotated boxes (x, y, w, h, angle) modelled as Gaussians
__device__ __forceinline__ void nms_cov(const float* b, float& A, float& B, float& Cc)
{
    const float a = b[2] * b[2] / 12.f, bb = b[3] * b[3] / 12.f;
    const float c = cosf(b[4]), s = sinf(b[4]);
    A = a * c * c + bb * s * s;
    B = a * s * s + bb * c * c;
    Cc = (a - bb) * c * s;
}
__device__ __forceinline__ float nms_probiou(const float* p, const float* q)
{
    const float eps = 1e-7f;
    float a1, b1, c1, a2, b2, c2;
    nms_cov(p, a1, b1, c1);
    nms_cov(q, a2, b2, c2);
    const float dx = p[0] - q[0], dy = p[1] - q[1];
    const float den = (a1 + a2) * (b1 + b2) - (c1 + c2) * (c1 + c2);
    const float t1 = (((a1 + a2) * dy * dy + (b1 + b2) * dx * dx) / (den + eps)) * 0.25f;
    const float t2 = (((c1 + c2) * (-dx) * dy) / (den + eps)) * 0.5f;
    const float d1 = fmaxf(a1 * b1 - c1 * c1, 0.f), d2 = fmaxf(a2 * b2 - c2 * c2, 0.f);
    const float t3 = logf(den / (4.f * sqrtf(d1 * d2) + eps) + eps) * 0.5f;
    const float bd = fminf(fmaxf(t1 + t2 + t3, eps), 100.f);
    const float hd = sqrtf(1.f - expf(-bd) + eps);
    return 1.f - hd;
}

// mask[i][w] bit b: the box at sorted position j = 64w + b (j > i) overlaps the box at position i by more than thr
// grid (n), 256 threads: block = row i, its four waves walk the words of the row
__global__ __launch_bounds__(256) void k_nms_mask(const float* __restrict__ boxes, const int* __restrict__ order, int n, int words, float thr,
                                                 unsigned long long* __restrict__ mask)
{
    const int i = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float* a = boxes + (size_t)order[i] * 4;
    const float bi[4] = {a[0], a[1], a[2], a[3]};
    for (int wj = wv; wj < words; wj += 4) {
        const int j = wj * 64 + lane;
        bool hit = false;
        if (wj >= (i >> 6) && j < n && j > i) hit = nms_iou(bi, boxes + (size_t)order[j] * 4) > thr;
        const unsigned long long m = __ballot(hit);
        if (lane == 0) mask[(size_t)i * words + wj] = m;
    }
}

// one wave: greedy pass in score order.  The "removed" bitmap lives in registers (lane l owns words l, l + 64, l + 128, l + 192:
// 16384 candidates), the bit of candidate i is broadcast from its owner, and only a surviving candidate touches memory (its mask
// row).  keep_out: original indices of the survivors, in score order
#define NMS_WPL 4   // words per lane
__global__ __launch_bounds__(64) void k_nms_greedy(const unsigned long long* __restrict__ mask, const int* __restrict__ order, int n, int words,
                                                   int max_keep, int* __restrict__ keep_out, int* __restrict__ n_keep)
{
    const int lane = threadIdx.x;
    unsigned long long rem[NMS_WPL] = {0ull, 0ull, 0ull, 0ull};
    int kept = 0;
    for (int i = 0; i < n && kept < max_keep; i++) {
        const int w = i >> 6, slot = w >> 6, owner = w & 63;
        const unsigned long long mine = slot == 0 ? rem[0] : (slot == 1 ? rem[1] : (slot == 2 ? rem[2] : rem[3]));
        const unsigned long long word = __shfl(mine, owner);
        if ((word >> (i & 63)) & 1ull) continue;   // uniform
        if (lane == 0) keep_out[kept] = order[i];
        kept++;
        const unsigned long long* row = mask + (size_t)i * words;
#pragma unroll
        for (int k = 0; k < NMS_WPL; k++) {
            const int ww = lane + 64 * k;
            if (ww < words) rem[k] |= row[ww];
        }
    }
    if (lane == 0) *n_keep = kept;
}

// rotated variant, "fast" rule: a box survives when no higher-scored box (suppressed or not) overlaps it by >= thr.
// grid (n), 64 threads: block j tests the boxes at sorted positions i < j
__global__ __launch_bounds__(64) void k_nms_rot_alive(const float* __restrict__ boxes, const int* __restrict__ order, int n, float thr,
                                                      unsigned char* __restrict__ alive)
{
    const int j = blockIdx.x;
    const float* b = boxes + (size_t)order[j] * 5;
    const float bj[5] = {b[0], b[1], b[2], b[3], b[4]};
    bool hit = false;
    for (int i0 = 0; i0 < j && !hit; i0 += 64) {
        const int i = i0 + threadIdx.x;
        bool h = false;
        if (i < j) h = nms_probiou(boxes + (size_t)order[i] * 5, bj) >= thr;
        hit = __any(h);
    }
    if (threadIdx.x == 0) alive[j] = hit ? 0 : 1;
}
// one wave: survivors in score order
__global__ __launch_bounds__(64) void k_nms_compact(const unsigned char* __restrict__ alive, const int* __restrict__ order, int n, int max_keep,
                                                    int* __restrict__ keep_out, int* __restrict__ n_keep)
{
    int kept = 0;
    for (int j0 = 0; j0 < n && kept < max_keep; j0 += 64) {
        const int j = j0 + threadIdx.x;
        const bool a = j < n && alive[j];
        const unsigned long long m = __ballot(a);
        const int pos = kept + __popcll(m & ((1ull << threadIdx.x) - 1ull));
        if (a && pos < max_keep) keep_out[pos] = order[j];
        kept += __popcll(m);
    }
    if (threadIdx.x == 0) *n_keep = min(kept, max_keep);
}

size_t vp_nms_ws_bytes(int n) { const size_t words = (size_t)(n + 63) / 64; return vp_align((size_t)n * 4) + vp_align((size_t)n * words * 8) + vp_align((size_t)n) + 1024; }

// d_boxes: (n,4) xyxy or (n,5) xywhr; d_keep: max_keep ints; d_nkeep: 1 int.  rotated = 0: greedy IoU NMS (suppress IoU > thr);
// rotated = 1: probabilistic IoU, a box is dropped when some higher-scored box overlaps it by >= thr
int vpk_nms(vp_ctx* ctx, const float* d_boxes, const float* d_scores, int n, float thr, int rotated, int max_keep, int* d_keep, int* d_nkeep)
{
    if (n <= 0) { VP_HIP(ctx, hipMemsetAsync(d_nkeep, 0, 4, ctx->stream)); return VP_OK; }
    if (n > 16384) return vp_fail(ctx, VP_ERR_UNSUPPORTED, "nms: more than 16384 candidates");
    const int words = (n + 63) / 64;
    int* order = (int*)vp_ws_take(ctx, (size_t)n * 4);
    unsigned long long* mask = (unsigned long long*)vp_ws_take(ctx, (size_t)n * words * 8);
    unsigned char* alive = (unsigned char*)vp_ws_take(ctx, (size_t)n);
    if (!order || !mask || !alive) return vp_fail(ctx, VP_ERR_NOMEM, "nms workspace");
    hipStream_t s = ctx->stream;
    vp_prof_scope ps(ctx, VPK_OTHER);
    hipLaunchKernelGGL(k_nms_rank, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_scores, n, order);
    if (rotated) {
        hipLaunchKernelGGL(k_nms_rot_alive, dim3((unsigned)n), dim3(64), 0, s, d_boxes, order, n, thr, alive);
        hipLaunchKernelGGL(k_nms_compact, dim3(1), dim3(64), 0, s, alive, order, n, max_keep, d_keep, d_nkeep);
    } else {
        hipLaunchKernelGGL(k_nms_mask, dim3((unsigned)n), dim3(256), 0, s, d_boxes, order, n, words, thr, mask);
        hipLaunchKernelGGL(k_nms_greedy, dim3(1), dim3(64), 0, s, mask, order, n, words, max_keep, d_keep, d_nkeep);
    }
    VP_HIP(ctx, hipGetLastError());
    return VP_OK;
}
