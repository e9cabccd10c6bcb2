/*
 * vp_oracle_warp.c — CPU ORACLE, part 3: cv2.warpAffine with bilinear interpolation on 8-bit images and cv2.Canny (at the end)
 * (modules/preprocessor.py:130-135,145-149; utils/transform.py:180-210 rotate / translate).
 *
 * TEST INFRASTRUCTURE ONLY (same rules as vp_oracle.c): imported by tests/, never by the product.
 * PARITY UNPINNED: OpenCV is absent and the reference holds no vectors for this call.  What is restated is OpenCV's
 * classical fixed-point path (imgwarp.cpp, cv::warpAffine + WarpAffineInvoker + remapBilinear with FixedPtCast<int, uchar, 15>),
 * the arithmetic of every release up to 4.10:
 *   1. the 2x3 matrix is converted to double and inverted in place unless WARP_INVERSE_MAP is given
 *      (D = 1 / (M0 M4 - M1 M3), 0 when singular);
 *   2. source coordinates are tracked in 22.10 fixed point: adelta[x] = cvRound(M0 x 1024), bdelta[x] = cvRound(M3 x 1024),
 *      X0 = cvRound((M1 y + M2) 1024) + 16, Y0 likewise; X = (X0 + adelta[x]) >> 5, so 5 fractional bits survive;
 *      the integer parts are saturated to int16;
 *   3. the four neighbours are blended with 15-bit weights (32 - fx)(32 - fy) 32, ... (they sum to 32768 exactly, so OpenCV's
 *      table correction step never fires for the bilinear table) and rounded half up: (sum + 16384) >> 15;
 *   4. outside the source: BORDER_REPLICATE clamps each neighbour's coordinates; BORDER_CONSTANT writes the border value when
 *      all four neighbours are outside and substitutes it for the outside ones otherwise.
 * cvRound is round-half-to-even (lrint under the default rounding mode).  The products M x and M y + b are plain IEEE double
 * operations without fused multiply-add (OpenCV's x86-64 baseline build; this file is compiled with -ffp-contract=off).
 * Releases from 4.11 on carry a second, float-based bilinear kernel for some type/channel combinations, which can differ by
 * one grey level on non-integer maps; integer translations are identical in both.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static int warp_round(double v)   /* saturate_cast<int>(double) */
{
    if (!(v > -2147483648.0)) return INT32_MIN;   /* also NaN */
    if (v >= 2147483647.0) return INT32_MAX;
    return (int)lrint(v);
}
static int warp_sat16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
static int warp_clip(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

/* border: 0 = BORDER_CONSTANT (value cval[c]), 1 = BORDER_REPLICATE.  inverse_map != 0: M already maps dst -> src. */
ORC_API int orc_warp_affine_u8(const uint8_t* src, int sw, int sh, int cn, const double* M23, int inverse_map, int border,
                               const uint8_t* cval, uint8_t* dst, int dw, int dh)
{
    if (!src || !dst || !M23 || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || cn < 1 || cn > 4 || (border != 0 && border != 1)) return -1;
    double M[6];
    memcpy(M, M23, sizeof M);
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
    int* adelta = (int*)malloc(sizeof(int) * 2 * (size_t)dw);
    if (!adelta) return -2;
    int* bdelta = adelta + dw;
    for (int x = 0; x < dw; x++) {
        adelta[x] = warp_round(M[0] * x * 1024);
        bdelta[x] = warp_round(M[3] * x * 1024);
    }
    const uint8_t zero[4] = {0, 0, 0, 0};
    if (!cval) cval = zero;
    for (int y = 0; y < dh; y++) {
        const int X0 = (int)((unsigned)warp_round((M[1] * y + M[2]) * 1024) + 16u);
        const int Y0 = (int)((unsigned)warp_round((M[4] * y + M[5]) * 1024) + 16u);
        for (int x = 0; x < dw; x++) {
            const int X = (int)((unsigned)X0 + (unsigned)adelta[x]) >> 5;
            const int Y = (int)((unsigned)Y0 + (unsigned)bdelta[x]) >> 5;
            const int sx = warp_sat16(X >> 5), sy = warp_sat16(Y >> 5);
            const int fx = X & 31, fy = Y & 31;
            const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            uint8_t* d = dst + ((size_t)y * dw + x) * cn;
            if (border == 0 && (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0)) {
                for (int c = 0; c < cn; c++) d[c] = cval[c];
                continue;
            }
            int x0, x1, y0, y1;
            if (border == 1) {
                x0 = warp_clip(sx, 0, sw); x1 = warp_clip(sx + 1, 0, sw);
                y0 = warp_clip(sy, 0, sh); y1 = warp_clip(sy + 1, 0, sh);
            } else {
                x0 = (sx >= 0 && sx < sw) ? sx : -1; x1 = (sx + 1 >= 0 && sx + 1 < sw) ? sx + 1 : -1;
                y0 = (sy >= 0 && sy < sh) ? sy : -1; y1 = (sy + 1 >= 0 && sy + 1 < sh) ? sy + 1 : -1;
            }
            for (int c = 0; c < cn; c++) {
                const int v00 = (x0 >= 0 && y0 >= 0) ? src[((size_t)y0 * sw + x0) * cn + c] : cval[c];
                const int v01 = (x1 >= 0 && y0 >= 0) ? src[((size_t)y0 * sw + x1) * cn + c] : cval[c];
                const int v10 = (x0 >= 0 && y1 >= 0) ? src[((size_t)y1 * sw + x0) * cn + c] : cval[c];
                const int v11 = (x1 >= 0 && y1 >= 0) ? src[((size_t)y1 * sw + x1) * cn + c] : cval[c];
                const int v = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
                d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
    free(adelta);
    return 0;
}

/* cv2.getRotationMatrix2D(center, angle, scale): the centre is a Point2f (float32) in OpenCV */
ORC_API void orc_rotation_matrix_2d(double cx, double cy, double angle_deg, double scale, double* M23)
{
    const float fcx = (float)cx, fcy = (float)cy;
    const double a = angle_deg * (3.14159265358979323846 / 180.0);
    const double alpha = cos(a) * scale, beta = sin(a) * scale;
    M23[0] = alpha; M23[1] = beta; M23[2] = (1 - alpha) * fcx - beta * fcy;
    M23[3] = -beta; M23[4] = alpha; M23[5] = beta * fcx + (1 - alpha) * fcy;
}

/* ---- cv2.Canny(image, threshold1, threshold2) with the defaults apertureSize = 3, L2gradient = false (utils/feature.py:43-101 canny /
 * simple_canny) — OpenCV canny.cpp restated sequentially:
 *   Sobel 3x3 derivatives (16-bit, BORDER_REPLICATE); for several channels the channel with the largest |dx| + |dy| (first on ties);
 *   thresholds ordered, then floored to int; a pixel with magnitude m > low survives non-maximum suppression when
 *     |dy| 2^15 < |dx| TG22            : m >  left  and m >= right                (TG22 = round(tan 22.5 deg 2^15) = 13573)
 *     |dy| 2^15 > |dx| (TG22 + 2^16)   : m >  above and m >= below
 *     otherwise (diagonal)             : m >  both diagonal neighbours on the line of the gradient (sign of dx ^ dy picks the diagonal)
 *   with magnitude 0 outside the image; survivors above `high` are edges, the others become edges when 8-connected to an edge through
 *   survivors (stack flood); output 255 / 0. */
ORC_API int orc_canny_u8(const uint8_t* src, int w, int h, int cn, double t1, double t2, uint8_t* dst)
{
    if (!src || !dst || w <= 0 || h <= 0 || cn < 1 || cn > 4) return -1;
    if (t1 > t2) { const double t = t1; t1 = t2; t2 = t; }
    const int low = (int)floor(t1), high = (int)floor(t2);
    const size_t n = (size_t)w * h;
    int* mag = (int*)calloc((size_t)(w + 2) * (h + 2), sizeof(int));
    short* dxs = (short*)malloc(n * sizeof(short));
    short* dys = (short*)malloc(n * sizeof(short));
    uint8_t* map = (uint8_t*)malloc((size_t)(w + 2) * (h + 2));
    size_t* stack = (size_t*)malloc(n * sizeof(size_t));
    if (!mag || !dxs || !dys || !map || !stack) { free(mag); free(dxs); free(dys); free(map); free(stack); return -2; }
    const size_t ms = (size_t)w + 2;
#define PX(yy, xx, c) ((int)src[((size_t)warp_clip((yy), 0, h) * w + warp_clip((xx), 0, w)) * cn + (c)])
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int bm = -1, bdx = 0, bdy = 0;
            for (int c = 0; c < cn; c++) {
                const int dx = (PX(y - 1, x + 1, c) + 2 * PX(y, x + 1, c) + PX(y + 1, x + 1, c)) - (PX(y - 1, x - 1, c) + 2 * PX(y, x - 1, c) + PX(y + 1, x - 1, c));
                const int dy = (PX(y + 1, x - 1, c) + 2 * PX(y + 1, x, c) + PX(y + 1, x + 1, c)) - (PX(y - 1, x - 1, c) + 2 * PX(y - 1, x, c) + PX(y - 1, x + 1, c));
                const int m = abs(dx) + abs(dy);
                if (m > bm) { bm = m; bdx = dx; bdy = dy; }
            }
            mag[(size_t)(y + 1) * ms + x + 1] = bm;
            dxs[(size_t)y * w + x] = (short)bdx;
            dys[(size_t)y * w + x] = (short)bdy;
        }
#undef PX
    memset(map, 1, (size_t)(w + 2) * (h + 2));
    size_t sp = 0;
    const int TG22 = 13573;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int* a = mag + (size_t)(y + 1) * ms + x + 1;
            const int m = a[0];
            uint8_t* pm = map + (size_t)(y + 1) * ms + x + 1;
            int keep = 0;
            if (m > low) {
                const int xs = dxs[(size_t)y * w + x], ys = dys[(size_t)y * w + x];
                const int ax = abs(xs), ay = abs(ys) << 15;
                const int tg22x = ax * TG22;
                if (ay < tg22x) keep = m > a[-1] && m >= a[1];
                else {
                    const int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) keep = m > a[-(ptrdiff_t)ms] && m >= a[ms];
                    else {
                        const int s = (xs ^ ys) < 0 ? -1 : 1;
                        keep = m > a[-(ptrdiff_t)ms - s] && m > a[(ptrdiff_t)ms + s];
                    }
                }
            }
            if (keep) {
                if (m > high) { *pm = 2; stack[sp++] = (size_t)(pm - map); }
                else *pm = 0;
            }
        }
    while (sp) {
        uint8_t* m = map + stack[--sp];
        const ptrdiff_t nb[8] = {-(ptrdiff_t)ms - 1, -(ptrdiff_t)ms, -(ptrdiff_t)ms + 1, -1, 1, (ptrdiff_t)ms - 1, (ptrdiff_t)ms, (ptrdiff_t)ms + 1};
        for (int k = 0; k < 8; k++)
            if (!m[nb[k]]) { m[nb[k]] = 2; stack[sp++] = (size_t)(m + nb[k] - map); }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[(size_t)y * w + x] = map[(size_t)(y + 1) * ms + x + 1] == 2 ? 255 : 0;
    free(mag); free(dxs); free(dys); free(map); free(stack);
    return 0;
}

/* ---- cv2.adaptiveThreshold(src, maxValue, ADAPTIVE_THRESH_MEAN_C, type, blockSize, C) (utils/color.py:220-254 adaptive_threshold_mean /
 * _inv) — thresh.cpp: mean = boxFilter(src, blockSize x blockSize, normalised, BORDER_REPLICATE) rounded to uint8, then a table over
 * src - mean + 255: THRESH_BINARY keeps src - mean > -ceil(C), THRESH_BINARY_INV keeps src - mean <= -floor(C).
 * The rounding of the mean: OpenCV has three code paths (16-bit sums with a Q23 reciprocal for windows of at most 256 pixels,
 * float32 in the vector body and double in the scalar tail otherwise); for odd block sizes up to 151 all three equal the exact
 * nearest integer of sum / blockSize^2 (no ties exist for an odd divisor) — checked exhaustively over every possible sum in
 * tests/test_oracle.py — so the restatement is the exact integer rounding.  type: 0 = THRESH_BINARY, 1 = THRESH_BINARY_INV. */
ORC_API int orc_adaptive_threshold_mean_u8(const uint8_t* src, int w, int h, double max_value, int type, int block, double C, uint8_t* dst)
{
    if (!src || !dst || w <= 0 || h <= 0 || block < 3 || (block & 1) == 0 || (type != 0 && type != 1)) return -1;
    const size_t n = (size_t)w * h;
    if (max_value < 0) { memset(dst, 0, n); return 0; }
    long mv = lrint(max_value);
    const int imax = (int)(mv < 0 ? 0 : (mv > 255 ? 255 : mv));
    const int idelta = type == 0 ? (int)ceil(C) : (int)floor(C);
    const int r = block / 2, d = block * block;
    uint32_t* hs = (uint32_t*)malloc(n * sizeof(uint32_t));
    if (!hs) return -2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int k = -r; k <= r; k++) s += src[(size_t)y * w + warp_clip(x + k, 0, w)];
            hs[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int k = -r; k <= r; k++) s += hs[(size_t)warp_clip(y + k, 0, h) * w + x];
            const int mean = (int)((2 * (uint64_t)s + (uint64_t)d) / (2 * (uint64_t)d));
            const int diff = (int)src[(size_t)y * w + x] - mean;
            const int on = type == 0 ? diff > -idelta : diff <= -idelta;
            dst[(size_t)y * w + x] = (uint8_t)(on ? imax : 0);
        }
    free(hs);
    return 0;
}
