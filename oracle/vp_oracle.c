/*
 * vp_oracle.c — CPU ORACLE for the colour -> threshold -> morphology -> CCL path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped path (libvp.so, HIP) never
 * links, imports or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference (ayf7/cuauv-vision-pipeline) ships no tests, fixtures or
 * golden vectors (build.ninja:72-73 are empty phony targets) and all of its hot-path
 * arithmetic lives in an un-vendored third-party dependency, OpenCV 4.x (cv2, version
 * unpinned: no requirements file in the tree; `opencv4` via pkg-config, configure.py:31).
 * cv2 is not installed in the build container, so this file restates OpenCV's published
 * 8-bit algorithms and is pinned only by (a) the widely published OpenCV known answers of
 * SURVEY.md Appendix A (tests/golden/known_answers.json), (b) SciPy as an independent
 * witness where semantics coincide, (c) float64 analytic formulas at +-1 LSB and (d) for the
 * one numpy-only stretch of the path (thresh_color_distance, utils/color.py:91-103) vectors made
 * under a real numpy 1.x (tests/golden/numpy1_color_distance.npz).
 * tests/test_live_cv2.py compares against a real cv2 whenever one is importable.
 *
 * Each function cites the reference call site it stands in for (file:line relative to
 * /root/reference) and the OpenCV routine whose semantics it restates.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: the table generators rely on
 * separately rounded multiplies and adds).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Tables.  OpenCV builds them once with softfloat (IEEE binary32 / binary64, round to nearest
 * even, no fused contraction except the explicit mulAdd) in color_lab.cpp initLabTabs() and
 * color_hsv.simd.hpp.  Plain C float/double arithmetic is the same arithmetic, provided the
 * compiler neither contracts a*b+c nor uses x87 precision (x86-64 SSE2: fine).
 * ---------------------------------------------------------------------------------------- */

static int cv_round_d(double v) { return (int)nearbyint(v); } /* cvRound: half to even */
static int cv_round_f(float v) { return (int)nearbyintf(v); }

/* cv::cubeRoot(softfloat) (core/src/softfloat.cpp f32_cbrt, mirroring mathfuncs.cpp
 * cubeRoot): exponent split + quartic rational polynomial evaluated in binary64, rounded
 * to binary32. */
static float cv_cube_root_f32(float value)
{
    union { float f; int32_t i; uint32_t u; } v, m;
    v.f = value;
    int32_t ix = v.i & 0x7fffffff;
    int32_t s = v.i & (int32_t)0x80000000;
    int ex = (ix >> 23) - 127;
    int shx = ex % 3;
    shx -= shx >= 0 ? 3 : 0;
    ex = (ex - shx) / 3; /* exponent of the cube root */
    v.i = (ix & ((1 << 23) - 1)) | ((shx + 127) << 23);
    double fr = (double)v.f; /* 0.125 <= fr < 1 */
    double num = ((((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr +
                    119.1654824285581628956914143) * fr + 13.43250139086239872172837314) * fr +
                  0.1636161226585754240958355063);
    double den = ((((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr +
                    168.5254414101568283957668343) * fr + 33.9905941350215598754191872) * fr + 1.0);
    float frf = (float)(num / den);
    m.f = value;
    v.f = frf;
    v.i = (v.i + (ex << 23) + s) & ((m.u * 2u) != 0 ? -1 : 0);
    return v.f;
}

static uint16_t g_gamma[256];   /* sRGBGammaTab_b */
static uint16_t g_cbrt[3072];   /* LabCbrtTab_b   */
static int32_t g_sdiv[256];     /* sdiv_table     */
static int32_t g_hdiv180[256];  /* hdiv_table180  */
static int32_t g_labC[9];       /* RGB2Lab_b coeffs, rows X,Y,Z, columns R,G,B */
static int g_tables_ready = 0;

/* variant 0: softfloat-faithful (binary32 where OpenCV uses softfloat); variant 1: plain
 * binary64 + half-even (what SURVEY.md Appendix A1 describes).  The two are compared by
 * tests/test_oracle_tables.py; entries that differ are the ones a live cv2 must settle. */
ORC_API void orc_build_lab_tables(int variant, uint16_t* gamma, uint16_t* cbrt_tab)
{
    for (int i = 0; i < 256; i++) {
        if (variant == 0) {
            float x = (float)i / 255.0f;
            double xd = (double)x;
            double g = xd <= 0.04045 ? xd / 12.92 : pow((xd + 0.055) / 1.055, 2.4);
            float gf = (float)g;
            gamma[i] = (uint16_t)cv_round_f(2040.0f * gf);
        } else {
            double x = i / 255.0;
            double g = x <= 0.04045 ? x / 12.92 : pow((x + 0.055) / 1.055, 2.4);
            gamma[i] = (uint16_t)cv_round_d(2040.0 * g);
        }
    }
    const float lthresh = 216.0f / 24389.0f;
    const float lbias = 16.0f / 116.0f;
    const float lscale = 841.0f / 108.0f;
    for (int i = 0; i < 3072; i++) {
        if (variant == 0) {
            float x = (float)i / 2040.0f;
            float f = x < lthresh ? fmaf(x, lscale, lbias) : cv_cube_root_f32(x);
            cbrt_tab[i] = (uint16_t)cv_round_f(32768.0f * f);
        } else {
            double x = i / 2040.0;
            double f = x < 216.0 / 24389.0 ? x * (841.0 / 108.0) + 16.0 / 116.0 : cbrt(x);
            cbrt_tab[i] = (uint16_t)cv_round_d(32768.0 * f);
        }
    }
}

static void build_tables(void)
{
    if (g_tables_ready) return;
    orc_build_lab_tables(0, g_gamma, g_cbrt);
    /* color_lab.cpp RGB2Lab_b ctor: coeffs = cvRound((1<<12) * M[i][k] / whitePt[i]) */
    static const double M[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169,
                                0.019334, 0.119193, 0.950227};
    static const double wp[3] = {0.950456, 1.0, 1.088754};
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) g_labC[i * 3 + k] = cv_round_d(4096.0 * M[i * 3 + k] / wp[i]);
    /* color_hsv.simd.hpp RGB2HSV_b: sdiv[i] = cvRound((255<<12)/i), hdiv[i] = cvRound((180<<12)/(6 i)) */
    g_sdiv[0] = g_hdiv180[0] = 0;
    for (int i = 1; i < 256; i++) {
        g_sdiv[i] = cv_round_d((255 << 12) / (1.0 * i));
        g_hdiv180[i] = cv_round_d((180 << 12) / (6.0 * i));
    }
    g_tables_ready = 1;
}

ORC_API void orc_get_tables(uint16_t* gamma, uint16_t* cbrt_tab, int32_t* sdiv, int32_t* hdiv180,
                            int32_t* labC)
{
    build_tables();
    if (gamma) memcpy(gamma, g_gamma, sizeof g_gamma);
    if (cbrt_tab) memcpy(cbrt_tab, g_cbrt, sizeof g_cbrt);
    if (sdiv) memcpy(sdiv, g_sdiv, sizeof g_sdiv);
    if (hdiv180) memcpy(hdiv180, g_hdiv180, sizeof g_hdiv180);
    if (labC) memcpy(labC, g_labC, sizeof g_labC);
}

static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

/* ------------------------------------------------------------------------------------------
 * a1  bgr_to_lab — utils/color.py:11-32 (cv2.cvtColor(COLOR_BGR2LAB) + cv2.split)
 *     OpenCV: color_lab.cpp RGB2Lab_b::operator(), blueIdx = 0, srgb = true.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_bgr2lab_u8(const uint8_t* src, size_t sstride, int w, int h, uint8_t* dst,
                            size_t dstride)
{
    build_tables();
    const int Lscale = (116 * 255 + 50) / 100;
    const int Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    const int32_t* C = g_labC;
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++, s += 3, d += 3) {
            int B = g_gamma[s[0]], G = g_gamma[s[1]], R = g_gamma[s[2]];
            int fX = g_cbrt[DESCALE(R * C[0] + G * C[1] + B * C[2], 12)];
            int fY = g_cbrt[DESCALE(R * C[3] + G * C[4] + B * C[5], 12)];
            int fZ = g_cbrt[DESCALE(R * C[6] + G * C[7] + B * C[8], 12)];
            int L = DESCALE(Lscale * fY + Lshift, 15);
            int a = DESCALE(500 * (fX - fY) + 128 * (1 << 15), 15);
            int b = DESCALE(200 * (fY - fZ) + 128 * (1 << 15), 15);
            d[0] = sat_u8(L);
            d[1] = sat_u8(a);
            d[2] = sat_u8(b);
        }
    }
}

/* a2  bgr_to_hsv — utils/color.py:26-32, modules/bins.py:13, modules/preprocessor.py:62
 *     OpenCV: color_hsv.simd.hpp RGB2HSV_b::operator(), hrange = 180, blueIdx = 0. */
ORC_API void orc_bgr2hsv_u8(const uint8_t* src, size_t sstride, int w, int h, uint8_t* dst,
                            size_t dstride)
{
    build_tables();
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++, s += 3, d += 3) {
            int b = s[0], g = s[1], r = s[2];
            int v = b > g ? b : g;
            if (r > v) v = r;
            int vmin = b < g ? b : g;
            if (r < vmin) vmin = r;
            int diff = v - vmin;
            int vr = v == r ? -1 : 0;
            int vg = v == g ? -1 : 0;
            int sv = (diff * g_sdiv[v] + (1 << 11)) >> 12;
            int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
            hh = (hh * g_hdiv180[diff] + (1 << 11)) >> 12;
            hh += hh < 0 ? 180 : 0;
            d[0] = sat_u8(hh);
            d[1] = (uint8_t)sv;
            d[2] = (uint8_t)v;
        }
    }
}

/* a3  bgr_to_gray — utils/color.py:26-32 (handlers/torpedoes.py:207-209)
 *     OpenCV: color_rgb.simd.hpp RGB2Gray<uchar>: (b*B2Y + g*G2Y + r*R2Y + 2^13) >> 14 */
ORC_API void orc_bgr2gray_u8(const uint8_t* src, size_t sstride, int w, int h, uint8_t* dst,
                             size_t dstride)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++, s += 3)
            d[x] = (uint8_t)((s[0] * 1868 + s[1] * 9617 + s[2] * 4899 + (1 << 13)) >> 14);
    }
}

/* COLOR_BGR2YCrCb, 8-bit — utils/color.py:26-32 bgr_to_ycrcb, modules/preprocessor.py:71-75.  OpenCV RGB2YCrCb_i<uchar>
 * (color_yuv.simd.hpp): Q14 integers, Y as in BGR2GRAY, Cr = DESCALE((R - Y) 11682 + 128 2^14, 14), Cb = DESCALE((B - Y) 9241 + ...),
 * stored Y, Cr, Cb.  Known answers: red -> (76, 255, 85), blue -> (29, 107, 255), greys -> (v, 128, 128). */
ORC_API void orc_bgr2ycrcb_u8(const uint8_t* src, size_t sstride, int w, int h, uint8_t* dst, size_t dstride)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++, s += 3, d += 3) {
            const int b = s[0], g = s[1], r = s[2];
            const int Y = (b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14;
            const int Cr = ((r - Y) * 11682 + (128 << 14) + (1 << 13)) >> 14;
            const int Cb = ((b - Y) * 9241 + (128 << 14) + (1 << 13)) >> 14;
            d[0] = sat_u8(Y); d[1] = sat_u8(Cr); d[2] = sat_u8(Cb);
        }
    }
}

/* COLOR_BGR2HLS, 8-bit — utils/color.py:26-32 bgr_to_hls, modules/preprocessor.py:66-70.  OpenCV RGB2HLS_b (color_hsv.simd.hpp):
 * the pixel goes to float32 through a multiplication by (1.f/255.f), RGB2HLS_f's scalar statement sequence runs with hrange 180, and
 * H, L*255, S*255 are stored with round-half-to-even saturation.  Every operation below is a single IEEE float32 operation in the
 * order of the reference statements (this file is compiled with -ffp-contract=off).  OpenCV's vector form of the same function may
 * fuse the multiply-add of the hue; that form is not restated. */
static uint8_t sat_u8_f(float v)
{
    const long r = lrintf(v);
    return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}
ORC_API void orc_bgr2hls_u8(const uint8_t* src, size_t sstride, int w, int h, uint8_t* dst, size_t dstride)
{
    const float scale = 1.f / 255.f;
    for (int y = 0; y < h; y++) {
        const uint8_t* sp = src + (size_t)y * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++, sp += 3, d += 3) {
            const float b = sp[0] * scale, g = sp[1] * scale, r = sp[2] * scale;
            float hh = 0.f, s = 0.f, l, vmin, vmax, diff;
            vmax = vmin = r;
            if (vmax < g) vmax = g;
            if (vmax < b) vmax = b;
            if (vmin > g) vmin = g;
            if (vmin > b) vmin = b;
            diff = vmax - vmin;
            l = (vmax + vmin) * 0.5f;
            if (diff > 1.1920928955078125e-7f) {
                s = l < 0.5f ? diff / (vmax + vmin) : diff / (2 - vmax - vmin);
                diff = 60.f / diff;
                if (vmax == r) hh = (g - b) * diff;
                else if (vmax == g) hh = (b - r) * diff + 120.f;
                else hh = (r - g) * diff + 240.f;
                if (hh < 0.f) hh += 360.f;
            }
            d[0] = sat_u8_f(hh * 0.5f); d[1] = sat_u8_f(l * 255.f); d[2] = sat_u8_f(s * 255.f);
        }
    }
}

/* COLOR_GRAY2BGR — modules/bins.py:19: replicate */
ORC_API void orc_gray2bgr_u8(const uint8_t* src, size_t sstride, int w, int h, uint8_t* dst,
                             size_t dstride)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = s[x];
    }
}

/* cv2.split on a 3-channel u8 image — utils/color.py:22 */
ORC_API void orc_split3_u8(const uint8_t* src, size_t sstride, int w, int h, uint8_t* p0,
                           uint8_t* p1, uint8_t* p2)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            p0[(size_t)y * w + x] = s[3 * x];
            p1[(size_t)y * w + x] = s[3 * x + 1];
            p2[(size_t)y * w + x] = s[3 * x + 2];
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * a4  range_threshold / cv2.inRange — utils/color.py:105-121, modules/bins.py:16
 *     OpenCV: core/src/arithm.cpp inRange(): bounds are integers after cvRound; a channel whose
 *     (lo > hi || lo > 255 || hi < 0) selects nothing; otherwise bounds saturate to u8.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_inrange_u8(const uint8_t* src, size_t sstride, int w, int h, int cn,
                            const int32_t* lo, const int32_t* hi, uint8_t* dst, size_t dstride)
{
    int l[4], u[4];
    for (int k = 0; k < cn; k++) {
        l[k] = lo[k];
        u[k] = hi[k];
        if (l[k] > u[k] || l[k] > 255 || u[k] < 0) { l[k] = 1; u[k] = 0; }
        else { if (l[k] < 0) l[k] = 0; if (u[k] > 255) u[k] = 255; }
    }
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            int ok = 1;
            for (int k = 0; k < cn; k++) ok &= (s[x * cn + k] >= l[k]) & (s[x * cn + k] <= u[k]);
            d[x] = ok ? 255 : 0;
        }
    }
}

/* inRange on CV_32FC1 (the `dists` image of thresh_color_distance, utils/color.py:103) */
ORC_API void orc_inrange_f32(const float* src, size_t sstride_bytes, int w, int h, float lo, float hi,
                             uint8_t* dst, size_t dstride)
{
    for (int y = 0; y < h; y++) {
        const float* s = (const float*)((const uint8_t*)src + (size_t)y * sstride_bytes);
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) d[x] = (s[x] >= lo && s[x] <= hi) ? 255 : 0;
    }
}

/* a5  thresh_color_distance — utils/color.py:66-103 (numpy, float32 accumulation in channel
 *     order 0,1,2; numpy-1.x scalar semantics: the float64 weight is demoted to float32).
 *     wts are the already normalised float32 weights (0 for ignored channels, which the
 *     reference skips entirely: skipmask bit i).  Returns d2 (float32) and uint8(sqrt(d2))
 *     (C truncation, modulo 256 like numpy's astype on x86). */
ORC_API void orc_color_distance_u8(const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, int w, int h,
                                   const float* color, const float* wts, int skipmask, float* d2out,
                                   uint8_t* sqrt_out)
{
    const uint8_t* p[3] = {p0, p1, p2};
    size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; i++) {
        float acc = 0.0f;
        for (int c = 0; c < 3; c++) {
            if (skipmask & (1 << c)) continue;
            float t = (float)p[c][i] - color[c];
            float sq = t * t;
            float term = wts[c] * sq;
            acc = acc + term;
        }
        if (d2out) d2out[i] = acc;
        if (sqrt_out) sqrt_out[i] = (uint8_t)(int32_t)sqrtf(acc);
    }
}

/* ------------------------------------------------------------------------------------------
 * a6  rect_kernel / elliptic_kernel — utils/transform.py:27-77
 *     OpenCV: imgproc/src/morph.dispatch.cpp getStructuringElement().
 *     shape 0 = MORPH_RECT, 1 = MORPH_CROSS, 2 = MORPH_ELLIPSE; anchor = centre.
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_structuring_element(int shape, int kw, int kh, uint8_t* out)
{
    if (kw <= 0 || kh <= 0) return -1;
    int r = 0, c = 0;
    double inv_r2 = 0;
    int ax = kw / 2, ay = kh / 2;
    if (kw == 1 && kh == 1) shape = 0;
    if (shape == 2) {
        r = kh / 2;
        c = kw / 2;
        inv_r2 = r ? 1.0 / ((double)r * r) : 0;
    }
    for (int i = 0; i < kh; i++) {
        int j1 = 0, j2 = 0;
        if (shape == 0 || (shape == 1 && i == ay)) j2 = kw;
        else if (shape == 1) { j1 = ax; j2 = j1 + 1; }
        else {
            int dy = i - r;
            if (abs(dy) <= r) {
                int dx = cv_round_d(c * sqrt((r * r - dy * dy) * inv_r2));
                j1 = c - dx > 0 ? c - dx : 0;
                j2 = c + dx + 1 < kw ? c + dx + 1 : kw;
            }
        }
        for (int j = 0; j < kw; j++) out[i * kw + j] = (j >= j1 && j < j2) ? 1 : 0;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a7/a8  erode / dilate / morphologyEx — utils/transform.py:80-164, modules/preprocessor.py:120-129
 *     OpenCV: morph.dispatch.cpp morphOp(): empty kernel -> 3x3 rect scaled by iterations;
 *     iterations==0 or 1x1 kernel -> copy; an all-ones kernel with iterations>1 collapses to one
 *     pass with size 1+n(k-1) and anchor*n; otherwise n passes.  Border: BORDER_CONSTANT with
 *     morphologyDefaultBorderValue() => out-of-image samples never win.
 *     dst(x,y) = min/max over kernel(i,j)!=0 of src(x + j - ax, y + i - ay).
 * ---------------------------------------------------------------------------------------- */
enum { ORC_ERODE = 0, ORC_DILATE = 1, ORC_OPEN = 2, ORC_CLOSE = 3, ORC_GRADIENT = 4 };

static void morph_pass(int is_dilate, const uint8_t* src, int w, int h, int cn, const uint8_t* k, int kw,
                       int kh, int ax, int ay, uint8_t* dst)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                int best = is_dilate ? 0 : 255;
                for (int i = 0; i < kh; i++) {
                    int yy = y + i - ay;
                    if (yy < 0 || yy >= h) continue;
                    for (int j = 0; j < kw; j++) {
                        if (!k[i * kw + j]) continue;
                        int xx = x + j - ax;
                        if (xx < 0 || xx >= w) continue;
                        int v = src[((size_t)yy * w + xx) * cn + c];
                        if (is_dilate ? v > best : v < best) best = v;
                    }
                }
                dst[((size_t)y * w + x) * cn + c] = (uint8_t)best;
            }
}

/* separable fast path for all-ones kernels (same result as morph_pass; used so that the
 * cpu_baseline is not a strawman) */
static void morph_rect_pass(int is_dilate, const uint8_t* src, int w, int h, int cn, int kw, int kh, int ax,
                            int ay, uint8_t* dst, uint8_t* tmp)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int x0 = x - ax < 0 ? 0 : x - ax, x1 = x - ax + kw - 1 >= w ? w - 1 : x - ax + kw - 1;
            for (int c = 0; c < cn; c++) {
                int best = is_dilate ? 0 : 255;
                for (int xx = x0; xx <= x1; xx++) {
                    int v = src[((size_t)y * w + xx) * cn + c];
                    if (is_dilate ? v > best : v < best) best = v;
                }
                tmp[((size_t)y * w + x) * cn + c] = (uint8_t)best;
            }
        }
    size_t rowb = (size_t)w * cn;
    for (int y = 0; y < h; y++) {
        int y0 = y - ay < 0 ? 0 : y - ay, y1 = y - ay + kh - 1 >= h ? h - 1 : y - ay + kh - 1;
        uint8_t* d = dst + (size_t)y * rowb;
        if (y0 > y1) { memset(d, is_dilate ? 0 : 255, rowb); continue; }
        memcpy(d, tmp + (size_t)y0 * rowb, rowb);
        for (int yy = y0 + 1; yy <= y1; yy++) {
            const uint8_t* t = tmp + (size_t)yy * rowb;
            if (is_dilate) { for (size_t i = 0; i < rowb; i++) if (t[i] > d[i]) d[i] = t[i]; }
            else { for (size_t i = 0; i < rowb; i++) if (t[i] < d[i]) d[i] = t[i]; }
        }
    }
}

static void morph_basic(int is_dilate, const uint8_t* src, int w, int h, int cn, const uint8_t* kernel,
                        int kw, int kh, int ax, int ay, int iterations, uint8_t* dst, int allow_fast)
{
    size_t n = (size_t)w * h * cn;
    uint8_t rect3[9];
    uint8_t* kbuf = NULL;
    if (kernel == NULL || kw * kh == 0) {
        kw = kh = 1 + iterations * 2;
        ax = ay = iterations;
        iterations = 1;
        kbuf = (uint8_t*)malloc((size_t)kw * kh);
        memset(kbuf, 1, (size_t)kw * kh);
        kernel = kbuf;
        (void)rect3;
    }
    if (ax < 0) ax = kw / 2;
    if (ay < 0) ay = kh / 2;
    if (iterations == 0 || kw * kh == 1) {
        if (dst != src) memmove(dst, src, n);
        free(kbuf);
        return;
    }
    int allones = 1;
    for (int i = 0; i < kw * kh; i++) allones &= kernel[i] != 0;
    if (iterations > 1 && allones) {
        ax *= iterations;
        ay *= iterations;
        kw = kw + (iterations - 1) * (kw - 1);
        kh = kh + (iterations - 1) * (kh - 1);
        free(kbuf);
        kbuf = (uint8_t*)malloc((size_t)kw * kh);
        memset(kbuf, 1, (size_t)kw * kh);
        kernel = kbuf;
        iterations = 1;
    }
    uint8_t* a = (uint8_t*)malloc(n);
    uint8_t* b = (uint8_t*)malloc(n);
    uint8_t* tmp = (allones && allow_fast) ? (uint8_t*)malloc(n) : NULL;
    memcpy(a, src, n);
    for (int it = 0; it < iterations; it++) {
        if (tmp) morph_rect_pass(is_dilate, a, w, h, cn, kw, kh, ax, ay, b, tmp);
        else morph_pass(is_dilate, a, w, h, cn, kernel, kw, kh, ax, ay, b);
        uint8_t* t = a; a = b; b = t;
    }
    memcpy(dst, a, n);
    free(a); free(b); free(tmp); free(kbuf);
}

/* src/dst: tightly packed h*w*cn.  ax/ay < 0 => centre.  fast: 1 allows the separable path. */
ORC_API int orc_morph_u8(int op, const uint8_t* src, int w, int h, int cn, const uint8_t* kernel, int kw,
                         int kh, int ax, int ay, int iterations, uint8_t* dst, int fast)
{
    size_t n = (size_t)w * h * cn;
    if (op == ORC_ERODE || op == ORC_DILATE) {
        morph_basic(op == ORC_DILATE, src, w, h, cn, kernel, kw, kh, ax, ay, iterations, dst, fast);
        return 0;
    }
    uint8_t* t = (uint8_t*)malloc(n);
    if (op == ORC_OPEN) {
        morph_basic(0, src, w, h, cn, kernel, kw, kh, ax, ay, iterations, t, fast);
        morph_basic(1, t, w, h, cn, kernel, kw, kh, ax, ay, iterations, dst, fast);
    } else if (op == ORC_CLOSE) {
        morph_basic(1, src, w, h, cn, kernel, kw, kh, ax, ay, iterations, t, fast);
        morph_basic(0, t, w, h, cn, kernel, kw, kh, ax, ay, iterations, dst, fast);
    } else if (op == ORC_GRADIENT) {
        uint8_t* e = (uint8_t*)malloc(n);
        morph_basic(0, src, w, h, cn, kernel, kw, kh, ax, ay, iterations, e, fast);
        morph_basic(1, src, w, h, cn, kernel, kw, kh, ax, ay, iterations, t, fast);
        for (size_t i = 0; i < n; i++) dst[i] = sat_u8((int)t[i] - (int)e[i]);
        free(e);
    } else { free(t); return -1; }
    free(t);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a10  connected components with stats (north-star CCL; target
 *      cv2.connectedComponentsWithStats(mask, connectivity=8, ltype=CV_32S)).
 *      OpenCV: imgproc/src/connectedcomponents.cpp.  The default 8-way algorithm scans 2x2
 *      blocks (aligned to even rows/cols) in raster order, gives a block a fresh provisional
 *      label when no already-visited neighbouring block is 8-connected to it, merges with
 *      union-find keeping the smaller label as root, then flattenL() renumbers roots
 *      consecutively in increasing provisional-label order.  All foreground pixels of a 2x2
 *      block are mutually 8-adjacent, so one label per block suffices.  block = 2 restates
 *      that (Grana BBDT / Bolelli Spaghetti); block = 1 restates the SAUF (CCL_WU) variant,
 *      which issues provisional labels per pixel.  CCStatsOp: stats rows [left, top, width,
 *      height, area] incl. label 0 = background; centroids = integer coordinate sums / area
 *      in double.
 * ---------------------------------------------------------------------------------------- */
static int uf_find(int* P, int i) { while (P[i] < i) i = P[i]; return i; }
static int uf_merge(int* P, int i, int j)
{
    i = uf_find(P, i);
    j = uf_find(P, j);
    if (i < j) { P[j] = i; return i; }
    P[i] = j;
    return j;
}

ORC_API int orc_ccl_u8(const uint8_t* src, size_t sstride, int w, int h, int block, int32_t* labels,
                       int32_t* stats, double* centroids, int max_k)
{
    if (block != 1 && block != 2) return -1;
    int bw = (w + block - 1) / block, bh = (h + block - 1) / block;
    size_t nb = (size_t)bw * bh;
    int* bl = (int*)calloc(nb ? nb : 1, sizeof(int)); /* provisional label per block, 0 = empty */
    int* P = (int*)malloc((nb + 1) * sizeof(int));
    int count = 1;
    P[0] = 0;
#define FG(yy, xx) ((yy) >= 0 && (yy) < h && (xx) >= 0 && (xx) < w && src[(size_t)(yy)*sstride + (xx)] != 0)
    for (int by = 0; by < bh; by++)
        for (int bx = 0; bx < bw; bx++) {
            int y0 = by * block, x0 = bx * block;
            int any = 0;
            for (int dy = 0; dy < block; dy++)
                for (int dx = 0; dx < block; dx++) any |= FG(y0 + dy, x0 + dx);
            if (!any) continue;
            int lab = 0;
            /* visited neighbour blocks: W, NW, N, NE */
            static const int nby[4] = {0, -1, -1, -1}, nbx[4] = {-1, -1, 0, 1};
            for (int k = 0; k < 4; k++) {
                int qy = by + nby[k], qx = bx + nbx[k];
                if (qy < 0 || qx < 0 || qx >= bw) continue;
                int ql = bl[(size_t)qy * bw + qx];
                if (!ql) continue;
                /* is some fg pixel of this block 8-adjacent to some fg pixel of block q? */
                int conn = 0;
                for (int dy = 0; dy < block && !conn; dy++)
                    for (int dx = 0; dx < block && !conn; dx++) {
                        if (!FG(y0 + dy, x0 + dx)) continue;
                        for (int ey = 0; ey < block && !conn; ey++)
                            for (int ex = 0; ex < block && !conn; ex++) {
                                int py = qy * block + ey, px = qx * block + ex;
                                if (!FG(py, px)) continue;
                                if (abs(py - (y0 + dy)) <= 1 && abs(px - (x0 + dx)) <= 1) conn = 1;
                            }
                    }
                if (!conn) continue;
                lab = lab ? uf_merge(P, lab, ql) : uf_find(P, ql);
            }
            if (!lab) { lab = count; P[count] = count; count++; }
            bl[(size_t)by * bw + bx] = lab;
        }
    /* flattenL */
    int k = 1;
    for (int i = 1; i < count; i++) {
        if (P[i] < i) P[i] = P[P[i]];
        else P[i] = k++;
    }
    int nlabels = k;
    int ns = nlabels < max_k ? nlabels : max_k;
    uint64_t* sx = (uint64_t*)calloc(ns ? ns : 1, sizeof(uint64_t));
    uint64_t* sy = (uint64_t*)calloc(ns ? ns : 1, sizeof(uint64_t));
    if (stats)
        for (int l = 0; l < ns; l++) {
            stats[l * 5 + 0] = INT_MAX; stats[l * 5 + 1] = INT_MAX;
            stats[l * 5 + 2] = INT_MIN; stats[l * 5 + 3] = INT_MIN; stats[l * 5 + 4] = 0;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int l = 0;
            if (src[(size_t)y * sstride + x]) l = P[bl[(size_t)(y / block) * bw + x / block]];
            if (labels) labels[(size_t)y * w + x] = l;
            if (stats && l < ns) {
                int32_t* r = stats + l * 5;
                if (x < r[0]) r[0] = x;
                if (x > r[2]) r[2] = x;
                if (y < r[1]) r[1] = y;
                if (y > r[3]) r[3] = y;
                r[4]++;
                sx[l] += (uint64_t)x;
                sy[l] += (uint64_t)y;
            }
        }
    if (stats)
        for (int l = 0; l < ns; l++) {
            int32_t* r = stats + l * 5;
            r[2] = (int32_t)((uint32_t)r[2] - (uint32_t)r[0] + 1u);
            r[3] = (int32_t)((uint32_t)r[3] - (uint32_t)r[1] + 1u);
            if (centroids) {
                double area = (double)(uint32_t)r[4];
                centroids[l * 2 + 0] = (double)sx[l] / area;
                centroids[l * 2 + 1] = (double)sy[l] / area;
            }
        }
#undef FG
    free(sx); free(sy); free(bl); free(P);
    return nlabels;
}

/* ------------------------------------------------------------------------------------------
 * The red_buoy / bins chain in one call, for the cpu_baseline leg of bench.py and for the
 * end-to-end parity tests: modules/red_buoy.py:21-38 (bgr_to_lab -> range_threshold on one LAB
 * channel -> morph_remove_noise -> morph_close_holes, 5x5 rect) and modules/bins.py:13-27
 * (BGR2HSV -> inRange C3 -> morph_remove_noise), followed by the north-star CCL.
 * mode 0 = LAB, 1 = HSV, 2 = GRAY (channel 0 only).  n_ops morphology ops (ORC_*), all with the
 * same kw x kh all-ones kernel.  Any output pointer may be NULL.
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_chain_u8(const uint8_t* bgr, int w, int h, int mode, const int32_t* lo, const int32_t* hi,
                         const int32_t* ops, int n_ops, int kw, int kh, int block, uint8_t* threshed,
                         uint8_t* cleaned, int32_t* labels, int32_t* stats, double* centroids, int max_k)
{
    size_t n = (size_t)w * h;
    uint8_t* conv = (uint8_t*)malloc(n * 3);
    uint8_t* m0 = (uint8_t*)malloc(n);
    uint8_t* m1 = (uint8_t*)malloc(n);
    uint8_t* kern = (uint8_t*)malloc((size_t)kw * kh);
    memset(kern, 1, (size_t)kw * kh);
    if (mode == 0) orc_bgr2lab_u8(bgr, (size_t)w * 3, w, h, conv, (size_t)w * 3);
    else if (mode == 1) orc_bgr2hsv_u8(bgr, (size_t)w * 3, w, h, conv, (size_t)w * 3);
    else orc_bgr2gray_u8(bgr, (size_t)w * 3, w, h, conv, (size_t)w);
    if (mode == 2) orc_inrange_u8(conv, (size_t)w, w, h, 1, lo, hi, m0, (size_t)w);
    else orc_inrange_u8(conv, (size_t)w * 3, w, h, 3, lo, hi, m0, (size_t)w);
    if (threshed) memcpy(threshed, m0, n);
    for (int i = 0; i < n_ops; i++) {
        orc_morph_u8(ops[i], m0, w, h, 1, kern, kw, kh, -1, -1, 1, m1, 1);
        uint8_t* t = m0; m0 = m1; m1 = t;
    }
    if (cleaned) memcpy(cleaned, m0, n);
    int k = 0;
    if (block) k = orc_ccl_u8(m0, (size_t)w, w, h, block, labels, stats, centroids, max_k);
    free(conv); free(m0); free(m1); free(kern);
    return k;
}

/* ------------------------------------------------------------------------------------------
 * a9  outer_contours / all_contours — utils/feature.py:5-40
 *     (cv2.findContours, RETR_EXTERNAL / RETR_LIST, CHAIN_APPROX_NONE / CHAIN_APPROX_SIMPLE, offset 0)
 *     OpenCV: imgproc/src/contours.cpp cvFindNextContour() + icvFetchContour() (Suzuki-Abe border
 *     following on a signed-char work image padded by one background pixel):
 *       pixel values 0 = background, 1 = unvisited foreground, 2 = visited, 2|-128 = visited with the
 *       east neighbour examined and empty.  Raster scan; at a value change prev -> p: (prev == 0, p == 1)
 *       starts an outer border at x; (p == 0, prev >= 1) starts a hole border at x-1; RETR_EXTERNAL skips
 *       holes and any outer border whose last marked pixel to the left on the row (lnbd) is positive.
 *       Tracing: first neighbour clockwise from west (outer) / east (hole); then repeatedly the first
 *       non-zero neighbour counter-clockwise after the direction we came from; a point is emitted when
 *       the chain direction changes (SIMPLE) or always (NONE); ends when back at the start pixel moving to
 *       the same second pixel.  Contours are returned newest first (each is pushed at the list head).
 *     mode: 0 = RETR_EXTERNAL, 1 = RETR_LIST.  method: 1 = CHAIN_APPROX_NONE, 2 = CHAIN_APPROX_SIMPLE.
 *     Output: points (x,y) int32 pairs of all contours back to back in *return* order, counts[k] points
 *     of contour k, is_hole[k]; returns the number of contours, or -(needed) if a capacity is too small
 *     (needed = contours if max_contours is short, else points).
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_find_contours(const uint8_t* src, size_t sstride, int w, int h, int mode, int method, int32_t* points,
                              long max_points, int32_t* counts, uint8_t* is_hole_out, int max_contours, long* total_points)
{
    const int W = w + 2, H = h + 2;
    const int step = W;
    signed char* img = (signed char*)calloc((size_t)W * H, 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(size_t)(y + 1) * W + x + 1] = src[(size_t)y * sstride + x] ? 1 : 0;
    const int dx8[8] = {1, 1, 0, -1, -1, -1, 0, 1}, dy8[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    int deltas[16];
    for (int k = 0; k < 8; k++) deltas[k] = deltas[k + 8] = dy8[k] * step + dx8[k];
    /* discovery-order storage; reversed at the end */
    long cap_pts = 1024, npts = 0;
    int cap_c = 64, nc = 0;
    int32_t* pts = (int32_t*)malloc(sizeof(int32_t) * 2 * cap_pts);
    long* first = (long*)malloc(sizeof(long) * cap_c);
    int* cnt = (int*)malloc(sizeof(int) * cap_c);
    uint8_t* hole = (uint8_t*)malloc(cap_c);
#define PUSH_PT(px, py) do { if (npts == cap_pts) { cap_pts *= 2; pts = (int32_t*)realloc(pts, sizeof(int32_t) * 2 * cap_pts); } \
                             pts[2 * npts] = (px); pts[2 * npts + 1] = (py); npts++; } while (0)
    for (int y = 1; y < H - 1; y++) {
        signed char* row = img + (size_t)y * W;
        int lnbd_x = 0;
        int prev = 0;
        for (int x = 1; x < W; x++) {   /* x == W-1 is the right padding column (always 0): closes the last run */
            int p = row[x];
            if (p == prev) continue;
            int is_hole = 0;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1) goto resume_scan;
                if (prev & -2) lnbd_x = x - 1;
                is_hole = 1;
            }
            if (mode == 0 && (is_hole || row[lnbd_x] > 0)) goto resume_scan;
            {
                const int ox = x - is_hole;
                lnbd_x = ox;
                if (nc == cap_c) { cap_c *= 2; first = (long*)realloc(first, sizeof(long) * cap_c); cnt = (int*)realloc(cnt, sizeof(int) * cap_c); hole = (uint8_t*)realloc(hole, cap_c); }
                first[nc] = npts;
                hole[nc] = (uint8_t)is_hole;
                /* icvFetchContour */
                signed char* i0 = row + ox;
                signed char *i1, *i3, *i4 = 0;
                int px = ox - 1, py = y - 1;   /* back to unpadded coordinates */
                int s, s_end, prev_s;
                s_end = s = is_hole ? 0 : 4;
                do { s = (s - 1) & 7; i1 = i0 + deltas[s]; } while (*i1 == 0 && s != s_end);
                if (s == s_end) {            /* single pixel */
                    *i0 = (signed char)(2 | -128);
                    PUSH_PT(px, py);
                } else {
                    i3 = i0;
                    prev_s = s ^ 4;
                    for (;;) {
                        s_end = s;
                        while (s < 15) { i4 = i3 + deltas[++s]; if (*i4 != 0) break; }
                        s &= 7;
                        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (signed char)(2 | -128);
                        else if (*i3 == 1) *i3 = 2;
                        if (s != prev_s || method == 1) { PUSH_PT(px, py); prev_s = s; }
                        px += dx8[s];
                        py += dy8[s];
                        if (i4 == i0 && i3 == i1) break;
                        i3 = i4;
                        s = (s + 4) & 7;
                    }
                }
                cnt[nc] = (int)(npts - first[nc]);
                nc++;
                p = row[x];
            }
        resume_scan:
            prev = p;
            if (prev & -2) lnbd_x = x;
        }
    }
#undef PUSH_PT
    if (total_points) *total_points = npts;
    int rc = nc;
    if (nc > max_contours) rc = -nc;
    else if (npts > max_points) rc = (int)-npts;
    else {
        long o = 0;
        for (int k = nc - 1, j = 0; k >= 0; k--, j++) {   /* newest first */
            memcpy(points + 2 * o, pts + 2 * first[k], sizeof(int32_t) * 2 * cnt[k]);
            o += cnt[k];
            counts[j] = cnt[k];
            if (is_hole_out) is_hole_out[j] = hole[k];
        }
    }
    free(pts); free(first); free(cnt); free(hole); free(img);
    return rc;
}

/* cv2.moments on a contour (imgproc/src/moments.cpp contourMoments) — m00, m10, m01 only — and
 * cv2.contourArea(oriented=False): utils/feature.py:240-265 */
ORC_API void orc_contour_moments(const int32_t* pts, int n, double* m00, double* m10, double* m01, double* area)
{
    double a00 = 0, a10 = 0, a01 = 0;
    if (n > 0) {
        double xi_1 = pts[2 * (n - 1)], yi_1 = pts[2 * (n - 1) + 1];
        for (int i = 0; i < n; i++) {
            double xi = pts[2 * i], yi = pts[2 * i + 1];
            double dxy = xi_1 * yi - xi * yi_1;
            a00 += dxy;
            a10 += dxy * (xi_1 + xi);
            a01 += dxy * (yi_1 + yi);
            xi_1 = xi; yi_1 = yi;
        }
    }
    if (area) *area = fabs(a00 * 0.5);
    if (fabs(a00) > 1.1920929e-07) {
        double s2 = a00 > 0 ? 0.5 : -0.5, s6 = a00 > 0 ? 1.0 / 6 : -1.0 / 6;
        *m00 = a00 * s2; *m10 = a10 * s6; *m01 = a01 * s6;
    } else { *m00 = *m10 = *m01 = 0; }
}
