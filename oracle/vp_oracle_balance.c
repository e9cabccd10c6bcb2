/*
 * vp_oracle_balance.c — CPU ORACLE, part 2: the colour-balance entry of the reference and the 8-bit HSV -> BGR
 * conversion it needs.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as vp_oracle.c): imported by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg, never by the product.  PARITY UNPINNED: the reference's source for this path
 * (utils/color_correction/color_balance.cpp) includes <opencv2/opencv.hpp> and misc/utils.h, neither present here,
 * so it cannot be compiled; it has no tests or vectors.  What follows restates its arithmetic statement by
 * statement (file:line cited at each step) on top of this oracle's BGR2HSV and the OpenCV HSV2BGR restated below.
 *
 * Deliberate, documented departures from a literal reading (each one is a place where the reference's behaviour is
 * undefined or crashes):
 *   - `abs(local_avg - avg)` (cpp:472) is taken as fabs() (with only <cmath> in scope the call may resolve to the
 *     integer abs() and truncate; for the default 1x1 tiling the difference is ~1e-12 either way);
 *   - (unsigned char) casts of doubles outside [0,255] or NaN (cpp:584-588 with degenerate ranges) follow what gcc
 *     emits on x86-64: cvttsd2si to int32 (0x80000000 when out of range / NaN), low 8 bits kept;
 *   - integer division by zero in the HSV stretch (cpp:654-655, s_max == s_min) kills the reference with SIGFPE;
 *     here the channel is left as clipped;
 *   - tilings that do not divide the frame (cpp:441-446) make the reference wrap into the next row and process
 *     pixels twice; restated literally (sequential block order) so the behaviour is at least inspectable.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

void orc_bgr2hsv_u8(const uint8_t* src, size_t sstride, int w, int h, uint8_t* dst, size_t dstride);

static inline uint8_t sat_round_u8(float x)   /* cv::saturate_cast<uchar>(float): cvRound (half to even), clamp */
{
    long v = lrintf(x);
    return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

/* cv2.cvtColor(COLOR_HSV2BGR) on 8-bit input, hrange 180 — OpenCV imgproc/src/color_hsv.simd.hpp HSV2RGB_b:
 * h, s/255, v/255 as float32 -> HSV2RGB_f -> *255 -> saturate_cast.  HSV2RGB_f has two arithmetic forms:
 *   variant 0  the universal-intrinsics form used for all full vectors of a row on SIMD builds (HSV2RGB_simd):
 *              tab1 = v - v*s, tab2 = v - (v*s)*h, tab3 = (v - v*s) + (v*s)*h, sector = trunc(h) mod 6;
 *   variant 1  the scalar tail form (HSV2RGB_native): tab1 = v*(1-s), tab2 = v*(1-s*h), tab3 = v*(1-s*(1-h)).
 * They differ in float rounding only; after *255 and rounding to u8 they disagree on a small set of (h,s,v)
 * (tests count it).  Variant 0 is what the GPU implements. */
static void hsv2bgr_px(int H, int S, int V, int variant, uint8_t* out)
{
    const float hscale = 6.f / 180.f;
    float h = (float)H, s = (float)S * (1.f / 255.f), v = (float)V * (1.f / 255.f);
    float tab[4];
    int sector;
    static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    if (variant == 0) {
        h = h * hscale;
        const float pre = truncf(h);
        h = h - pre;
        const float vs = v * s;
        const float vsh = vs * h;
        tab[0] = v;
        tab[1] = v - vs;
        tab[2] = v - vsh;
        tab[3] = (v - vs) + vsh;
        float sec = truncf(pre * (1.0f / 6.0f));
        sec = pre - sec * 6.0f;
        sector = (int)sec;
    } else {
        if (S == 0) {
            out[0] = out[1] = out[2] = sat_round_u8(v * 255.f);
            return;
        }
        h *= hscale;
        h = fmodf(h, 6.f);
        sector = (int)floorf(h);
        h -= (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        tab[0] = v;
        tab[1] = v * (1.f - s);
        tab[2] = v * (1.f - s * h);
        tab[3] = v * (1.f - s * (1.f - h));
    }
    out[0] = sat_round_u8(tab[sector_data[sector][0]] * 255.f);
    out[1] = sat_round_u8(tab[sector_data[sector][1]] * 255.f);
    out[2] = sat_round_u8(tab[sector_data[sector][2]] * 255.f);
}

ORC_API void orc_hsv2bgr_u8(const uint8_t* src, size_t sstride, int w, int h, uint8_t* dst, size_t dstride, int variant)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++, s += 3, d += 3) hsv2bgr_px(s[0], s[1], s[2], variant, d);
    }
}

/* ---- utils/color_correction/color_balance.cpp ------------------------------------------------------------ */

static inline uint8_t cast_u8(double v)   /* (unsigned char)double as gcc/x86-64 does it: cvttsd2si, low byte */
{
    int32_t i;
    if (!(v > -2147483649.0 && v < 2147483648.0)) i = INT32_MIN;   /* NaN and out-of-range: "integer indefinite" */
    else i = (int32_t)v;
    return (uint8_t)(i & 0xff);
}
static inline uint8_t constrain255(double val)   /* cpp:13-23 constrain(val, 0, 255) */
{
    if (val < 0) return 0;
    if (val > 255) return 255;
    return cast_u8(val);
}
static void clip_channel(uint8_t* c, size_t n, int lo, int hi)   /* cpp:25-45 (min/max arrive as unsigned char) */
{
    const uint8_t l = (uint8_t)lo, h = (uint8_t)hi;
    for (size_t i = 0; i < n; i++) {
        if (c[i] < l) c[i] = l;
        else if (c[i] > h) c[i] = h;
    }
}
static void percentile_min_max(const uint8_t* c, size_t n, float lower, float upper, int* mn, int* mx)   /* cpp:111-139 */
{
    int low_bound = (int)(lower * (float)n);
    int high_bound = (int)n - (int)(upper * (float)n);
    int counts[256] = {0};
    for (size_t i = 0; i < n; i++) counts[c[i]]++;
    *mn = 0; *mx = 255;
    for (int i = 0; i < 256; i++) {
        if (low_bound < counts[i]) { *mn = i; break; }
        low_bound -= counts[i];
    }
    for (int i = 255; i >= 0; i--) {
        if (high_bound < counts[i]) { *mx = i; break; }
        high_bound -= counts[i];
    }
}
static double mean_u8(const uint8_t* c, size_t n)   /* cv::mean: exact integer sum / n */
{
    uint64_t s = 0;
    for (size_t i = 0; i < n; i++) s += c[i];
    return (double)s / (double)n;
}

/* ---- HSI stage (cpp:141-341, 678-775) ---------------------------------------------------------------------
 * float/double mixing follows the C++ expressions operand by operand ((float)x op float -> float; anything touching a
 * double literal or a <cmath> call -> double; stores to float arrays round to float).  `sqrt` is taken as the double
 * function on the (exactly representable) float radicand.  glibc's acos/cos are correctly rounded in almost all cases;
 * a device libm may differ in the last bit, which is why GPU parity for this stage is stated with a tolerance of 1. */
static int feq(float a, float b) { return fabs(a - b) < 0.000001; }   /* cpp:9-11: float difference, double compare */
static uint8_t uchar_clip(float f)   /* cpp:155-164; (int) of NaN / out-of-range floats as x86 cvttss2si: INT_MIN */
{
    int n = (f > -2147483904.f && f < 2147483648.f) ? (int)f : INT32_MIN;
    if (n < 0) n = 0; else if (n > 255) n = 255;
    return (uint8_t)n;
}
static void clip_channel_f(float* c, size_t n, float mn, float mx)   /* cpp:47-69 */
{
    for (size_t i = 0; i < n; i++) {
        if (c[i] < mn) c[i] = mn;
        else if (c[i] > mx) c[i] = mx;
        else if (isnan(c[i])) c[i] = mn;
        else if (isinf(c[i])) c[i] = mx;
    }
}
static void rgb_to_hsi_px(uint8_t r, uint8_t g, uint8_t b, float* H, float* S, float* I)   /* cpp:166-215 */
{
    uint8_t mn = 255;
    *I = (float)(((float)r + (float)g + (float)b) / 3.);
    if (r < mn) mn = r;
    if (g < mn) mn = g;
    if (b < mn) mn = b;
    if (*I > 0) *S = (float)(1. - ((float)mn / *I));
    else *S = 0;
    const float rad = (float)r * r + (float)g * g + (float)b * b - (float)(r * g) - (float)(r * b) - (float)(g * b);
    *H = (float)acos(((float)r - (0.5 * g) - (0.5 * b)) / sqrt((double)rad));
    if (b > g) *H = (float)((M_PI * 2) - *H);
}
static int cmp_f32(const void* a, const void* b) { const float x = *(const float*)a, y = *(const float*)b; return (x > y) - (x < y); }
static void hsi_to_rgb_px(float h, float s, float i, uint8_t* r, uint8_t* g, uint8_t* b)   /* cpp:261-306 */
{
    if (feq(h, 0)) {
        *r = uchar_clip(i + 2 * i * s); *g = uchar_clip(i - i * s); *b = uchar_clip(i - i * s);
    } else if (0. < h && h < 2. * M_PI / 3.) {
        *r = uchar_clip((float)(i + i * s * cos(h) / cos(M_PI / 3. - h)));
        *g = uchar_clip((float)(i + i * s * (1 - cos(h) / cos(M_PI / 3. - h))));
        *b = uchar_clip(i - i * s);
    } else if (feq(h, (float)(2. * M_PI / 3.))) {
        *r = uchar_clip(i - i * s); *g = uchar_clip(i + 2 * i * s); *b = uchar_clip(i - i * s);
    } else if (2. * M_PI / 3. < h && h < 4. * M_PI / 3.) {
        *r = uchar_clip(i - i * s);
        *g = uchar_clip((float)(i + i * s * cos(h - 2. * M_PI / 3.) / cos(M_PI - h)));
        *b = uchar_clip((float)(i + i * s * (1 - cos(h - 2. * M_PI / 3.) / cos(M_PI - h))));
    } else if (feq(h, (float)(4. * M_PI / 3.))) {
        *r = uchar_clip(i - i * s); *g = uchar_clip(i - i * s); *b = uchar_clip(i + 2 * i * s);
    } else {
        *r = uchar_clip((float)(i + i * s * (1 - cos(h - 4. * M_PI / 3.) / cos(5. * M_PI / 3. - h))));
        *g = uchar_clip(i - i * s);
        *b = uchar_clip((float)(i + i * s * cos(h - 4. * M_PI / 3.) / cos(5. * M_PI / 3. - h)));
    }
}
/* cpp:678-775.  percentile_min_max_qselect (cpp:141-153) returns the low_bound-th and high_bound-th smallest values
 * (0-based) - quickselect with random pivots is an exact order statistic - restated with a sort. */
static void hsi_stage(uint8_t* rc, uint8_t* gc, uint8_t* bc, size_t n)
{
    float* H = (float*)malloc(n * 4);
    float* S = (float*)malloc(n * 4);
    float* I = (float*)malloc(n * 4);
    float* tmp = (float*)malloc(n * 4);
    for (size_t k = 0; k < n; k++) rgb_to_hsi_px(rc[k], gc[k], bc[k], &H[k], &S[k], &I[k]);
    clip_channel_f(H, n, 0.f, (float)(2. * M_PI));
    clip_channel_f(S, n, 0.f, 1.f);
    clip_channel_f(I, n, 0.f, 255.f);
    const int low_bound = (int)(0.002f * (float)n), high_bound = (int)(0.998f * (float)n);
    memcpy(tmp, S, n * 4); qsort(tmp, n, 4, cmp_f32);
    const float s_min = tmp[low_bound], s_max = tmp[high_bound < (int)n ? high_bound : (int)n - 1];
    clip_channel_f(S, n, s_min, s_max);
    memcpy(tmp, I, n * 4); qsort(tmp, n, 4, cmp_f32);
    const float i_min = tmp[low_bound], i_max = tmp[high_bound < (int)n ? high_bound : (int)n - 1];
    clip_channel_f(I, n, i_min, i_max);
    const float s_mult = (float)(1. / (s_max - s_min)), i_mult = (float)(255. / (i_max - i_min));
    for (size_t k = 0; k < n; k++) { S[k] = (S[k] - s_min) * s_mult; I[k] = (I[k] - i_min) * i_mult; }
    clip_channel_f(S, n, 0.f, 1.f);
    clip_channel_f(I, n, 0.f, 255.f);
    for (size_t k = 0; k < n; k++) hsi_to_rgb_px(H[k], S[k], I[k], &rc[k], &gc[k], &bc[k]);
    free(H); free(S); free(I); free(tmp);
}

/* process_frame(arr, height, width, depth = 3, ...) — cpp:343-780.
 * mean_mode 0: the running mean of cpp:452-467 literally (avg += (x - avg) / count, row-major inside the tile);
 * mean_mode 1: the same quantity as an exact sum / count (what a parallel implementation computes; differs from the
 * running mean by rounding noise of ~1e-13 relative).  hsv_variant: see orc_hsv2bgr_u8. */
ORC_API int orc_color_balance(uint8_t* arr, size_t height, size_t width, int equalize_rgb, int rgb_contrast_correct,
                              int hsv_contrast_correct, int hsi_contrast_correct, int rgb_extrema_clipping,
                              int adaptive_cast_correction, int horizontal_blocks, int vertical_blocks, int mean_mode,
                              int hsv_variant)
{
    if (horizontal_blocks <= 0 || vertical_blocks <= 0 || !height || !width) return -1;
    const size_t n = height * width;
    uint8_t* bc = (uint8_t*)calloc(n, 1);
    uint8_t* gc = (uint8_t*)calloc(n, 1);
    uint8_t* rc = (uint8_t*)calloc(n, 1);
    if (!bc || !gc || !rc) { free(bc); free(gc); free(rc); return -3; }
    for (size_t i = 0; i < n; i++) { bc[i] = arr[3 * i]; gc[i] = arr[3 * i + 1]; rc[i] = arr[3 * i + 2]; }   /* cv::split cpp:372 */

    double r_min, r_max, g_min, g_max, b_min, b_max;
    if (rgb_extrema_clipping) {   /* cpp:398-419 */
        int mn, mx;
        percentile_min_max(rc, n, 0.002f, 0.998f, &mn, &mx); r_min = mn; r_max = mx; clip_channel(rc, n, mn, mx);
        percentile_min_max(gc, n, 0.002f, 0.998f, &mn, &mx); g_min = mn; g_max = mx; clip_channel(gc, n, mn, mx);
        percentile_min_max(bc, n, 0.002f, 0.998f, &mn, &mx); b_min = mn; b_max = mx; clip_channel(bc, n, mn, mx);
    } else {                      /* cv::minMaxLoc cpp:421-425 */
        uint8_t lo[3] = {255, 255, 255}, hi[3] = {0, 0, 0};
        const uint8_t* ch[3] = {rc, gc, bc};
        for (int k = 0; k < 3; k++)
            for (size_t i = 0; i < n; i++) { if (ch[k][i] < lo[k]) lo[k] = ch[k][i]; if (ch[k][i] > hi[k]) hi[k] = ch[k][i]; }
        r_min = lo[0]; r_max = hi[0]; g_min = lo[1]; g_max = hi[1]; b_min = lo[2]; b_max = hi[2];
    }
    const double r_avg = mean_u8(rc, n), g_avg = mean_u8(gc, n), b_avg = mean_u8(bc, n);   /* cpp:427-429 */

    if (equalize_rgb) {   /* cpp:441-543 */
        const int block_width = (int)width / horizontal_blocks, block_height = (int)height / vertical_blocks;
        if (width % (size_t)horizontal_blocks != 0) ++horizontal_blocks;
        if (height % (size_t)vertical_blocks != 0) ++vertical_blocks;
        for (int by = 0; by < vertical_blocks; ++by)
            for (int bx = 0; bx < horizontal_blocks; ++bx) {
                double lr = 0, lg = 0, lb = 0;
                int count = 0;
                uint64_t sr = 0, sg = 0, sb = 0;
                for (int j = 0; j < block_height; ++j)
                    for (int i = 0; i < block_width; ++i) {
                        const size_t ci = ((size_t)by * block_height + j) * width + ((size_t)bx * block_width + i);
                        if (ci >= n) break;
                        ++count;
                        lr += (rc[ci] - lr) / count;
                        lg += (gc[ci] - lg) / count;
                        lb += (bc[ci] - lb) / count;
                        sr += rc[ci]; sg += gc[ci]; sb += bc[ci];
                    }
                if (mean_mode == 1 && count > 0) { lr = (double)sr / count; lg = (double)sg / count; lb = (double)sb / count; }
                if (fabs(lr - r_avg) > r_avg / 6 || fabs(lb - b_avg) > b_avg / 6 || fabs(lg - g_avg) > g_avg / 6) {
                    lr = r_avg; lb = b_avg; lg = g_avg;
                }
                uint8_t *c1, *c2;
                double gain1, gain2;
                if (lr > lg && lr > lb) { c1 = gc; c2 = bc; gain1 = lr / lg; gain2 = lr / lb; }          /* red cast */
                else if (lg > lr && lg > lb) { c1 = rc; c2 = bc; gain1 = lg / lr; gain2 = lg / lb; }     /* green cast */
                else { c1 = rc; c2 = gc; gain1 = lb / lr; gain2 = lb / lg; }                             /* blue cast (and ties) */
                for (int j = 0; j < block_height; ++j)
                    for (int i = 0; i < block_width; ++i) {
                        const size_t ci = ((size_t)by * block_height + j) * width + ((size_t)bx * block_width + i);
                        if (ci >= n) break;
                        if (adaptive_cast_correction) {
                            c1[ci] = constrain255(c1[ci] * (pow((255. - c1[ci]) / 255., 0.25) * (gain1 - 1.) + 1.));
                            c2[ci] = constrain255(c2[ci] * (pow((255. - c2[ci]) / 255., 0.25) * (gain2 - 1.) + 1.));
                        } else {
                            c1[ci] = constrain255(c1[ci] * gain1);
                            c2[ci] = constrain255(c2[ci] * gain2);
                        }
                    }
            }
    }

    if (rgb_contrast_correct) {   /* cpp:545-597 */
        uint8_t *min_c, *mid_c, *max_c;
        int min_min, min_max, mid_min, mid_max, max_min, max_max;
#define SETC(which, chan, lo, hi) do { which##_c = chan; which##_min = (int)(lo); which##_max = (int)(hi); } while (0)
        if (r_avg > g_avg) {
            if (r_avg > b_avg) {
                SETC(max, rc, r_min, r_max);
                if (g_avg > b_avg) { SETC(mid, gc, g_min, g_max); SETC(min, bc, b_min, b_max); }
                else { SETC(mid, bc, b_min, b_max); SETC(min, gc, g_min, g_max); }
            } else { SETC(max, bc, b_min, b_max); SETC(mid, rc, r_min, r_max); SETC(min, gc, g_min, g_max); }
        } else {
            if (g_avg > b_avg) {
                SETC(max, gc, g_min, g_max);
                if (r_avg > b_avg) { SETC(mid, rc, r_min, r_max); SETC(min, bc, b_min, b_max); }
                else { SETC(mid, bc, b_min, b_max); SETC(min, rc, r_min, r_max); }
            } else { SETC(max, bc, b_min, b_max); SETC(mid, gc, g_min, g_max); SETC(min, rc, r_min, r_max); }
        }
#undef SETC
        const double desired_max = (double)((min_max + mid_max + max_max) / 3);
        const double min_ratio = (desired_max - min_min) / (double)(min_max - min_min);
        const double mid_ratio = (desired_max - 0.0) / (double)(mid_max - mid_min);
        const double max_ratio = (max_max - 0.0) / (double)(max_max - max_min);
        for (size_t i = 0; i < n; i++) {
            min_c[i] = cast_u8((min_c[i] - min_min) * min_ratio);
            mid_c[i] = cast_u8((mid_c[i] - mid_min) * mid_ratio);
            max_c[i] = cast_u8((max_c[i] - max_min) * max_ratio);
        }
    }

    if (hsv_contrast_correct) {   /* cpp:599-676 */
        uint8_t* bgr = (uint8_t*)malloc(n * 3);
        uint8_t* hsv = (uint8_t*)malloc(n * 3);
        uint8_t* sc = (uint8_t*)malloc(n);
        uint8_t* vc = (uint8_t*)malloc(n);
        for (size_t i = 0; i < n; i++) { bgr[3 * i] = bc[i]; bgr[3 * i + 1] = gc[i]; bgr[3 * i + 2] = rc[i]; }
        orc_bgr2hsv_u8(bgr, width * 3, (int)width, (int)height, hsv, width * 3);
        for (size_t i = 0; i < n; i++) { sc[i] = hsv[3 * i + 1]; vc[i] = hsv[3 * i + 2]; }
        int s_min, s_max, v_min, v_max;
        percentile_min_max(sc, n, 0.002f, 0.998f, &s_min, &s_max); clip_channel(sc, n, s_min, s_max);
        percentile_min_max(vc, n, 0.002f, 0.998f, &v_min, &v_max); clip_channel(vc, n, v_min, v_max);
        for (size_t i = 0; i < n; i++) {
            if (s_max != s_min) sc[i] = (uint8_t)((((int)sc[i] - s_min) * 255) / (s_max - s_min));
            if (v_max != v_min) vc[i] = (uint8_t)((((int)vc[i] - v_min) * 255) / (v_max - v_min));
            hsv[3 * i + 1] = sc[i]; hsv[3 * i + 2] = vc[i];
        }
        orc_hsv2bgr_u8(hsv, width * 3, (int)width, (int)height, bgr, width * 3, hsv_variant);
        for (size_t i = 0; i < n; i++) { bc[i] = bgr[3 * i]; gc[i] = bgr[3 * i + 1]; rc[i] = bgr[3 * i + 2]; }
        free(bgr); free(hsv); free(sc); free(vc);
    }

    if (hsi_contrast_correct) hsi_stage(rc, gc, bc, n);   /* cpp:678-775 */

    for (size_t i = 0; i < n; i++) { arr[3 * i] = bc[i]; arr[3 * i + 1] = gc[i]; arr[3 * i + 2] = rc[i]; }   /* cv::merge cpp:777 */
    free(bc); free(gc); free(rc);
    return 0;
}

/* ---- cv2.GaussianBlur on 8-bit images (modules/preprocessor.py:110-114, utils/transform.py simple_gaussian_blur) ----------
 * OpenCV >= 4.0 takes the bit-exact fixed-point path for CV_8U (imgproc/src/smooth.dispatch.cpp): the double kernel of
 * getGaussianKernelBitExact (fixed tables for sigma <= 0 and n in {1,3,5,7,9}, otherwise exp(-x^2 / (2 sigma^2)) with
 * sigma = 0.15 n + 0.35 when not given, normalised), converted to 8.8 fixed point with error diffusion towards the centre
 * (getGaussianKernelFixedPoint_ED: the taps sum to exactly 256), a horizontal pass producing exact 8.8 sums, a vertical pass
 * producing 16.16 sums rounded half up to 8 bits; border BORDER_REFLECT_101.  (OpenCV evaluates exp() in its own softfloat; a
 * libm last-bit difference could only matter if tap * 256 fell within 1e-13 of a half.) */
static void gauss_kernel_fixed(int n, double sigma, uint16_t* out)
{
    double k[512];
    static const double t3[] = {0.25, 0.5, 0.25}, t5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625},
                        t7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125},
                        t9[] = {4 / 256., 13 / 256., 30 / 256., 51 / 256., 60 / 256., 51 / 256., 30 / 256., 13 / 256., 4 / 256.};
    const double* tab = NULL;
    if (sigma <= 0) tab = n == 3 ? t3 : n == 5 ? t5 : n == 7 ? t7 : n == 9 ? t9 : NULL;
    if (n == 1) { out[0] = 256; return; }
    if (tab) for (int i = 0; i < n; i++) k[i] = tab[i];
    else {
        const double sx = sigma > 0 ? sigma : (double)n * 0.15 + 0.35;
        const double scale2x = -0.125 / (sx * sx);
        const int n2 = (n - 1) / 2;
        double sum = 0;
        for (int i = 0, x = 1 - n; i < n2; i++, x += 2) { k[i] = exp((double)(x * x) * scale2x); sum += k[i]; }
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
        const long long v0 = llrint(adj);   /* cvRound: half to even */
        err = adj - (double)v0;
        out[i] = out[n - 1 - i] = (uint16_t)v0;
        sum += 2 * v0;
    }
    out[n2] = (uint16_t)(256 - sum);
}
static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}
ORC_API int orc_gaussian_kernel_fixed(int n, double sigma, uint16_t* out)
{
    if (n <= 0 || n > 511 || !(n & 1)) return -1;
    gauss_kernel_fixed(n, sigma, out);
    return 0;
}
ORC_API int orc_gaussian_blur_u8(const uint8_t* src, int w, int h, int cn, int kw, int kh, double sigma1, double sigma2, uint8_t* dst)
{
    if (w <= 0 || h <= 0 || cn < 1 || cn > 4 || kw <= 0 || kh <= 0 || !(kw & 1) || !(kh & 1) || kw > 511 || kh > 511) return -1;
    if (sigma1 < 0) sigma1 = 0;
    if (sigma2 <= 0) sigma2 = sigma1;
    uint16_t kx[512], ky[512];
    gauss_kernel_fixed(kw, sigma1, kx);
    gauss_kernel_fixed(kh, sigma2, ky);
    uint16_t* tmp = (uint16_t*)malloc((size_t)w * h * cn * 2);
    if (!tmp) return -3;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                uint32_t s = 0;
                for (int k = 0; k < kw; k++) s += (uint32_t)kx[k] * src[((size_t)y * w + reflect101(x + k - kw / 2, w)) * cn + c];
                tmp[((size_t)y * w + x) * cn + c] = (uint16_t)s;
            }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                uint32_t s = 0;
                for (int k = 0; k < kh; k++) s += (uint32_t)ky[k] * tmp[((size_t)reflect101(y + k - kh / 2, h) * w + x) * cn + c];
                const uint32_t v = (s + (1u << 15)) >> 16;
                dst[((size_t)y * w + x) * cn + c] = (uint8_t)(v > 255 ? 255 : v);
            }
    free(tmp);
    return 0;
}
