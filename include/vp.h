/*
 * vp.h — C ABI of libvp.so, the MI355X (gfx950) implementation of the per-frame detection hot
 * path of ayf7/cuauv-vision-pipeline: colour convert -> inRange -> erode/dilate -> connected
 * components.  Everything behind this header is hand-written HIP; there is no CPU fallback:
 * every compute entry point returns VP_ERR_HIP when no gfx950 device is usable.
 *
 * Each entry point names the reference interface it replaces (file:line under the reference
 * checkout).  The reference implements these by calling cv2 from Python (utils/color.py,
 * utils/transform.py, utils/feature.py); the Python mirror in
 * cuauv-vision-pipeline_amd/vision/utils binds this ABI with ctypes.
 *
 * Conventions: plain pointers and sizes only.  `_host` pointers are ordinary process memory
 * (staged through the context's device workspace); pointers inside vp_chain_buffers are device
 * (HBM) pointers.  Images are row-major, interleaved channels, uint8 unless stated.
 * All functions return VP_OK (0) or a negative VP_ERR_*; none throws.  A context is
 * thread-compatible (one caller at a time); use one context per stream / camera direction.
 */
#ifndef VP_H
#define VP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VP_OK 0
#define VP_ERR_INVALID (-1)     /* bad argument (NULL, size <= 0, unknown enum) */
#define VP_ERR_HIP (-2)         /* HIP runtime error / no device; see vp_last_error() */
#define VP_ERR_NOMEM (-3)       /* device or host allocation failed */
#define VP_ERR_UNSUPPORTED (-4) /* valid request outside what the kernels cover */

typedef struct vp_ctx vp_ctx;

/* colour conversion codes (values are libvp's own, not cv2's) */
enum { VP_BGR2LAB = 0, VP_BGR2HSV = 1, VP_BGR2GRAY = 2, VP_GRAY2BGR = 3, VP_HSV2BGR = 4, VP_BGR2YCRCB = 5, VP_BGR2HLS = 6 };
/* morphology ops — utils/transform.py:80-164 */
enum { VP_MORPH_ERODE = 0, VP_MORPH_DILATE = 1, VP_MORPH_OPEN = 2, VP_MORPH_CLOSE = 3, VP_MORPH_GRADIENT = 4 };
/* structuring element shapes — cv2.MORPH_RECT / MORPH_CROSS / MORPH_ELLIPSE */
enum { VP_SHAPE_RECT = 0, VP_SHAPE_CROSS = 1, VP_SHAPE_ELLIPSE = 2 };
/* label numbering: VP_CCL_BLOCK2X2 = order of cv2.connectedComponents' default 8-way
 * algorithm (raster order of each component's first 2x2 block); VP_CCL_PIXEL = raster order of
 * each component's first pixel (cv2 CCL_WU / SAUF). */
enum { VP_CCL_PIXEL = 1, VP_CCL_BLOCK2X2 = 2 };

/* ---- context ------------------------------------------------------------------------- */
int vp_version(void);
const char* vp_strerror(int code);
/* Creates a context on HIP device `device` with its own non-blocking stream. NULL on failure
 * (vp_last_error(NULL) tells why). */
/* Devices visible to the process, and the PCI address ("0000:05:00.0") of one: /sys/bus/pci/devices/<address>/numa_node names the
 * host NUMA node next to it (vision/dispatch.py binds each device's feeder threads there).  No context needed. */
int vp_device_count(void);
int vp_device_pci_bus_id(int device, char* out, int len);
vp_ctx* vp_create(int device);
int vp_destroy(vp_ctx* ctx);
const char* vp_last_error(const vp_ctx* ctx);
/* Adopt an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL reverts to
 * the context's own stream. */
int vp_set_stream(vp_ctx* ctx, void* hip_stream);
void* vp_get_stream(vp_ctx* ctx);
int vp_synchronize(vp_ctx* ctx);
/* Options.  VP_OPT_CHAIN_STREAMS (1..4, default 1): vp_chain_run splits a batch into that many sub-batches
 * on internal streams (fork/join around the context's stream); results are identical for every value.
 * Measured on MI355X: no gain over 1 (the kernels of the two halves slow each other down), hence the default.
 * VP_OPT_CCL_LEVELS (1 or 2, default 2): 2 = strip-local components merged by one block per frame, frames that do not fit
 * finished by the one-level kernels; 1 = one-level kernels only.  Results are identical.
 * VP_OPT_CCL_MERGE_CAP (-1 = capacity of the merge block, or a smaller count): strip components per frame above which a frame
 * is handed to the one-level kernels (test hook: 0 sends every non-empty frame there).
 * VP_OPT_FLAT_OPS (0 or 1, default 1): the per-operator kernels (vp_cvt_color_*, vp_inrange_u8_*) use their 16-pixels-per-lane forms
 * whenever rows are packed and pointers 16-B aligned; 0 forces the generic one-pixel-per-thread kernels (the tests run both). */
enum { VP_OPT_CHAIN_STREAMS = 1, VP_OPT_CCL_LEVELS = 2, VP_OPT_CCL_MERGE_CAP = 3, VP_OPT_FLAT_OPS = 4 };
int vp_set_option(vp_ctx* ctx, int option, int value);
/* HIP-event stopwatch on the context's stream (bench.py: roofline.achieved). */
int vp_timer_start(vp_ctx* ctx);
int vp_timer_stop(vp_ctx* ctx, float* elapsed_ms); /* records, synchronises, returns ms */
/* Per-kernel attribution: between vp_profile_begin and vp_profile_end every kernel the context
 * launches is bracketed by HIP events on its stream.  vp_profile_end synchronises and fills
 * total_ms[id] / launches[id] for id < VP_PROF_KERNELS (names: vp_profile_kernel_name). */
#define VP_PROF_KERNELS 15
int vp_profile_begin(vp_ctx* ctx, int max_records);
int vp_profile_end(vp_ctx* ctx, double* total_ms, int32_t* launches);
const char* vp_profile_kernel_name(int id);
/* Copies of the integer tables the kernels use (for parity tests against the oracle):
 * gamma[256] u16, cbrt[3072] u16, sdiv[256] i32, hdiv180[256] i32, lab_coeffs[9] i32. */
int vp_get_tables(uint16_t* gamma, uint16_t* cbrt_tab, int32_t* sdiv, int32_t* hdiv180, int32_t* lab_coeffs);

/* ---- per-operator API, host pointers --------------------------------------------------- */

/* utils/color.py:11-32 `_convert_colorspace` (cv2.cvtColor + cv2.split): bgr_to_lab, bgr_to_hsv,
 * bgr_to_gray, gray_to_bgr, bgr_to_ycrcb, bgr_to_hls (modules/preprocessor.py:66-75), plus HSV2BGR (color_balance.cpp:669).
 * YCrCb is OpenCV's Q14 integer form; HLS is the float32 statement sequence of RGB2HLS_f behind RGB2HLS_b.  src is (h,w,3) (or (h,w) for GRAY2BGR) with `src_stride` bytes per
 * row.  dst_interleaved (tightly packed, may be NULL) receives the converted image; dst_planes[k]
 * (each (h,w) tightly packed, each may be NULL, array may be NULL) receive the split channels. */
int vp_cvt_color_u8(vp_ctx* ctx, int code, const uint8_t* src_host, size_t src_stride, int w, int h,
                    uint8_t* dst_interleaved_host, uint8_t* const* dst_planes_host);

/* Extension (BASELINE north star "LAB floats within 1e-4"; no reference call site converts float images): BGR float32 in
 * [0,1] (h,w,3 tightly packed) -> CIE L*a*b* float32 (L 0..100, a/b about -127..127), analytic sRGB / D65. */
int vp_cvt_bgr2lab_f32(vp_ctx* ctx, const float* src_host, int w, int h, float* dst_host);

/* utils/color_correction/color_balance.hpp:9-14 `process_frame` (modules/color_balance.py:93-110 `balance`,
 * modules/preprocessor.py:87-88): colour-cast equalisation, optional RGB / HSV contrast stretch, 0.2 % extrema clipping.
 * flags = OR of VP_CB_*; the reference's default call is VP_CB_DEFAULT with 1x1 tiles.  Not implemented
 * (VP_ERR_UNSUPPORTED): tilings that do not divide the frame (the reference wraps into the next row and processes
 * pixels twice there).  src and dst are (h,w,3) BGR, tightly packed; dst may equal src. */
enum { VP_CB_EQUALIZE_RGB = 1, VP_CB_RGB_CONTRAST = 2, VP_CB_HSV_CONTRAST = 4, VP_CB_HSI_CONTRAST = 8, VP_CB_EXTREMA_CLIPPING = 16,
       VP_CB_ADAPTIVE_CAST = 32, VP_CB_DEFAULT = 1 | 4 | 16 };
int vp_color_balance_u8(vp_ctx* ctx, const uint8_t* src_host, int w, int h, int flags, int horizontal_blocks, int vertical_blocks,
                        uint8_t* dst_host);
/* Diagnostic: in how many tiles the last colour-balance call on this context had to run the reference's sequential tile mean
 * (cpp:452-467) because its value could have mattered (see csrc/vp_balance.hip); valid until the next call that uses the workspace. */
int vp_color_balance_last_folds(vp_ctx* ctx, int32_t* tiles_folded);
/* Same on device memory for a batch of n frames ((n,h,w,3), packed; dst may equal src); enqueues and returns. */
int vp_color_balance_dev(vp_ctx* ctx, const uint8_t* src_dev, uint8_t* dst_dev, int w, int h, int n_frames, int flags,
                         int horizontal_blocks, int vertical_blocks);

/* utils/color.py:105-121 `range_threshold` / modules/bins.py:16 (cv2.inRange): cn = 1 or 3;
 * lo/hi have cn entries (already rounded to integers); dst is (h,w) 0/255. */
int vp_inrange_u8(vp_ctx* ctx, const uint8_t* src_host, size_t src_stride, int w, int h, int cn,
                  const int32_t* lo, const int32_t* hi, uint8_t* dst_host);
/* cv2.inRange on a CV_32FC1 image (utils/color.py:103 on `dists`). */
int vp_inrange_f32(vp_ctx* ctx, const float* src_host, size_t src_stride_bytes, int w, int h, float lo,
                   float hi, uint8_t* dst_host);

/* utils/color.py:66-103 `thresh_color_distance` arithmetic: d2 = sum_c wts[c]*(f32(p_c)-color[c])^2
 * in float32, channels in order, channel c skipped when bit c of skipmask is set.
 * dist2_out ((h,w) f32) and sqrt_out ((h,w) u8 = uint8(sqrt(d2))) may each be NULL. */
int vp_color_distance_u8(vp_ctx* ctx, const uint8_t* const* planes_host, int w, int h, const float* color,
                         const float* wts, int skipmask, float* dist2_out_host, uint8_t* sqrt_out_host);
/* Order statistics for np.percentile(dists, p) in utils/color.py:98: the k-th and (k+1)-th smallest of n float32 values
 * (exact, by radix selection on the device); v_k1 may be NULL; k+1 is clamped to n-1. */
int vp_order_stats_f32(vp_ctx* ctx, const float* src_host, size_t n, size_t k, float* v_k, float* v_k1);
/* utils/transform.py:27-77 `elliptic_kernel` / `rect_kernel` (cv2.getStructuringElement):
 * out is (kh,kw) 0/1.  Integer geometry on the host, no device needed. */
int vp_structuring_element(int shape, int kw, int kh, uint8_t* out);

/* utils/transform.py:80-164 erode / dilate / morph_remove_noise / morph_close_holes /
 * morph_borders and modules/preprocessor.py:120-129 (cv2.erode / dilate / morphologyEx):
 * src/dst (h,w,cn) tightly packed, cn in 1..4; kernel (kh,kw) non-zero = member, NULL = 3x3 rect;
 * anchor (-1,-1) = centre; border = cv2 default (outside never wins). */
int vp_morph_u8(vp_ctx* ctx, int op, const uint8_t* src_host, int w, int h, int cn, const uint8_t* kernel,
                int kw, int kh, int anchor_x, int anchor_y, int iterations, uint8_t* dst_host);

/* North-star CCL (replaces the cv2.findContours stage of utils/feature.py:5-40 with labels;
 * semantics of cv2.connectedComponentsWithStats(mask, 8, CV_32S)): non-zero = foreground,
 * 8-connectivity.  labels (h,w) i32 may be NULL.  stats (max_labels,5) i32 rows
 * [left, top, width, height, area] and centroids (max_labels,2) f64, row 0 = background; rows
 * beyond max_labels are dropped but *nlabels is always the true count (incl. background). */
int vp_ccl_u8(vp_ctx* ctx, const uint8_t* src_host, size_t src_stride, int w, int h, int numbering,
              int32_t* labels_host, int32_t* stats_host, double* centroids_host, int max_labels,
              int32_t* nlabels);

/* utils/feature.py:5-40 `outer_contours` / `all_contours` (cv2.findContours): mode VP_RETR_EXTERNAL or VP_RETR_LIST,
 * method VP_CHAIN_APPROX_NONE or VP_CHAIN_APPROX_SIMPLE, offset (0,0); non-zero = foreground.  Contours come back in
 * cv2's order (last found first): counts[k] points of contour k, is_hole[k] (may be NULL), points = (x, y) int32 pairs
 * of all contours back to back.  *n_contours / *n_points are always the true totals; when either exceeds its capacity
 * nothing is copied out and the caller retries with larger buffers. */
enum { VP_RETR_EXTERNAL = 0, VP_RETR_LIST = 1 };
enum { VP_CHAIN_APPROX_NONE = 1, VP_CHAIN_APPROX_SIMPLE = 2 };
int vp_find_contours_u8(vp_ctx* ctx, const uint8_t* src_host, size_t src_stride, int w, int h, int mode, int method,
                        int32_t* points_host, int64_t max_points, int32_t* counts_host, uint8_t* is_hole_host, int max_contours,
                        int32_t* n_contours, int64_t* n_points);

/* utils/feature.py:240-265 `contour_centroid` / `contour_area` (cv2.moments / cv2.contourArea of an integer contour): the Green sums
 * out3 = {sum d, sum d (x' + x), sum d (y' + y)}, d = x' y - x y' over consecutive points (x', y') -> (x, y), as exact integers;
 * host code, no context.  The Python mirror applies cv2's factors 1/2 and 1/6 in float64. */
int vp_polygon_sums_i32(const int32_t* pts_xy, int npts, int64_t* out3);
/* Convex hull of integer points by the monotone chain (host code, exact): distinct hull vertices counter-clockwise from the
 * lexicographically smallest point, collinear points dropped; `out` holds up to npts points.  Used by the cv2.minAreaRect stand-in
 * (modules/bins.py:62). */
int vp_convex_hull_i32(const int32_t* pts, int npts, int32_t* out, int* nout);
/* cv2.minAreaRect stand-in for integer points (host code): rotating calipers over that hull; out5 = centre x, y, width, height,
 * angle in degrees in (0, 90] (OpenCV >= 4.5.1 convention), as floats.  modules/bins.py:62 calls it for every contour. */
int vp_min_area_rect_i32(const int32_t* pts, int npts, float* out5);

/* utils/draw.py:283-327 `draw_contours` / `draw_polylines` (modules/red_buoy.py:39): in-place polyline on a HOST image (no device
 * work, no context): Bresenham steps with a square brush of `thickness` pixels - the Python mirror's rasteriser in C.  pts = npts
 * (x, y) int32 pairs; color has cn entries.  Debug overlay only: agreement with cv2's line drawing is not claimed. */
int vp_draw_polyline_u8(uint8_t* img_host, size_t stride, int w, int h, int cn, const int32_t* pts, int npts, int closed,
                        const uint8_t* color, int thickness);
/* several polylines in one call: counts[k] points each, back to back in pts (cv2.drawContours(img, contours, -1, ...)) */
int vp_draw_polylines_u8(uint8_t* img_host, size_t stride, int w, int h, int cn, const int32_t* pts, const int32_t* counts, int npolys,
                         int closed, const uint8_t* color, int thickness);

/* ---- device-resident forms of the per-operator entry points ----------------------------- *
 * Same arithmetic and argument meaning as vp_cvt_color_u8 / vp_inrange_u8 / vp_morph_u8 / vp_find_contours_u8; images are
 * device pointers.  Nothing is copied; the first three enqueue on the context's stream and return without synchronising, so a
 * module's process() (modules/red_buoy.py:21-38: bgr_to_lab -> range_threshold -> morph_remove_noise -> morph_close_holes ->
 * outer_contours) uploads its frame once and downloads only what Python reads (the Python mirror hands these images around as
 * lazily materialised arrays, vision/devmat.py).  vp_morph_u8_dev: dst must not overlap src; binary_hint 1 = the image is known
 * to hold only 0 / 255 (a mask this library produced), 0 = unknown (one flag is then read back, synchronising).
 * vp_find_contours_dev returns the lists in host memory and synchronises, like its host form. */
int vp_cvt_color_dev(vp_ctx* ctx, int code, const uint8_t* src_dev, size_t src_stride, int w, int h, uint8_t* dst_interleaved_dev,
                     uint8_t* const* dst_planes_dev);
int vp_inrange_u8_dev(vp_ctx* ctx, const uint8_t* src_dev, size_t src_stride, int w, int h, int cn, const int32_t* lo, const int32_t* hi,
                      uint8_t* dst_dev);
/* vp_inrange_u8_dev that can also leave the mask's BIT-PACKED form ((h, ceil(w / 64)) u64, bit i of word j = pixel 64 j + i) in bits_dev
 * (nullable): written when rows are packed, pointers 16-B aligned and w % 64 == 0 - *made_bits says whether (otherwise bits_dev is
 * untouched).  vp_find_contours_bits_dev takes that plane in place of the image and saves the packing launch: the pair serves
 * range_threshold -> outer_contours (modules/red_buoy.py:22, :38) when the mask has not been written to in between. */
int vp_inrange_u8_bits_dev(vp_ctx* ctx, const uint8_t* src_dev, size_t src_stride, int w, int h, int cn, const int32_t* lo, const int32_t* hi,
                           uint8_t* dst_dev, unsigned long long* bits_dev, int* made_bits);
int vp_morph_u8_dev(vp_ctx* ctx, int op, const uint8_t* src_dev, int w, int h, int cn, const uint8_t* kernel, int kw, int kh,
                    int anchor_x, int anchor_y, int iterations, int binary_hint, uint8_t* dst_dev);
/* vp_draw_polylines_u8 into a packed device image (points and counts are host arrays): the same pixels, written by the device, so
 * that an overlay which is only posted (modules/bins.py:20-79) never has to visit the host. */
int vp_draw_polylines_dev(vp_ctx* ctx, uint8_t* img_dev, int w, int h, int cn, const int32_t* pts, const int32_t* counts, int npolys,
                          int closed, const uint8_t* color, int thickness);
/* cv2.addWeighted(a, alpha, b, beta, gamma) on two device images of n bytes (modules/bins.py:20, the mask overlay):
 * saturate(round-half-even(a*alpha + b*beta + gamma)) in correctly rounded doubles; dst may be one of the sources. */
int vp_add_weighted_u8_dev(vp_ctx* ctx, const uint8_t* a_dev, double alpha, const uint8_t* b_dev, double beta, double gamma, size_t n,
                           uint8_t* dst_dev);
int vp_find_contours_bits_dev(vp_ctx* ctx, const unsigned long long* bits_dev, int w, int h, int mode, int method, int32_t* points_host,
                              int64_t max_points, int32_t* counts_host, uint8_t* is_hole_host, int max_contours, int32_t* n_contours,
                              int64_t* n_points);
/* Diagnostic: the largest number of border segments ("heads") a frame of this context's last contour pass held, as far as it has been
 * reported back (single-image calls: exact; batched passes: read from pinned memory without synchronising, so possibly one call
 * late).  It is what the next pass chooses the form of its bookkeeping by (one block per frame / launches over the chip) - a choice
 * the results do not depend on. */
unsigned int vp_contours_last_heads(vp_ctx* ctx);
int vp_find_contours_dev(vp_ctx* ctx, const uint8_t* src_dev, size_t src_stride, int w, int h, int mode, int method,
                         int32_t* points_host, int64_t max_points, int32_t* counts_host, uint8_t* is_hole_host, int max_contours,
                         int32_t* n_contours, int64_t* n_points);

/* ---- fused, batched, device-resident chain -------------------------------------------- */

#define VP_CHAIN_MAX_MORPH 8
typedef struct vp_chain_desc {
    int32_t width, height;
    int32_t color_mode;            /* VP_BGR2LAB, VP_BGR2HSV or VP_BGR2GRAY */
    int32_t lo[3], hi[3];          /* inclusive per-channel bounds on the converted image */
    int32_t n_morph;               /* 0..VP_CHAIN_MAX_MORPH ops applied in order */
    int32_t morph_op[VP_CHAIN_MAX_MORPH];   /* VP_MORPH_ERODE/DILATE/OPEN/CLOSE */
    int32_t morph_kw[VP_CHAIN_MAX_MORPH];   /* all-ones (rect) kernels, anchor = centre */
    int32_t morph_kh[VP_CHAIN_MAX_MORPH];
    int32_t morph_iter[VP_CHAIN_MAX_MORPH];
    int32_t ccl;                   /* 0 = none, 1 = label the cleaned mask, 2 = the threshold mask */
    int32_t numbering;             /* VP_CCL_BLOCK2X2 or VP_CCL_PIXEL */
    int32_t max_labels;            /* stats/centroid rows per frame, incl. background row 0 */
} vp_chain_desc;

typedef struct vp_chain_buffers {  /* device pointers; any output may be NULL */
    const uint8_t* bgr;            /* (n,h,w,3) */
    uint8_t* threshed;             /* (n,h,w) result of inRange — modules/red_buoy.py:23-28 */
    uint8_t* cleaned;              /* (n,h,w) after the morphology ops — red_buoy.py:31-34 */
    int32_t* labels;               /* (n,h,w) */
    int32_t* stats;                /* (n,max_labels,5) */
    double* centroids;             /* (n,max_labels,2) */
    int32_t* nlabels;              /* (n) */
} vp_chain_buffers;

/* Enqueues the whole chain for n frames (n <= 65535, width*height <= 2^30) on the context's stream and returns without
 * synchronising.  This is the hot path bench.py times (one call = one "step"). */
int vp_chain_run(vp_ctx* ctx, const vp_chain_desc* desc, const vp_chain_buffers* dev, int n_frames);
/* Same with host buffers: H2D, chain, D2H, synchronised on return. */
int vp_chain_run_host(vp_ctx* ctx, const vp_chain_desc* desc, const vp_chain_buffers* host, int n_frames);
/* Algorithmic HBM bytes one vp_chain_run call moves (3 B/px read + 1 B/px per mask
 * written + 4 B/px labels), for roofline accounting. */
uint64_t vp_chain_algorithmic_bytes(const vp_chain_desc* desc, const vp_chain_buffers* bufs, int n_frames);

/* ---- chain + contours for the whole batch ------------------------------------------------
 * What modules/red_buoy.py:36-38 (`outer_contours(cleaned)`) and modules/bins.py:27 (`outer_contours`
 * after the morphology) do per frame, for n frames in one launch sequence: the chain above, then
 * cv2.findContours semantics (utils/feature.py:5-40) on the cleaned or the threshold mask. */
typedef struct vp_contour_desc {
    int32_t source;        /* 1 = the cleaned mask, 2 = the threshold mask */
    int32_t mode;          /* VP_RETR_EXTERNAL or VP_RETR_LIST */
    int32_t method;        /* VP_CHAIN_APPROX_NONE or VP_CHAIN_APPROX_SIMPLE */
    int32_t max_contours;  /* capacity per frame */
    int64_t max_points;    /* capacity per frame */
} vp_contour_desc;

typedef struct vp_contour_buffers {  /* device pointers (vp_chain_run_contours) / host pointers (.._host) */
    int32_t* info;         /* (n,2): contours found in the frame; points of the contours that fit */
    int32_t* counts;       /* (n,max_contours) points per contour, in discovery order = raster order of
                              the start pixel; cv2 returns the reverse order */
    int32_t* offsets;      /* (n,max_contours) first point of the contour within the frame's point list */
    uint8_t* is_hole;      /* (n,max_contours) */
    int32_t* points;       /* (n,max_points,2) (x,y) */
    double* features;      /* optional (may be NULL): (n,max_contours,8) per contour {m00, m10, m01 as cv2.moments gives them
                              for the contour (utils/feature.py:240-252), area = cv2.contourArea (:255-265), bounding box x, y,
                              width, height as cv2.boundingRect} - computed on the device, exact (the sums are integers) */
} vp_contour_buffers;

/* Contours beyond max_contours are counted in info[.][0] but not traced; a contour whose points
 * would pass max_points is skipped (info[.][1] still counts them), so a caller can repeat with
 * larger capacities.  Enqueues and returns without synchronising. */
int vp_chain_run_contours(vp_ctx* ctx, const vp_chain_desc* desc, const vp_chain_buffers* dev, const vp_contour_desc* cdesc,
                          const vp_contour_buffers* cdev, int n_frames);
/* Host buffers: H2D, chain, contours, D2H, synchronised on return. */
int vp_chain_run_contours_host(vp_ctx* ctx, const vp_chain_desc* desc, const vp_chain_buffers* host, const vp_contour_desc* cdesc,
                               const vp_contour_buffers* chost, int n_frames);

/* ---- detector pre / post-processing (BASELINE config 5) ----------------------
 * modules/yolo.py:112 `self.model.track(image)` hides these steps inside ultralytics (not in the reference tree, not
 * installed): LetterBox (scale to fit, centre, pad, BGR->RGB, HWC->CHW, /255) before the network, non-maximum suppression
 * after it.  Checked against plain PyTorch fp32 restatements; parity with the third-party package is unpinned.
 *
 * vp_letterbox_*: src (h,w,3) BGR uint8 -> dst (3,dst_h,dst_w) float32 RGB in [0,1]; the resize is cv2.resize INTER_LINEAR's
 * 8-bit arithmetic; geom_out (3 floats, may be NULL) = {scale r, left pad, top pad} for mapping boxes back.
 * vp_nms_*: boxes (n,4) x1,y1,x2,y2 (rotated: (n,5) x,y,w,h,angle[rad]), scores (n); keep_out receives up to max_keep original
 * indices in descending score order (ties: lower index first).  rotated = 0: greedy, suppress IoU > thr.  rotated = 1:
 * probabilistic IoU of the boxes' Gaussian models; a box is dropped when any higher-scored box overlaps it by >= thr.
 * n <= 16384.  The _dev forms take device pointers (e.g. tensors of a PyTorch-ROCm model via data_ptr()), enqueue on the
 * context's stream and return without synchronising. */
/* cv2.threshold(src, thresh, maxval, type)[1] on 8-bit data of any channel count (utils/color.py:124-199: binary_threshold,
 * binary_threshold_inv, max_threshold, above_threshold, below_threshold): the comparison is against floor(thresh), maxval is
 * rounded and saturated.  n_bytes = h * w * channels of a tightly packed image; dst may equal src. */
enum { VP_THRESH_BINARY = 0, VP_THRESH_BINARY_INV = 1, VP_THRESH_TRUNC = 2, VP_THRESH_TOZERO = 3, VP_THRESH_TOZERO_INV = 4 };
int vp_threshold_u8(vp_ctx* ctx, const uint8_t* src_host, size_t n_bytes, double thresh, double maxval, int type, uint8_t* dst_host);
/* cv2.threshold(src, 0, maxval, type | THRESH_OTSU) on a single-channel 8-bit image (utils/color.py:204-217 otsu_threshold):
 * threshold chosen by getThreshVal_Otsu_8u from the histogram (device) with the reference's double-precision scan (host),
 * returned in *thresh_out, then applied as above. */
int vp_otsu_threshold_u8(vp_ctx* ctx, const uint8_t* src_host, size_t n_bytes, double maxval, int type, double* thresh_out, uint8_t* dst_host);
/* cv2.GaussianBlur(src, (kw, kh), sigma1, sigma2) on 8-bit images, cn = 1..4 (modules/preprocessor.py:110-114,
 * utils/transform.py simple_gaussian_blur): OpenCV's bit-exact fixed-point path (8.8 taps summing to 256, 16.16 vertical sums
 * rounded half up, BORDER_REFLECT_101).  kw, kh odd, 1..511; sigma <= 0 means "from the kernel size", sigma2 <= 0 means sigma1.
 * dst may equal src. */
int vp_gaussian_blur_u8(vp_ctx* ctx, const uint8_t* src_host, int w, int h, int cn, int kw, int kh, double sigma1, double sigma2,
                        uint8_t* dst_host);
/* cv2.resize(src, (dst_w, dst_h)) with the default INTER_LINEAR on 8-bit images, cn = 1..4 interleaved channels
 * (modules/preprocessor.py:136-144): OpenCV's generic fixed-point path, including the 2x2 box average it substitutes at an
 * exact halving.  (IPP-enabled OpenCV builds may round differently.) */
int vp_resize_u8(vp_ctx* ctx, const uint8_t* src_host, int w, int h, int cn, int dst_w, int dst_h, uint8_t* dst_host);
/* cv2.adaptiveThreshold(src, max_value, ADAPTIVE_THRESH_MEAN_C, type, block_size, c) on a single-channel 8-bit image
 * (utils/color.py:220-254 adaptive_threshold_mean / adaptive_threshold_mean_inv).  type: VP_THRESH_BINARY or VP_THRESH_BINARY_INV;
 * block_size odd, 3..151 (the range in which OpenCV's three roundings of the box mean provably coincide).  dst may equal src. */
int vp_adaptive_threshold_mean_u8(vp_ctx* ctx, const uint8_t* src_host, int w, int h, double max_value, int type, int block_size, double c,
                                  uint8_t* dst_host);
/* cv2.Canny(image, threshold1, threshold2) with the default 3x3 aperture and L1 gradient on 8-bit images, cn = 1..4
 * (utils/feature.py:43-101 canny / simple_canny): Sobel derivatives, non-maximum suppression with OpenCV's integer direction test,
 * hysteresis as connected components of the surviving pixels that hold a pixel above the high threshold.  dst: (h, w) 0 / 255. */
int vp_canny_u8(vp_ctx* ctx, const uint8_t* src_host, int w, int h, int cn, double threshold1, double threshold2, uint8_t* dst_host);
/* cv2.warpAffine(src, M, (dst_w, dst_h), flags, borderMode, borderValue) with bilinear interpolation on 8-bit images, cn = 1..4
 * (modules/preprocessor.py:130-135 rotate with BORDER_REPLICATE, :145-149 translate; utils/transform.py:180-210).  m23: the 2x3
 * matrix, row-major doubles, mapping source to destination unless VP_WARP_INVERSE_MAP is set.  OpenCV's classical fixed-point path
 * (every release up to 4.10): coordinates in 22.10 fixed point with 5 fractional bits kept, 15-bit weights, round half up.
 * border_value: cn bytes or NULL (0).  src and dst must not overlap. */
enum { VP_BORDER_CONSTANT = 0, VP_BORDER_REPLICATE = 1 };
enum { VP_WARP_INVERSE_MAP = 16 };
int vp_warp_affine_u8(vp_ctx* ctx, const uint8_t* src_host, int w, int h, int cn, const double* m23, int flags, int border_mode,
                      const uint8_t* border_value, uint8_t* dst_host, int dst_w, int dst_h);
int vp_letterbox_u8_f32(vp_ctx* ctx, const uint8_t* src_host, int w, int h, int dst_w, int dst_h, int pad_value, float* dst_host,
                        float* geom_out);
int vp_letterbox_dev(vp_ctx* ctx, const uint8_t* src_dev, int w, int h, int dst_w, int dst_h, int pad_value, float* dst_dev, float* geom_out);
int vp_nms_f32(vp_ctx* ctx, const float* boxes_host, const float* scores_host, int n, float thr, int rotated, int max_keep,
               int32_t* keep_out_host, int32_t* n_keep_out);
int vp_nms_dev(vp_ctx* ctx, const float* boxes_dev, const float* scores_dev, int n, float thr, int rotated, int max_keep,
               int32_t* keep_out_dev, int32_t* n_keep_dev);

/* ---- device memory helpers (so a host program needs no HIP binding of its own) --------- */
int vp_dev_alloc(vp_ctx* ctx, size_t bytes, void** dev_ptr);
int vp_dev_free(vp_ctx* ctx, void* dev_ptr);    /* ctx may be NULL (memory that outlived its context) */
/* Page-locked host memory: transfers to/from it run at PCIe speed (hipHostMalloc / hipHostFree). */
int vp_host_alloc(vp_ctx* ctx, size_t bytes, void** host_ptr);
int vp_host_free(vp_ctx* ctx, void* host_ptr);   /* ctx may be NULL */
/* Page-locks memory the caller already has (hipHostRegister), so that vp_memcpy_h2d_async from it is a DMA at PCIe speed with no
 * staging copy: the runtime registers the mapping of a camera_message_framework block once (cmf_block_mapping,
 * include/camera_message_framework_c.h) and then uploads frames straight out of the ring slots - the replacement for the two host
 * copies of lib/camera_message_framework.cpp:421-452 + core/base.py:765-768.  VP_ERR_UNSUPPORTED when the runtime refuses the range
 * (the caller keeps its copying path); unregister before the memory is unmapped. */
int vp_host_register(vp_ctx* ctx, void* host_ptr, size_t bytes);
int vp_host_unregister(vp_ctx* ctx, void* host_ptr);   /* ctx may be NULL */
/* ---- frame feeder: the newest frame of a camera_message_framework block kept in HBM by a thread of the library's own ----------
 * Replaces, for a module on the runtime, the per-iteration sequence read_frame (seqlock memcpy, lib/camera_message_framework.cpp:
 * 379-455) -> np.array(copy) (core/base.py:765-768) -> upload: the feeder waits on the block's condition variable, copies every new
 * frame out of its ring slot into one of four device buffers on its own stream (the block's mapping must be page-locked:
 * vp_host_register), checks the slot's sequence number after the copy and publishes the buffer; vp_feeder_take hands the newest one
 * over without waiting for anything.  The block library is not linked: its five entry points (cmf_wait_for_frame, cmf_peek_frame,
 * cmf_peek_validate, create_frame, delete_frame of include/camera_message_framework_c.h) are passed as addresses.
 * take -> 0: meta_out (sizeof(Frame) = 360 bytes) + *dev_out; 1: nothing newer than the last frame taken; 2: block deleted.
 * release: the buffer may be overwritten once the work queued so far on `consumer`'s stream has passed (consumer may be NULL).
 * stop ends the thread (buffers handed out stay valid); destroy frees everything. */
typedef struct vp_feeder vp_feeder;
vp_feeder* vp_feeder_start(int device, void* block, size_t entry_bytes, void* fn_wait_for_frame, void* fn_peek_frame, void* fn_peek_validate,
                           void* fn_create_frame, void* fn_delete_frame);
int vp_feeder_take(vp_feeder* f, void* meta_out, void** dev_out);
int vp_feeder_release(vp_feeder* f, vp_ctx* consumer, void* dev);
int vp_feeder_counts(vp_feeder* f, unsigned long long* fetched, unsigned long long* dropped_as_lapped);
int vp_feeder_stop(vp_feeder* f);
int vp_feeder_destroy(vp_feeder* f);

/* ---- posts by DMA: a device image into the ring slot of a camera_message_framework block, no host pass over the pixels ---------
 * Replaces, for ModuleBase.post() of an image that lives in HBM, the download into a fresh array (core/base.py:846-876 `np.array(copy)`)
 * plus write_frame's memcpy into the slot (lib/camera_message_framework.cpp:306-374) at the flush (core/base.py:832-839).  The caller
 * opens the slot (cmf_write_begin, include/camera_message_framework_c.h), then:
 *   vp_post_d2h(ctx, lane, slot_bytes, image_dev, bytes, &done): the image AS IT IS NOW (everything queued so far on the context's
 *     stream) is copied into `slot_bytes` (inside a mapping page-locked with vp_host_register) on post stream `lane` (taken modulo the
 *     four the context has: copies of one lane run in order - a block keeps its lane - different lanes side by side); *done is an
 *     opaque handle on the end of that copy.  Returns at once.
 *   vp_post_done(ctx, done) -> 1 the bytes are in the slot (commit the write), 0 not yet, negative on a device error.
 *   vp_post_wait(ctx, done): blocks the calling thread until they are.
 *   vp_post_fence(ctx, done): work queued on the context's stream from now on starts after the copy - call before anything OVERWRITES
 *     the image (readers may run beside the copy).
 *   vp_post_free(ctx, done): hands the handle back (ctx may be NULL: the context is gone). */
int vp_post_d2h(vp_ctx* ctx, int lane, void* slot_bytes_host, const void* image_dev, size_t bytes, void** done);
int vp_post_done(vp_ctx* ctx, void* done);
int vp_post_wait(vp_ctx* ctx, void* done);
int vp_post_fence(vp_ctx* ctx, void* done);
int vp_post_free(vp_ctx* ctx, void* done);

/* device -> device on the context's stream (ordered with everything else the context runs); returns at once */
int vp_memcpy_d2d_async(vp_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);
int vp_memcpy_h2d(vp_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes); /* synchronous */
/* The same copy enqueued on the context's stream: src_host must stay unchanged until vp_wait_uploads (or vp_synchronize) returns.
 * vp_wait_uploads waits for the copies only, not for kernels enqueued behind them: an operator enqueues the copy of its input,
 * does its host-side work and launches, and waits just before handing control back to code that may change the input. */
int vp_memcpy_h2d_async(vp_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int vp_wait_uploads(vp_ctx* ctx);
int vp_memcpy_d2h(vp_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes); /* synchronous */

#ifdef __cplusplus
}
#endif
#endif /* VP_H */
