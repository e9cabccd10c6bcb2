// Internal declarations shared by the libvp translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/vp.h"

typedef unsigned long long u64;
typedef unsigned int u32;

struct vp_tables {          // device copies of the OpenCV integer tables
    const uint16_t* gamma;  // [256]  sRGBGammaTab_b
    const uint16_t* cbrt;   // [2048] LabCbrtTab_b (indices 0..2040 are reachable)
    const int32_t* sdiv;    // [256]
    const int32_t* hdiv;    // [256]  hdiv_table180
};

enum { VPK_COLOR = 0, VPK_MORPH, VPK_CCL_LOCAL, VPK_CCL_BOUNDARY, VPK_CCL_FLATTEN, VPK_CCL_RANK, VPK_CCL_BG, VPK_CCL_STATS,
       VPK_CCL_FINAL, VPK_CCL_WRITE, VPK_MEMSET, VPK_OTHER, VPK_CCL2_LOCAL, VPK_CCL2_MERGE, VPK_CCL2_WRITE, VPK_COUNT };

struct vp_prof {
    bool on;
    int cap, used;          // records
    hipEvent_t* ev;         // 2 per record
    int* ids;
};

struct vp_ctx {
    int device;
    hipStream_t stream;
    hipStream_t own_stream;
    hipEvent_t ev0, ev1;
    void* d_tables;
    vp_tables tab;
    uint8_t* ws;      // grow-only device workspace, carved per call
    size_t ws_cap;
    size_t ws_off;
    uint8_t* hstage;  // grow-only pinned host staging for small results (contour lists)
    size_t hstage_cap;
    // small host -> device hand-overs that must not wait (overlay vertices): a ring of pinned chunks, each free again once the work
    // queued behind its copy has run
    uint8_t* ring_buf[4];
    size_t ring_cap[4];
    hipEvent_t ring_ev[4];
    int ring_busy[4];
    int ring_next;
    int ct_lds_set;               // k_ct_jump's LDS attribute has been set on this context's device
    uint32_t* ct_hint_host;       // pinned: head counts of the last batched contour pass (vp_contours.inl vp_ct_hint_slots)
    int ct_hint_n;
    uint32_t ct_heads_hint;   // border segments the last single-image contour pass counted (vp_find_contours_*: which form of the bookkeeping to launch)
    int num_cu;
    int chain_streams;            // sub-batches of a chain run on this many internal streams (>= 1)
    hipStream_t aux[4];
    hipEvent_t ev_fork, ev_join[4];
    hipStream_t fb_stream;        // side stream of the labelling: the one-level kernels for crowded frames run here, beside the label write
    hipEvent_t ev_fb_fork, ev_fb_join;
    hipEvent_t ev_upload;         // recorded after an enqueued host-to-device copy (vp_memcpy_h2d_async / vp_wait_uploads)
    hipStream_t post_stream[4];   // posts by DMA (vp_post.hip): device image -> ring slot copies run on these lanes, beside the context's stream; made on first use
    hipEvent_t post_fork;         // "the image as it is now" on the context's stream
    hipEvent_t post_free[32];     // end-of-copy events handed back by vp_post_free
    int post_nfree;
    int post_lock;                // spin lock of the three fields above (a context is zero-filled at creation: no constructors in here)
    const u32* cb_folds_dev;      // colour balance: device counter of tiles whose running mean had to be folded (last call); null or cb_folds_own
    u32* cb_folds_own;            // context-owned device word the counter is copied to (the workspace it is made in is carved anew per call)
    int ccl_levels;               // 2: two-level labelling with the one-level kernels as fallback (default); 1: one-level only
    size_t c3_lds_set[6];         // dynamic LDS the crowded-frame kernels have been allowed on THIS device (link, label; short and tall strips): the attribute is per device
    void* c3_acc;                 // crowded-frame labelling: accumulators of components that span strips, all empty between calls (vp_ccl.hip)
    size_t c3_acc_bytes;
    int c3_acc_dirty;             // a call was cut short after its labelling launch: reinitialise before the next use
    int flat_ops;                 // 1 (default): the per-operator kernels take their 16-px-per-lane forms when rows are packed and pointers aligned; 0: always the generic kernels (tests)
    int ccl_mcap;                 // components per frame the merge block accepts (-1: its LDS capacity); tests lower it to force the fallback
    vp_prof prof;
    char err[256];
};

// brackets one kernel launch with events when profiling is on
struct vp_prof_scope {
    vp_ctx* c;
    int rec;
    vp_prof_scope(vp_ctx* ctx, int id) : c(ctx), rec(-1)
    {
        if (c->prof.on && c->prof.used < c->prof.cap) {
            rec = c->prof.used++;
            c->prof.ids[rec] = id;
            (void)hipEventRecord(c->prof.ev[2 * rec], c->stream);
        }
    }
    ~vp_prof_scope()
    {
        if (rec >= 0) (void)hipEventRecord(c->prof.ev[2 * rec + 1], c->stream);
    }
};

void vp_post_teardown(vp_ctx* ctx);                        // vp_post.hip: joins and frees the post stream (vp_destroy)

// ---- workspace ---------------------------------------------------------------------------
int vp_ws_reserve(vp_ctx* ctx, size_t bytes);               // may reallocate (synchronises)
void* vp_ws_take(vp_ctx* ctx, size_t bytes);                // 256-B aligned carve; NULL if exhausted
void* vp_hstage(vp_ctx* ctx, size_t bytes);                 // pinned host staging of at least `bytes` (NULL: out of memory)
static inline size_t vp_align(size_t n, size_t a = 256) { return (n + a - 1) / a * a; }
int vp_fail(vp_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess);
#define VP_HIP(ctx, call)                                                   \
    do {                                                                    \
        hipError_t e__ = (call);                                            \
        if (e__ != hipSuccess) return vp_fail((ctx), VP_ERR_HIP, #call, e__); \
    } while (0)

// ---- host-side table generation (vp_tables.cpp) ------------------------------------------
void vp_host_tables(uint16_t* gamma, uint16_t* cbrt_tab, int32_t* sdiv, int32_t* hdiv180, int32_t* labC);

// ---- colour kernels (vp_color.hip) ---------------------------------------------------------
struct vp_range3 { int lo[3], hi[3]; int lo2, hi2; };   // lo2 / hi2: second interval of the hue test inside the HSV threshold kernels (vpk_color_thresh)
// fused convert + inRange (+ optional u8 mask, + optional bit-packed mask) over n frames
int vpk_color_thresh(vp_ctx* ctx, int mode, const uint8_t* d_bgr, size_t stride, int w, int h, int n,
                     const vp_range3& r, uint8_t* d_mask /*nullable*/, u64* d_bits /*nullable*/);
int vpk_cvt_color(vp_ctx* ctx, int code, const uint8_t* d_src, size_t stride, int w, int h, uint8_t* d_dst,
                  uint8_t* d_p0, uint8_t* d_p1, uint8_t* d_p2);
int vpk_inrange_u8(vp_ctx* ctx, const uint8_t* d_src, size_t stride, int w, int h, int cn, const vp_range3& r,
                   uint8_t* d_dst, u64* d_bits = nullptr, int* made_bits = nullptr);
int vpk_inrange_f32(vp_ctx* ctx, const float* d_src, size_t stride_bytes, int w, int h, float lo, float hi,
                    uint8_t* d_dst);
int vpk_kth_f32(vp_ctx* ctx, const float* d_src, size_t n, size_t k, u32* d_hist, float* out);
int vpk_bgr2lab_f32(vp_ctx* ctx, const float* d_src, size_t npx, float* d_dst);
int vpk_color_distance(vp_ctx* ctx, const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, size_t npx,
                       const float* color, const float* wts, int skipmask, float* d2, uint8_t* sq);

// ---- colour balance (vp_balance.hip) ----------------------------------------------------------
int vpk_hsv2bgr(vp_ctx* ctx, const uint8_t* d_src, size_t npx, uint8_t* d_dst);
size_t vp_balance_ws_bytes(int n, int tiles);
int vpk_color_balance(vp_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int w, int h, int n, int flags, int hblocks, int vblocks);

// ---- detector pre / post-processing (vp_yolo.hip) ------------------------------------------------
int vpk_letterbox(vp_ctx* ctx, const uint8_t* d_src, int sw, int sh, int dw, int dh, int pad, float* d_dst, float* geom_out);
int vpk_resize_u8(vp_ctx* ctx, const uint8_t* d_src, int sw, int sh, int cn, int dw, int dh, uint8_t* d_dst);
int vpk_warp_affine_u8(vp_ctx* ctx, const uint8_t* d_src, int sw, int sh, int cn, const double* M23, int inverse_map, int border,
                       const uint8_t* cval, uint8_t* d_dst, int dw, int dh);
size_t vp_nms_ws_bytes(int n);
int vpk_nms(vp_ctx* ctx, const float* d_boxes, const float* d_scores, int n, float thr, int rotated, int max_keep, int* d_keep, int* d_nkeep);

// ---- filters (vp_filter.hip) -----------------------------------------------------------------------
int vpk_threshold_u8(vp_ctx* ctx, const uint8_t* d_src, size_t n, int ithresh, int imaxval, int type, uint8_t* d_dst);
int vpk_hist_u8(vp_ctx* ctx, const uint8_t* d_src, size_t n, u32* d_hist);   // d_hist: 256 counters
int vpk_adaptive_threshold_mean(vp_ctx* ctx, const uint8_t* d_src, int w, int h, int imax, int idelta, int inv, int block, uint16_t* d_tmp, uint8_t* d_dst);
size_t vp_canny_ws_bytes(int w, int h);
int vpk_canny_u8(vp_ctx* ctx, const uint8_t* d_src, int w, int h, int cn, int low, int high, uint8_t* d_dst);
void vp_gaussian_taps(int n, double sigma, uint16_t* out);   // n odd, <= 511
int vpk_gaussian_blur(vp_ctx* ctx, const uint8_t* d_src, int w, int h, int cn, const uint16_t* d_taps, int kw, int kh, uint16_t* d_tmp, uint8_t* d_dst);

// ---- morphology (vp_morph.hip) ---------------------------------------------------------------
struct vp_bitstage { int dilate; int l, r, u, d; };  // window [-l, r] x [-u, d]
#define VP_MAX_STAGES 32
struct vp_bitplan { int n; vp_bitstage s[VP_MAX_STAGES]; };
// words per row of a bit image
static inline int vp_ww(int w) { return (w + 63) / 64; }
// u8 (non-zero = 1) -> bits; d_flags[0] |= 1 if any byte is neither 0 nor 255
int vpk_pack_bits(vp_ctx* ctx, const uint8_t* d_src, size_t stride, int w, int h, int n, u64* d_bits,
                  int* d_flags /*nullable*/);
int vpk_unpack_bits(vp_ctx* ctx, const u64* d_bits, int w, int h, int n, uint8_t* d_dst);
// runs the stage plan on bit images; outputs (each nullable): bits, u8 mask
int vpk_morph_bits(vp_ctx* ctx, const vp_bitplan& plan, const u64* d_in, int w, int h, int n, u64* d_out_bits,
                   uint8_t* d_out_mask);
// generic u8 morphology, arbitrary structuring element, cn channels, one pass
int vpk_morph_generic(vp_ctx* ctx, int dilate, const uint8_t* d_src, int w, int h, int cn, const int16_t* d_offs,
                      int noffs, uint8_t* d_dst);
// span form (see vp_morph.hip): d_spans = nspans triples (dy, x0, x1); d_tab = 7 planes of w*h*cn bytes; max_len = longest span
int vpk_morph_spans(vp_ctx* ctx, int dilate, const uint8_t* d_src, int w, int h, int cn, const int16_t* d_spans, int nspans, int max_len,
                    uint8_t* d_tab, uint8_t* d_dst);
int vpk_draw_small(vp_ctx* ctx, uint8_t* d_img, int w, int h, int cn, const int32_t* pts, const int32_t* nxt, int npts, int thickness, const uint8_t* color);
int vpk_draw_segments(vp_ctx* ctx, uint8_t* d_img, int w, int h, int cn, const int32_t* d_pts, const int32_t* d_nxt, int npts, int thickness,
                      const uint8_t* color);
int vpk_add_weighted_u8(vp_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, double alpha, double beta, double gamma, uint8_t* dst);
int vpk_absdiff_sub_u8(vp_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* dst);  // a - b saturating

// ---- CCL (vp_ccl.hip) ----------------------------------------------------------------------------
struct vp_ccl_ws {           // per-batch scratch, all device pointers
    u32* parent;             // [n][nids]
    u32* seglabel;           // [n][nids]
    u32* flags;              // [n][nids/32]   root bitmap
    u32* prefix;             // [n][nids/32]   exclusive popcount prefix
    void* acc;               // [n][max_labels] accumulators
    u32* wordlabel;          // [n][h*ww]      label of the first segment of each word
    void* bgpart;            // [n][8]         background partial records
    // two-level path (vp_ccl2.inl); wordlabel / seglabel double as its per-word / per-segment component indices
    u32* c2_ncomp;           // [n][strips]            components per strip (0xffffffff: strip not resolved in LDS)
    void* c2_recs;           // [n][strips][C2_RC]     component records (statistics + numbering key)
    void* c2_bgbox;          // [n][strips]            bounding box of the strip's zero pixels
    u32* c2_label;           // [n][strips][C2_RC]     (strip, component) -> label
    u32* c2_crowded;         // [n]                    1: frame handed over to the crowded-frame path
    // crowded-frame path (vp_ccl3.inl); parent / flags / prefix / acc are shared with the one-level kernels, seglabel holds its u16 roots
    u32* c3_child;           // [n][nids/32]           roots that absorbed a root of another strip
    u32* c3_lroot;           // [n][nids/32]           the strip-local roots (the root bitmap before the boundary unions)
    u32* c3_clist;           // [n]                    the frames handed over, in no particular order
    u32* c3_ncrowded;        // [1]                    their number
    void* c3_state;          // [n]                    per-frame counters and totals
    unsigned char* c3_items; // [n * strips]: which labelling launch takes the strip (vp_ccl3.inl)
    u32* c3_barr;            // [n][strips + 1]        arrivals at every strip boundary
    void* c3_tot;            // [n][strips]            per strip: foreground sums and the box of its zero pixels (for the background row)
};
bool vp_ccl_ws_ok(const vp_ccl_ws& ws);
size_t vp_ccl_nids(int w, int h);   // multiple of 32
size_t vp_ccl_ws_bytes(int w, int h, int n, int max_labels);
void vp_ccl_ws_carve(vp_ctx* ctx, int w, int h, int n, int max_labels, vp_ccl_ws* out);
size_t vp_contours_ws_bytes(int w, int h, int n, int max_contours);
int vpk_contour_features(vp_ctx* ctx, const int32_t* d_info, const int32_t* d_counts, const int32_t* d_offsets, const int32_t* d_points, int n,
                         int max_contours, long long max_points, double* d_features);
// contours of n bit images (no labelling involved: vp_contours.inl).  many_heads: the caller expects a frame with very many border
// segments (its last pass said so through d_nheads_out) - a choice between two forms of the same steps, not of the result.
// host (n == 1 only): pinned, device-visible buffers the producing kernels write the results into as well - info {n_contours,
// n_points, heads}, counts / offsets / is_hole [max_contours], the first points_cap points - so that the caller only synchronises.
// defer_big (n == 1, not many_heads): a frame with more heads than the one block's LDS tables hold reports n_contours = -1 and nothing
// else; the caller repeats the pass with many_heads.
uint32_t vp_ct_batch_hint(vp_ctx* ctx);      // largest head count the last batched pass reported (a guess; no synchronisation)
struct vp_contour_mirror { int32_t* info; int32_t* counts; int32_t* offsets; uint8_t* is_hole; int32_t* points; long long points_cap; };
int vpk_find_contours(vp_ctx* ctx, const u64* d_bits, int w, int h, int n, int mode, int method, int32_t* d_counts, uint8_t* d_is_hole,
                      int32_t* d_offsets, int32_t* d_points, int max_contours, long long max_points, int32_t* d_info, bool many_heads = false,
                      uint32_t* d_nheads_out = nullptr, const vp_contour_mirror* host = nullptr, bool defer_big = false);
int vpk_ccl(vp_ctx* ctx, const u64* d_bits, int w, int h, int n, int numbering, const vp_ccl_ws& ws, int32_t* d_labels,
            int32_t* d_stats, double* d_centroids, int max_labels, int32_t* d_nlabels);

// 16-byte streaming store for write-once outputs (masks, labels).  VP_NT_STORES selects the nontemporal form.
#ifndef VP_NT_STORES
#define VP_NT_STORES 1
#endif
typedef int vp_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void vp_store16(void* dst, u32 a, u32 b, u32 c, u32 d)
{
    vp_v4i v = {(int)a, (int)b, (int)c, (int)d};
#if VP_NT_STORES
    __builtin_nontemporal_store(v, reinterpret_cast<vp_v4i*>(dst));
#else
    *reinterpret_cast<vp_v4i*>(dst) = v;
#endif
}
