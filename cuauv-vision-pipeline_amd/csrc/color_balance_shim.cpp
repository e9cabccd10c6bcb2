// libauv-color-balance.so: the reference's `process_frame` entry (utils/color_correction/color_balance.hpp:9-14) on top of libvp.
//
// The reference binding ignores the return value (modules/color_balance.py:105: the reference's function always returns 0), so a
// failure here must not pass silently: every distinct error is written to stderr once, and a tiling that does not divide the frame
// (the one case the GPU path does not implement: the reference wraps such tiles into the next row and processes pixels twice) is
// balanced with one tile for the whole frame instead of leaving the caller's image unbalanced.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include "../../include/color_balance_c.h"
#include "../../include/vp.h"

namespace {
std::mutex g_mu;
vp_ctx* g_ctx = nullptr;
char g_err[320] = "";
std::set<std::string>* g_seen = nullptr;

void report(int rc, const char* what)   // g_mu held
{
    const char* msg = g_ctx ? vp_last_error(g_ctx) : vp_last_error(nullptr);
    snprintf(g_err, sizeof g_err, "libauv-color-balance: %s: %s (%d)%s%s", what, vp_strerror(rc), rc, (msg && *msg) ? ": " : "", (msg && *msg) ? msg : "");
    if (!g_seen) g_seen = new std::set<std::string>();
    if (g_seen->insert(g_err).second) fprintf(stderr, "%s\n", g_err);
}
}

extern "C" {

int process_frame(unsigned char* arr, size_t height, size_t width, size_t depth, bool equalize_rgb, bool rgb_contrast_correct,
                  bool hsv_contrast_correct, bool hsi_contrast_correct, bool rgb_extrema_clipping, bool adaptive_cast_correction,
                  int horizontal_blocks, int vertical_blocks)
{
    std::lock_guard<std::mutex> lk(g_mu);   // a context is thread-compatible, the reference's entry is callable from any thread
    if (!arr || depth != 3 || !height || !width || height > 0x7fffffff || width > 0x7fffffff) {
        report(VP_ERR_INVALID, "process_frame arguments (the image is left as it was)");
        return VP_ERR_INVALID;
    }
    if (!g_ctx) {
        const char* e = getenv("VP_DEVICE");
        g_ctx = vp_create(e ? atoi(e) : 0);
        if (!g_ctx) { report(VP_ERR_HIP, "no device context: the image is NOT balanced"); return VP_ERR_HIP; }
    }
    const int flags = (equalize_rgb ? VP_CB_EQUALIZE_RGB : 0) | (rgb_contrast_correct ? VP_CB_RGB_CONTRAST : 0) |
                      (hsv_contrast_correct ? VP_CB_HSV_CONTRAST : 0) | (hsi_contrast_correct ? VP_CB_HSI_CONTRAST : 0) |
                      (rgb_extrema_clipping ? VP_CB_EXTREMA_CLIPPING : 0) | (adaptive_cast_correction ? VP_CB_ADAPTIVE_CAST : 0);
    int rc = vp_color_balance_u8(g_ctx, arr, (int)width, (int)height, flags, horizontal_blocks, vertical_blocks, arr);
    if (rc == VP_ERR_UNSUPPORTED && (horizontal_blocks != 1 || vertical_blocks != 1)) {
        report(rc, "tiles do not divide the frame; balancing with one tile for the whole frame instead");
        rc = vp_color_balance_u8(g_ctx, arr, (int)width, (int)height, flags, 1, 1, arr);
    }
    if (rc != VP_OK) report(rc, "the image is NOT balanced");
    return rc;
}

const char* color_balance_last_error(void)
{
    static thread_local char copy[sizeof g_err];
    std::lock_guard<std::mutex> lk(g_mu);
    memcpy(copy, g_err, sizeof copy);        // copied while the lock is held: the text a caller reads cannot change under it
    return copy;
}

}  // extern "C"
