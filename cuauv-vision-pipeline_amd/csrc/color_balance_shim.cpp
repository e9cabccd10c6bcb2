// libauv-color-balance.so: the reference's `process_frame` entry (utils/color_correction/color_balance.hpp:9-14) on top of libvp.
#include <cstdlib>
#include <mutex>
#include "../../include/color_balance_c.h"
#include "../../include/vp.h"

namespace {
std::mutex g_mu;
vp_ctx* g_ctx = nullptr;
}

extern "C" {

int process_frame(unsigned char* arr, size_t height, size_t width, size_t depth, bool equalize_rgb, bool rgb_contrast_correct,
                  bool hsv_contrast_correct, bool hsi_contrast_correct, bool rgb_extrema_clipping, bool adaptive_cast_correction,
                  int horizontal_blocks, int vertical_blocks)
{
    if (!arr || depth != 3 || !height || !width || height > 0x7fffffff || width > 0x7fffffff) return VP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_mu);   // a context is thread-compatible, the reference's entry is callable from any thread
    if (!g_ctx) {
        const char* e = getenv("VP_DEVICE");
        g_ctx = vp_create(e ? atoi(e) : 0);
        if (!g_ctx) return VP_ERR_HIP;
    }
    const int flags = (equalize_rgb ? VP_CB_EQUALIZE_RGB : 0) | (rgb_contrast_correct ? VP_CB_RGB_CONTRAST : 0) |
                      (hsv_contrast_correct ? VP_CB_HSV_CONTRAST : 0) | (hsi_contrast_correct ? VP_CB_HSI_CONTRAST : 0) |
                      (rgb_extrema_clipping ? VP_CB_EXTREMA_CLIPPING : 0) | (adaptive_cast_correction ? VP_CB_ADAPTIVE_CAST : 0);
    return vp_color_balance_u8(g_ctx, arr, (int)width, (int)height, flags, horizontal_blocks, vertical_blocks, arr);
}

const char* color_balance_last_error(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return vp_last_error(g_ctx);
}

}  // extern "C"
