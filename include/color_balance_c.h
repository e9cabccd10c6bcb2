/*
 * color_balance_c.h — C ABI of libauv-color-balance.so as the reference declares it
 * (utils/color_correction/color_balance.hpp:9-14) and loads it (modules/color_balance.py:12
 * `load_library('libauv-color-balance.so')`, called at :104-106).  This repo's library of the same name
 * exports the same symbol with the same argument list; the work runs on the GPU through libvp
 * (vp_color_balance_u8, include/vp.h) with one process-wide context created on first use.
 *
 * arr: (height, width, depth = 3) BGR uint8, modified in place.  Returns 0, or a negative VP_ERR_* code
 * (the reference always returns 0 and its binding, modules/color_balance.py:105, ignores the value: so every distinct
 * failure is also written to stderr once).  horizontal_blocks / vertical_blocks that do not divide the frame (the
 * reference wraps such tiles into the next row and processes pixels twice) are not implemented: the frame is then
 * balanced with a single tile and a warning.  Device selection: environment variable VP_DEVICE (default 0).
 */
#ifndef COLOR_BALANCE_C_H
#define COLOR_BALANCE_C_H
#include <stdbool.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
int process_frame(unsigned char* arr, size_t height, size_t width, size_t depth, bool equalize_rgb, bool rgb_contrast_correct,
                  bool hsv_contrast_correct, bool hsi_contrast_correct, bool rgb_extrema_clipping, bool adaptive_cast_correction,
                  int horizontal_blocks, int vertical_blocks);
/* last error text of the shim's context (extension; not in the reference) */
const char* color_balance_last_error(void);
#ifdef __cplusplus
}
#endif
#endif
