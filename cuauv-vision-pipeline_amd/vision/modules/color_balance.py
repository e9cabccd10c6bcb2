"""Mirror of the reference modules/color_balance.py: `balance` (:93-110) and the `ColorBalance` module (:112-120).

The reference hands the frame to `process_frame` of libauv-color-balance.so (utils/color_correction/color_balance.cpp:343-780,
single-threaded C++ over OpenCV); here the same entry runs on the GPU (libvp `vp_color_balance_u8`: three reads and one
write of the frame).  `modules/preprocessor.py:87-88` imports `balance` from this module path.

Not implemented: tilings that do not divide the frame (the reference wraps into the next row there).
"""
import numpy as np

from vision import _vp
from vision.core.base import ModuleBase


def balance(mat, equalize_rgb=True, rgb_contrast_correct=False, hsv_contrast_correct=True, hsi_contrast_correct=False,
            rgb_extrema_clipping=True, adaptive_cast_correction=False, horizontal_blocks=1, vertical_blocks=1):
    """modules/color_balance.py:93-110: returns a new (rows, cols, 3) uint8 image."""
    mat = np.ascontiguousarray(mat, dtype=np.uint8)
    if mat.ndim != 3 or mat.shape[2] != 3 or mat.size == 0:
        raise ValueError("expected a non-empty (rows, cols, 3) BGR image")
    rows, cols = mat.shape[:2]
    flags = ((_vp.CB_EQUALIZE_RGB if equalize_rgb else 0) | (_vp.CB_RGB_CONTRAST if rgb_contrast_correct else 0) |
             (_vp.CB_HSV_CONTRAST if hsv_contrast_correct else 0) | (_vp.CB_HSI_CONTRAST if hsi_contrast_correct else 0) |
             (_vp.CB_EXTREMA_CLIPPING if rgb_extrema_clipping else 0) |
             (_vp.CB_ADAPTIVE_CAST if adaptive_cast_correction else 0))
    out = np.empty_like(mat)
    ctx = _vp.default_context()
    _vp.check(_vp.lib().vp_color_balance_u8(ctx.handle, _vp.ptr(mat), cols, rows, flags, int(horizontal_blocks), int(vertical_blocks),
                                            _vp.ptr(out)), ctx.handle)
    return out


class ColorBalance(ModuleBase):
    """modules/color_balance.py:112-120."""

    def process(self, *mat):
        mat = mat[0]
        self.post('orig', mat)
        mat = balance(mat)
        self.post('balanced', mat)
