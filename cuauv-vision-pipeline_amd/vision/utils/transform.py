"""Mirror of the reference utils/transform.py for the hot path.

Reference: utils/transform.py:27-77 (`elliptic_kernel`, `rect_kernel`), :80-112 (`erode`, `dilate`),
:115-164 (`morph_remove_noise`, `morph_close_holes`, `morph_borders`).  Kernels: libvp vp_morph_u8
(bit-plane LDS stencils for 0/255 masks with rect kernels, generic kernel otherwise).
"""
from typing import Optional

import numpy as np

from vision import _vp
from vision.devmat import DeviceMat, defer_enabled, lazy_enabled
from vision.utils.helpers import as_mat, device_image


def _structuring_element(shape, x, y):
    out = np.empty((y, x), np.uint8)
    _vp.check(_vp.lib().vp_structuring_element(shape, x, y, _vp.ptr(out)))
    return out


def elliptic_kernel(x: int, y: Optional[int] = None) -> np.ndarray:
    """utils/transform.py:27-51: odd positive sizes only (ValueError otherwise)."""
    if y is None:
        y = x
    if x % 2 == 0 or y % 2 == 0 or x <= 0 or y <= 0:
        raise ValueError("x and y must be odd positive integers")
    return _structuring_element(_vp.SHAPE_ELLIPSE, int(x), int(y))


def rect_kernel(x: int, y: Optional[int] = None) -> np.ndarray:
    """utils/transform.py:54-77."""
    if y is None:
        y = x
    if x <= 0 or y <= 0:
        raise ValueError("x and y must be positive integers")
    return _structuring_element(_vp.SHAPE_RECT, int(x), int(y))


def _morph(op, mat, kernel, iterations, anchor=(-1, -1)):
    mat = as_mat(mat)
    if not isinstance(mat, (np.ndarray, DeviceMat)) or mat.dtype != np.uint8 or mat.ndim not in (2, 3):
        raise TypeError("expected a uint8 (h, w) or (h, w, c) image")
    if kernel is None or np.size(kernel) == 0:
        kp, kw, kh = None, 0, 0
    else:
        kernel = np.ascontiguousarray(np.asarray(kernel) != 0, dtype=np.uint8)
        if kernel.ndim != 2:
            raise ValueError("kernel must be 2-D")
        kh, kw = kernel.shape
        kp = _vp.ptr(kernel)
    ctx = _vp.default_context()
    if (lazy_enabled() or isinstance(mat, DeviceMat)) and mat.size and (mat.ndim == 2 or mat.shape[2] <= 4):
        # device-resident: the result stays in HBM; a mask known to be 0/255 takes the bit-plane path without a flag read-back
        src = device_image(ctx, mat, 0)
        h, w = src.shape[:2]
        cn = 1 if src.ndim == 2 else src.shape[2]
        binary = src.binary
        ax, ay, it = int(anchor[0]), int(anchor[1]), int(iterations)

        def run(out, src=src, kernel=kernel):          # (keeps the source image and the kernel array alive until it has run)
            _vp.check(_vp.lib().vp_morph_u8_dev(ctx.handle, op, src.dev_ptr, w, h, cn, kp, kw, kh, ax, ay, it, 1 if binary else 0, out.dev_ptr), ctx.handle)
        if defer_enabled() and not src.host_escaped:   # (an image whose host copy is aliased by the caller can change without notice: launch now)
            if it < 0 or (kp is not None and (ax >= kw or ay >= kh)):
                raise _vp.VpError("libvp: invalid argument: structuring element")   # what the launch would report, reported at the call
            return DeviceMat.deferred(ctx, src.shape, np.uint8, binary, (src,), run)
        out = DeviceMat(ctx, src.shape, binary=binary)
        run(out)
        return out
    src = np.ascontiguousarray(mat)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    out = np.empty_like(src)
    _vp.check(_vp.lib().vp_morph_u8(ctx.handle, op, _vp.ptr(src), w, h, cn, kp, kw, kh, int(anchor[0]), int(anchor[1]),
                                    int(iterations), _vp.ptr(out)), ctx.handle)
    return out


def erode(mat: np.ndarray, kernel: np.ndarray, iterations: int = 1) -> np.ndarray:
    """utils/transform.py:80-94 (cv2.erode)."""
    return _morph(_vp.MORPH_ERODE, mat, kernel, iterations)


def dilate(mat: np.ndarray, kernel: np.ndarray, iterations: int = 1) -> np.ndarray:
    """utils/transform.py:97-112 (cv2.dilate)."""
    return _morph(_vp.MORPH_DILATE, mat, kernel, iterations)


def morph_remove_noise(mat: np.ndarray, kernel: np.ndarray, iterations: int = 1) -> np.ndarray:
    """utils/transform.py:115-129 (cv2.morphologyEx MORPH_OPEN)."""
    return _morph(_vp.MORPH_OPEN, mat, kernel, iterations)


def morph_close_holes(mat: np.ndarray, kernel: np.ndarray, iterations: int = 1) -> np.ndarray:
    """utils/transform.py:132-146 (cv2.morphologyEx MORPH_CLOSE)."""
    return _morph(_vp.MORPH_CLOSE, mat, kernel, iterations)


def morph_borders(mat: np.ndarray, kernel: np.ndarray, iterations: int = 1) -> np.ndarray:
    """utils/transform.py:149-164 (cv2.morphologyEx MORPH_GRADIENT)."""
    return _morph(_vp.MORPH_GRADIENT, mat, kernel, iterations)


def resize(mat: np.ndarray, width: int, height: int) -> np.ndarray:
    """utils/transform.py:167-179 (cv2.resize, default bilinear interpolation) on the GPU (libvp vp_resize_u8)."""
    from vision import cv2_facade
    return cv2_facade.resize(mat, (width, height))



def simple_gaussian_blur(mat: np.ndarray, kernel_size: int, std_dev: float) -> np.ndarray:
    """utils/transform.py:3-25 (cv2.GaussianBlur with a square kernel); ValueError for an even kernel size as in the reference."""
    if kernel_size % 2 == 0:
        raise ValueError("kernel_size must be an odd integer")
    from vision import cv2_facade
    return cv2_facade.GaussianBlur(mat, (kernel_size, kernel_size), std_dev)


def rotate(mat: np.ndarray, degrees: float) -> np.ndarray:
    """utils/transform.py:180-196: rotation about the image centre, positive = counterclockwise, borders replicated."""
    from vision import cv2_facade
    rot_mat = cv2_facade.getRotationMatrix2D((mat.shape[1] / 2, mat.shape[0] / 2), degrees, 1)
    return cv2_facade.warpAffine(mat, rot_mat, (mat.shape[1], mat.shape[0]), borderMode=cv2_facade.BORDER_REPLICATE)


def translate(mat: np.ndarray, x: int, y: int) -> np.ndarray:
    """utils/transform.py:199-215: shift by (x, y); uncovered pixels become 0.  The matrix goes through float32 as in the reference."""
    from vision import cv2_facade
    trans_mat = np.float32([[1, 0, x], [0, 1, y]])
    return cv2_facade.warpAffine(mat, trans_mat, (mat.shape[1], mat.shape[0]))


def decode_normal(mat: np.ndarray) -> np.ndarray:
    """utils/transform.py:218-233 (modules/normal.py:26): [0, 255] normal map back to float32 vectors in [-1, 1]; plain numpy in the
    reference as well, same operations in the same order."""
    img = np.asarray(mat).astype(np.float32)
    img = (img / 255.0) * 2.0 - 1.0
    return img
