"""Mirror of the reference utils/color.py for the hot path (same names, arguments, return structure).

Reference: utils/color.py:11-32 (`_convert_colorspace`: cv2.cvtColor + cv2.split), :105-121
(`range_threshold`: cv2.inRange), :66-103 (`thresh_color_distance`).  The arithmetic runs in libvp's
HIP kernels (include/vp.h: vp_cvt_color_u8, vp_inrange_u8, vp_inrange_f32, vp_color_distance_u8).
Returned arrays are fresh, writable, caller-owned numpy arrays, as with cv2.
"""
from math import sqrt
from typing import Callable, List, Tuple

import numpy as np

from vision import _vp
from vision.devmat import DeviceMat, finish_uploads, lazy_enabled, to_host
from vision.utils.helpers import as_mat, device_image


def _u8_image(mat, channels):
    mat = to_host(as_mat(mat))
    if not isinstance(mat, np.ndarray) or mat.dtype != np.uint8:
        raise TypeError("expected a uint8 numpy image")
    if channels == 3 and not (mat.ndim == 3 and mat.shape[2] == 3):
        raise ValueError("expected an (h, w, 3) image")
    if channels == 1:
        if mat.ndim == 3 and mat.shape[2] == 1:
            mat = mat[:, :, 0]
        if mat.ndim != 2:
            raise ValueError("expected an (h, w) image")
    if mat.shape[0] == 0 or mat.shape[1] == 0:
        raise ValueError("empty image")
    # rows must be dense; a row pitch is fine (views of wider images)
    if mat.strides[-1] != 1 or (mat.ndim == 3 and mat.strides[1] != mat.shape[2]) or mat.strides[0] < mat.shape[1] * (channels):
        mat = np.ascontiguousarray(mat)
    return mat


def _convert_colorspace(code: int) -> Callable[[np.ndarray], Tuple[np.ndarray, Tuple[np.ndarray, ...]]]:
    """utils/color.py:11-23: returns f(mat) -> (converted image, split channels)."""
    scn = 1 if code == _vp.GRAY2BGR else 3
    dcn = 1 if code == _vp.BGR2GRAY else 3
    want_planes = code != _vp.HSV2BGR

    def _inner_device(mat):
        """The image stays in HBM: results are DeviceMat (vision/devmat.py), nothing is downloaded here."""
        ctx = _vp.default_context()
        up = []                                       # a host image is copied while the results are set up and the kernel is enqueued
        src = device_image(ctx, mat, scn, pending=up)
        try:
            h, w = src.shape[:2]
            conv = DeviceMat(ctx, (h, w) if dcn == 1 else (h, w, 3))
            planes = [DeviceMat(ctx, (h, w)) for _ in range(dcn)] if (dcn == 3 and want_planes) else []
            arr = (_vp.C.c_void_p * 3)(*[p.dev_ptr for p in planes], *([None] * (3 - len(planes))))
            _vp.check(_vp.lib().vp_cvt_color_dev(ctx.handle, code, src.dev_ptr, w * scn, w, h, conv.dev_ptr, arr if planes else None), ctx.handle)
            if not want_planes:                           # HSV2BGR (colour-balance helper): channel copies are host views
                hc = conv.host(writable=False)
                return conv, tuple(np.ascontiguousarray(hc[:, :, c]) for c in range(3))
            if planes:
                return conv, tuple(planes)
            second = DeviceMat(ctx, (h, w))               # cv2.split of a single-channel image: a 1-tuple holding a copy
            _vp.check(_vp.lib().vp_cvt_color_dev(ctx.handle, code, src.dev_ptr, w * scn, w, h, second.dev_ptr, None), ctx.handle)
            return conv, (second,)
        finally:
            finish_uploads(ctx, up)                   # the caller's array may change from here on

    def _inner(mat: np.ndarray):
        mat = as_mat(mat)
        if lazy_enabled() or isinstance(mat, DeviceMat):
            return _inner_device(mat)
        mat = _u8_image(mat, scn)
        h, w = mat.shape[:2]
        conv = np.empty((h, w) if dcn == 1 else (h, w, 3), np.uint8)
        planes = [np.empty((h, w), np.uint8) for _ in range(dcn)] if (dcn == 3 and want_planes) else []
        arr = (_vp.C.c_void_p * 3)(*[p.ctypes.data for p in planes], *([None] * (3 - len(planes))))
        ctx = _vp.default_context()
        _vp.check(_vp.lib().vp_cvt_color_u8(ctx.handle, code, _vp.ptr(mat), mat.strides[0], w, h, _vp.ptr(conv),
                                            arr if planes else None), ctx.handle)
        # cv2.split of a single-channel image returns a 1-tuple holding a copy
        if not want_planes:
            return conv, tuple(np.ascontiguousarray(conv[:, :, c]) for c in range(3))
        return conv, (tuple(planes) if planes else (conv.copy(),))
    return _inner


def _unsupported(name):
    def _inner(mat):
        raise NotImplementedError(f"{name}: conversion is outside the accelerated hot path")
    return _inner


bgr_to_lab = _convert_colorspace(_vp.BGR2LAB)
bgr_to_hsv = _convert_colorspace(_vp.BGR2HSV)
bgr_to_gray = _convert_colorspace(_vp.BGR2GRAY)
gray_to_bgr = _convert_colorspace(_vp.GRAY2BGR)
bgr_to_hls = _convert_colorspace(_vp.BGR2HLS)
bgr_to_ycrcb = _convert_colorspace(_vp.BGR2YCRCB)
bgr_to_luv = _unsupported("bgr_to_luv")
lab_to_bgr = _unsupported("lab_to_bgr")
hsv_to_bgr = _convert_colorspace(_vp.HSV2BGR)


def bgr_to_lab_f32(mat: np.ndarray):
    """Extension: float32 BGR in [0,1] -> (lab float32 (h,w,3), (L, a, b) planes), analytic CIE L*a*b* (L 0..100)."""
    mat = np.ascontiguousarray(as_mat(mat), dtype=np.float32)
    if mat.ndim != 3 or mat.shape[2] != 3 or mat.size == 0:
        raise ValueError("expected a non-empty (h, w, 3) float image")
    h, w = mat.shape[:2]
    out = np.empty_like(mat)
    ctx = _vp.default_context()
    _vp.check(_vp.lib().vp_cvt_bgr2lab_f32(ctx.handle, _vp.ptr(mat), w, h, _vp.ptr(out)), ctx.handle)
    return out, tuple(np.ascontiguousarray(out[:, :, c]) for c in range(3))


def percentile_f32(arr: np.ndarray, q: float) -> float:
    """np.percentile(arr, q) (method 'linear') of a float32 array: the two neighbouring order statistics are selected on
    the GPU (libvp vp_order_stats_f32), the interpolation follows numpy's _lerp in float64."""
    flat = np.ascontiguousarray(arr, dtype=np.float32).ravel()
    n = flat.size
    if n == 0:
        raise ValueError("percentile of an empty array")
    virtual = (n - 1) * (float(q) / 100.0)
    lo = int(np.floor(virtual))
    lo = min(max(lo, 0), n - 1)
    gamma = virtual - lo
    a, b = _vp.C.c_float(), _vp.C.c_float()
    ctx = _vp.default_context()
    _vp.check(_vp.lib().vp_order_stats_f32(ctx.handle, _vp.ptr(flat), n, lo, _vp.C.byref(a), _vp.C.byref(b)), ctx.handle)
    a, b = np.float64(np.float32(a.value)), np.float64(np.float32(b.value))
    diff = b - a
    return float(b - diff * (1 - gamma)) if gamma >= 0.5 else float(a + diff * gamma)


def color_dist(c1, c2) -> float:
    """utils/color.py:35-48."""
    return sqrt((c1[0] - c2[0])**2 + (c1[1] - c2[1])**2 + (c1[2] - c2[2])**2)


def elementwise_color_dist(mat: np.ndarray, c) -> np.ndarray:
    """utils/color.py:51-63 (plain numpy in the reference too)."""
    return np.linalg.norm(as_mat(mat) - c, axis=2)


def _round_half_even(v):
    return int(np.rint(v))


def range_threshold(mat: np.ndarray, min, max) -> np.ndarray:
    """utils/color.py:105-121: cv2.inRange(mat, min, max) -> 0/255 mask.

    uint8 input with scalar or per-channel bounds (rounded half-to-even like cv2's scalar
    conversion), or float32 single-channel input (the `dists` image of thresh_color_distance)."""
    mat = as_mat(mat)
    ctx = _vp.default_context()
    if isinstance(mat, DeviceMat) and mat.dtype != np.uint8:
        mat = mat.host()
    if isinstance(mat, np.ndarray) and mat.dtype == np.float32:
        if mat.ndim == 3 and mat.shape[2] == 1:
            mat = mat[:, :, 0]
        if mat.ndim != 2:
            raise ValueError("float32 inRange supports single-channel images")
        if mat.strides[1] != 4:
            mat = np.ascontiguousarray(mat)
        h, w = mat.shape
        out = np.empty((h, w), np.uint8)
        lo = float(np.float32(np.ravel(min)[0] if np.ndim(min) else min))
        hi = float(np.float32(np.ravel(max)[0] if np.ndim(max) else max))
        _vp.check(_vp.lib().vp_inrange_f32(ctx.handle, _vp.ptr(mat), mat.strides[0], w, h, lo, hi, _vp.ptr(out)), ctx.handle)
        return out
    cn = 3 if (isinstance(mat, (np.ndarray, DeviceMat)) and mat.ndim == 3 and mat.shape[2] == 3) else 1
    on_device = lazy_enabled() or isinstance(mat, DeviceMat)
    up = []
    mat = device_image(ctx, mat, cn, pending=up) if on_device else _u8_image(mat, cn)
    h, w = mat.shape[:2]

    def bounds(b):
        if cn == 1 and type(b) in (int, float):          # the common call: Python numbers for one plane
            v = float(b)
            if -1e18 < v < 1e18:                         # (NaN and infinities take the general path below)
                r = round(v)                             # half to even, like np.rint (`min` / `max` are this function's parameters)
                return np.array([-2**31 if r < -2**31 else 2**31 - 1 if r > 2**31 - 1 else r], np.int32)
        b = np.atleast_1d(np.asarray(b, dtype=np.float64)).ravel()
        if b.size == 1 and cn == 3:
            b = np.array([b[0], 0.0, 0.0])  # cv2 scalar -> (v, 0, 0, 0)
        if b.size < cn:
            raise ValueError("bounds need one value per channel")
        return np.ascontiguousarray(np.clip(np.rint(b[:cn]), -2**31, 2**31 - 1).astype(np.int32))
    try:
        lo, hi = bounds(min), bounds(max)
    except BaseException:
        finish_uploads(ctx, up)
        raise
    if on_device:
        try:
            out = DeviceMat(ctx, (h, w), binary=True)
            if w % 64 == 0:
                # the mask's bit-packed form comes out of the same launch (1/8 B/px more): the contour pass of this very mask
                # (modules/red_buoy.py:38 outer_contours(threshed)) then needs no packing launch of its own
                from vision.devmat import _DevBuf
                bits = _DevBuf(ctx, h * (w // 64) * 8)
                made = _vp.C.c_int(0)
                _vp.check(_vp.lib().vp_inrange_u8_bits_dev(ctx.handle, mat.dev_ptr, w * cn, w, h, cn, _vp.ptr(lo), _vp.ptr(hi), out.dev_ptr, bits.ptr,
                                                          _vp.C.byref(made)), ctx.handle)
                if made.value:
                    out._bits = bits
                return out
            _vp.check(_vp.lib().vp_inrange_u8_dev(ctx.handle, mat.dev_ptr, w * cn, w, h, cn, _vp.ptr(lo), _vp.ptr(hi), out.dev_ptr), ctx.handle)
            return out
        finally:
            finish_uploads(ctx, up)
    out = np.empty((h, w), np.uint8)
    _vp.check(_vp.lib().vp_inrange_u8(ctx.handle, _vp.ptr(mat), mat.strides[0], w, h, cn, _vp.ptr(lo), _vp.ptr(hi), _vp.ptr(out)), ctx.handle)
    return out


def binary_threshold(mat: np.ndarray, threshold: int) -> np.ndarray:
    """utils/color.py:124-137 (cv2.threshold THRESH_BINARY on uint8): > threshold -> 255."""
    return range_threshold(mat, int(np.floor(threshold)) + 1, 255)


def binary_threshold_inv(mat: np.ndarray, threshold: int) -> np.ndarray:
    """utils/color.py:140-153: <= threshold -> 255."""
    return range_threshold(mat, 0, int(np.floor(threshold)))


def thresh_color_distance(split: List[np.ndarray], color, distance: float, auto_distance_percentile: float = None,
                          ignore_channels: List[int] = [], weights=(1, 1, 1)) -> Tuple[np.ndarray, np.ndarray]:
    """utils/color.py:66-103.  d2 = sum_i w_i * (float32(split_i) - color_i)^2 in float32 (numpy 1.x
    scalar semantics: the normalised float64 weight multiplies a float32 array as float32), mask =
    inRange(d2, 0, distance^2 [or the percentile]), second result uint8(sqrt(d2))."""
    weights_cp = list(weights)
    for idx in ignore_channels:
        weights_cp[idx] = 0
    weights_cp = np.asarray(weights_cp, dtype=np.float64) / np.linalg.norm(weights)
    planes = [_u8_image(p, 1) for p in split[:3]]
    h, w = planes[0].shape
    planes = [np.ascontiguousarray(p) for p in planes]
    skip = 0
    for i in range(3):
        if i in ignore_channels:
            skip |= 1 << i
    col = np.ascontiguousarray(np.asarray([color[0], color[1], color[2]], dtype=np.float32))
    wts = np.ascontiguousarray(weights_cp.astype(np.float32))
    d2 = np.empty((h, w), np.float32)
    sq = np.empty((h, w), np.uint8)
    arr = (_vp.C.c_void_p * 3)(*[p.ctypes.data for p in planes])
    ctx = _vp.default_context()
    _vp.check(_vp.lib().vp_color_distance_u8(ctx.handle, arr, w, h, _vp.ptr(col), _vp.ptr(wts), skip, _vp.ptr(d2), _vp.ptr(sq)), ctx.handle)
    if auto_distance_percentile:
        distance = min(percentile_f32(d2, auto_distance_percentile), distance**2)
    else:
        distance = distance**2
    return range_threshold(d2, 0, distance), sq


def _outside_path(name):
    def _f(*_a, **_k):
        raise NotImplementedError(f"{name}: outside the accelerated path of this build")
    _f.__name__ = name
    return _f


# the rest of the reference's utils/color.py (:156-392): names kept so that `from vision.utils.color import ...` lines of existing modules
# import; calling them fails loudly instead of falling back to a CPU implementation


def _threshold(mat: np.ndarray, thresh: float, maxval: float, kind: int) -> np.ndarray:
    """cv2.threshold(mat, thresh, maxval, kind)[1] on uint8 images (libvp vp_threshold_u8)."""
    mat = to_host(as_mat(mat))
    if not isinstance(mat, np.ndarray) or mat.dtype != np.uint8 or mat.size == 0:
        raise TypeError("expected a non-empty uint8 numpy image")
    mat = np.ascontiguousarray(mat)
    out = np.empty_like(mat)
    ctx = _vp.default_context()
    _vp.check(_vp.lib().vp_threshold_u8(ctx.handle, _vp.ptr(mat), mat.size, float(thresh), float(maxval), int(kind), _vp.ptr(out)), ctx.handle)
    return out


def max_threshold(mat: np.ndarray, threshold: float) -> np.ndarray:
    """utils/color.py:156-169 (THRESH_TRUNC): values above the threshold become the threshold."""
    return _threshold(mat, threshold, 0, 2)


def above_threshold(mat: np.ndarray, threshold: float) -> np.ndarray:
    """utils/color.py:172-185 (THRESH_TOZERO): values above the threshold are kept, the rest become zero."""
    return _threshold(mat, threshold, 0, 3)


def below_threshold(mat: np.ndarray, threshold: float) -> np.ndarray:
    """utils/color.py:188-199 (THRESH_TOZERO_INV): values above the threshold become zero, the rest are kept."""
    return _threshold(mat, threshold, 0, 4)



def otsu_threshold(mat: np.ndarray):
    """utils/color.py:204-217 (cv2.threshold(mat, 0, 255, THRESH_OTSU)): (threshold chosen by Otsu's method, thresholded image)."""
    mat = _u8_image(mat, 1)
    mat = np.ascontiguousarray(mat)
    out = np.empty_like(mat)
    t = _vp.C.c_double(0)
    ctx = _vp.default_context()
    _vp.check(_vp.lib().vp_otsu_threshold_u8(ctx.handle, _vp.ptr(mat), mat.size, 255.0, 0, _vp.C.byref(t), _vp.ptr(out)), ctx.handle)
    return t.value, out



def _adaptive_mean(mat: np.ndarray, neighborhood_size: int, bias: float, kind: int) -> np.ndarray:
    mat = np.ascontiguousarray(_u8_image(mat, 1))
    out = np.empty_like(mat)
    ctx = _vp.default_context()
    _vp.check(_vp.lib().vp_adaptive_threshold_mean_u8(ctx.handle, _vp.ptr(mat), mat.shape[1], mat.shape[0], 255.0, kind, int(neighborhood_size),
                                                      float(bias), _vp.ptr(out)), ctx.handle)
    return out


def adaptive_threshold_mean(mat: np.ndarray, neighborhood_size: int, bias: float = 0) -> np.ndarray:
    """utils/color.py:220-235 (cv2.adaptiveThreshold, ADAPTIVE_THRESH_MEAN_C, THRESH_BINARY): 255 where the pixel exceeds the mean of
    its neighbourhood minus the bias."""
    return _adaptive_mean(mat, neighborhood_size, bias, 0)


def adaptive_threshold_mean_inv(mat: np.ndarray, neighborhood_size: int, bias: float = 0) -> np.ndarray:
    """utils/color.py:238-254 (THRESH_BINARY_INV): 255 where the pixel does not exceed the mean of its neighbourhood minus the bias."""
    return _adaptive_mean(mat, neighborhood_size, bias, 1)


adaptive_threshold_gaussian = _outside_path("adaptive_threshold_gaussian")
adaptive_threshold_gaussian_inv = _outside_path("adaptive_threshold_gaussian_inv")
kmeans = _outside_path("kmeans")


def mask_from_labels(labels: np.ndarray, centers: np.ndarray) -> List[np.ndarray]:
    """utils/color.py:326-345: one 0 / 255 mask per centre (plain numpy in the reference as well)."""
    acc = []
    for i, _c in enumerate(centers):
        mask = np.zeros(labels.shape, dtype=np.uint8)
        mask[labels == i] = 255
        acc.append(mask)
    return acc


mask_from_labels_target_color = _outside_path("mask_from_labels_target_color")
white_balance_bgr = _outside_path("white_balance_bgr")
white_balance_bgr_blur = _outside_path("white_balance_bgr_blur")
