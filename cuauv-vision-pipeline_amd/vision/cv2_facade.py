"""A narrow `cv2` look-alike for module files that call OpenCV directly (modules/bins.py:13-75 and the colour /
morphology lines of modules/preprocessor.py:52-129), so they run where no OpenCV exists.

`install()` registers this module as `cv2` ONLY when a real cv2 cannot be imported — a real OpenCV is never shadowed.
Image arithmetic (cvtColor, inRange, erode/dilate/morphologyEx, findContours, connectedComponentsWithStats) goes to
libvp (HIP); small polygon maths (moments, contourArea, arcLength, minAreaRect, boxPoints, approxPolyDP) and overlay
helpers (addWeighted, drawContours, split/merge) are host numpy, as they are CPU code in OpenCV too.  Anything else
raises AttributeError like a missing cv2 symbol would."""
import math
import sys

import numpy as np

from vision import _vp
from vision.utils import color as _color
from vision.utils import draw as _draw
from vision.utils import feature as _feature
from vision.utils import transform as _transform

COLOR_BGR2LAB, COLOR_BGR2HSV, COLOR_BGR2GRAY, COLOR_GRAY2BGR, COLOR_HSV2BGR = 44, 40, 6, 8, 54     # cv2's own enum values
COLOR_BGR2YCrCb, COLOR_BGR2YCR_CB, COLOR_BGR2HLS = 36, 36, 52
COLOR_BGR2Luv, COLOR_BGR2LUV = 50, 50     # named by modules/preprocessor.py:76; see DESIGN.md section 7 for what the stand-in does with it
MORPH_RECT, MORPH_CROSS, MORPH_ELLIPSE = 0, 1, 2
MORPH_ERODE, MORPH_DILATE, MORPH_OPEN, MORPH_CLOSE, MORPH_GRADIENT = 0, 1, 2, 3, 4
RETR_EXTERNAL, RETR_LIST = 0, 1
CHAIN_APPROX_NONE, CHAIN_APPROX_SIMPLE = 1, 2
CC_STAT_LEFT, CC_STAT_TOP, CC_STAT_WIDTH, CC_STAT_HEIGHT, CC_STAT_AREA = 0, 1, 2, 3, 4
CV_8U = 0
CV_32S = 4


class UMat:   # only so that `from cv2 import UMat` and isinstance checks work
    def __init__(self, arr):
        self._arr = np.asarray(arr)

    def get(self):
        return self._arr


class error(Exception):
    pass


_CVT = {COLOR_BGR2LAB: _color.bgr_to_lab, COLOR_BGR2HSV: _color.bgr_to_hsv, COLOR_BGR2GRAY: _color.bgr_to_gray,
        COLOR_GRAY2BGR: _color.gray_to_bgr, COLOR_HSV2BGR: _color.hsv_to_bgr, COLOR_BGR2YCrCb: _color.bgr_to_ycrcb,
        COLOR_BGR2HLS: _color.bgr_to_hls}


def cvtColor(src, code):
    if code not in _CVT:
        raise error(f"cvtColor code {code} is outside the accelerated path")
    return _CVT[code](src)[0]


def split(m):
    m = np.asarray(m)
    return tuple(np.ascontiguousarray(m[:, :, c]) for c in range(m.shape[2])) if m.ndim == 3 else (m.copy(),)


def merge(mv):
    return np.ascontiguousarray(np.dstack(list(mv)))


def inRange(src, lowerb, upperb):
    return _color.range_threshold(src, lowerb, upperb)


def getStructuringElement(shape, ksize):
    return _transform._structuring_element(int(shape), int(ksize[0]), int(ksize[1]))


BORDER_CONSTANT = 0
BORDER_REPLICATE = 1
BORDER_REFLECT_101 = 4
BORDER_DEFAULT = 4


def _into(dst, result):
    """cv2's optional `dst` argument: the result is also written into it when it has the right shape and type (cv2 reallocates
    otherwise, which a caller-owned numpy array cannot do: the returned array is the result either way)."""
    if dst is not None:
        from vision.devmat import to_host
        d = to_host(dst)
        r = np.asarray(result)
        if isinstance(d, np.ndarray) and d.shape == r.shape and d.dtype == r.dtype and d.flags.writeable:
            np.copyto(d, r)
            return dst
    return result


def _default_border_only(name, borderType, borderValue):
    # cv2's default for morphology: BORDER_CONSTANT with morphologyDefaultBorderValue() = "outside never wins"
    if borderType not in (None, BORDER_CONSTANT) or borderValue is not None:
        raise error(f"{name}: only the default border (constant, morphologyDefaultBorderValue) is on the accelerated path")


# positional order as in cv2: (src, kernel, dst, anchor, iterations, borderType, borderValue)
def erode(src, kernel, dst=None, anchor=None, iterations=1, borderType=None, borderValue=None):
    _default_border_only("erode", borderType, borderValue)
    return _into(dst, _transform._morph(_vp.MORPH_ERODE, src, kernel, iterations, anchor if anchor is not None else (-1, -1)))


def dilate(src, kernel, dst=None, anchor=None, iterations=1, borderType=None, borderValue=None):
    _default_border_only("dilate", borderType, borderValue)
    return _into(dst, _transform._morph(_vp.MORPH_DILATE, src, kernel, iterations, anchor if anchor is not None else (-1, -1)))


# (src, op, kernel, dst, anchor, iterations, borderType, borderValue)
def morphologyEx(src, op, kernel, dst=None, anchor=None, iterations=1, borderType=None, borderValue=None):
    _default_border_only("morphologyEx", borderType, borderValue)
    return _into(dst, _transform._morph(int(op), src, kernel, iterations, anchor if anchor is not None else (-1, -1)))


def findContours(image, mode, method):
    """-> (contours, hierarchy); hierarchy is None (RETR_EXTERNAL / RETR_LIST are flat)."""
    return _feature.find_contours(image, int(mode), int(method)), None


def connectedComponentsWithStats(image, connectivity=8, ltype=CV_32S):
    if connectivity != 8:
        raise error("only 8-connectivity is implemented")
    if ltype != CV_32S:
        raise error("only CV_32S labels are implemented")
    cap = 4096
    while True:                                    # stats and centroids always have one row per label, as in cv2
        n, labels, stats, cent = _feature.connected_components(image, max_labels=cap)
        if n <= cap:
            return n, labels, stats, cent
        cap = n


def moments(contour):
    m00, m10, m01 = _feature._polygon_moments(contour)
    return {"m00": m00, "m10": m10, "m01": m01}


def contourArea(contour, oriented=False):
    pts = np.asarray(contour).reshape(-1, 2).astype(np.float64)
    if len(pts) == 0:
        return 0.0
    x, y = pts[:, 0], pts[:, 1]
    a = 0.5 * float(np.sum(np.roll(x, 1) * y - np.roll(y, 1) * x))
    return a if oriented else abs(a)


def arcLength(curve, closed):
    """imgproc/src/shapedescr.cpp arcLength: segment lengths in float32, summed in float64."""
    pts = np.asarray(curve).reshape(-1, 2).astype(np.float32)
    if len(pts) <= 1:
        return 0.0
    prev = pts[-1] if closed else pts[0]
    total = 0.0
    for i in range(0 if closed else 1, len(pts)):
        d = pts[i] - prev
        total += float(np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1])))
        prev = pts[i]
    return total


def _convex_hull_py(pts):
    pts = sorted(set(map(tuple, pts.tolist())))
    if len(pts) <= 2:
        return np.array(pts, np.float64)

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    lower, upper = [], []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return np.array(lower[:-1] + upper[:-1], np.float64)


def _convex_hull(points):
    """Monotone-chain hull as float64 (k, 2): integer points (contours) through libvp's exact host routine, anything else in Python."""
    p = np.asarray(points).reshape(-1, 2)
    if np.issubdtype(p.dtype, np.integer) and len(p) and np.abs(p).max(initial=0) < 2**31:
        p32 = np.ascontiguousarray(p, np.int32)
        out = np.empty_like(p32)
        n = _vp.C.c_int(0)
        _vp.check(_vp.lib().vp_convex_hull_i32(p32.ctypes.data, len(p32), out.ctypes.data, _vp.C.byref(n)))
        return out[:n.value].astype(np.float64)
    return _convex_hull_py(p.astype(np.float64))


def minAreaRect(points):
    """Minimum-area enclosing rectangle by rotating calipers over the convex hull: ((cx, cy), (w, h), angle in degrees).
    Angle convention of OpenCV >= 4.5.1: in (0, 90], width measured along the edge that defines the angle."""
    p = np.asarray(points).reshape(-1, 2)
    # contours: hull and calipers in libvp's host routine (the doubles of _min_area_rect_loop, statement by statement).  What
    # find_contours returns is int32 already: no range check, no copy (numpy calls on a handful of points cost more than the calipers)
    if len(p) and (p.dtype == np.int32 or (np.issubdtype(p.dtype, np.integer) and np.abs(p).max(initial=0) < 2**31)):
        p32 = p if (p.dtype == np.int32 and p.flags.c_contiguous) else np.ascontiguousarray(p, np.int32)
        out = (_vp.C.c_float * 5)()
        _vp.check(_vp.lib().vp_min_area_rect_i32(p32.ctypes.data, len(p32), out))
        return (out[0], out[1]), (out[2], out[3]), out[4]
    hull = _convex_hull(points)
    if len(hull) == 0:
        return (0.0, 0.0), (0.0, 0.0), 0.0
    if len(hull) == 1:
        return (float(hull[0, 0]), float(hull[0, 1])), (0.0, 0.0), 90.0
    n = len(hull)
    # every edge at once (columns): the same products and sums as edge by edge, the first edge of minimal area wins
    e = (np.roll(hull, -1, axis=0) - hull)[: (n if n > 2 else 1)]
    ln = np.array([math.sqrt(x * x + y * y) for x, y in e.tolist()])
    keep = ln != 0
    if not keep.any():
        return (float(hull[0, 0]), float(hull[0, 1])), (0.0, 0.0), 90.0
    e, ln = e[keep], ln[keep]
    ux, uy = e[:, 0] / ln, e[:, 1] / ln
    hx, hy = hull[:, 0][:, None], hull[:, 1][:, None]
    a = hx * ux + hy * uy
    b = -hx * uy + hy * ux
    amax, amin, bmax, bmin = a.max(0), a.min(0), b.max(0), b.min(0)
    wds, hts = amax - amin, bmax - bmin
    k = int(np.argmin(wds * hts))
    ca, cb = (amax[k] + amin[k]) / 2, (bmax[k] + bmin[k]) / 2
    uxk, uyk = float(ux[k]), float(uy[k])
    cx, cy, wd, ht, ang = ca * uxk - cb * uyk, ca * uyk + cb * uxk, float(wds[k]), float(hts[k]), math.degrees(math.atan2(uyk, uxk))
    while ang <= 0:
        ang += 90
        wd, ht = ht, wd
    while ang > 90:
        ang -= 90
        wd, ht = ht, wd
    return (float(np.float32(cx)), float(np.float32(cy))), (float(np.float32(wd)), float(np.float32(ht))), float(np.float32(ang))


def _min_area_rect_loop(points):
    """The same statements edge by edge (the form the vectorised one is tested against)."""
    pts = np.asarray(points).reshape(-1, 2).astype(np.float64)
    hull = _convex_hull_py(pts)
    if len(hull) == 0:
        return (0.0, 0.0), (0.0, 0.0), 0.0
    if len(hull) == 1:
        return (float(hull[0, 0]), float(hull[0, 1])), (0.0, 0.0), 90.0
    best = None
    n = len(hull)
    for i in range(n if n > 2 else 1):
        e = hull[(i + 1) % n] - hull[i]
        ln = math.sqrt(e[0] * e[0] + e[1] * e[1])     # (exact under the root for integer points)
        if ln == 0:
            continue
        ux, uy = e[0] / ln, e[1] / ln
        a = hull[:, 0] * ux + hull[:, 1] * uy
        b = -hull[:, 0] * uy + hull[:, 1] * ux
        wd, ht = a.max() - a.min(), b.max() - b.min()
        if best is None or wd * ht < best[0]:
            ca, cb = (a.max() + a.min()) / 2, (b.max() + b.min()) / 2
            best = (wd * ht, (ca * ux - cb * uy, ca * uy + cb * ux), wd, ht, math.degrees(math.atan2(uy, ux)))
    _, (cx, cy), wd, ht, ang = best
    while ang <= 0:
        ang += 90
        wd, ht = ht, wd
    while ang > 90:
        ang -= 90
        wd, ht = ht, wd
    return (float(np.float32(cx)), float(np.float32(cy))), (float(np.float32(wd)), float(np.float32(ht))), float(np.float32(ang))


def boxPoints(box):
    """Corners of a rotated rect, in OpenCV's order (RotatedRect::points): bottom-left, top-left, top-right, bottom-right."""
    (cx, cy), (w, h), ang = box
    a = math.radians(ang)
    b, c = math.cos(a) * 0.5, math.sin(a) * 0.5
    p0 = (cx - c * h - b * w, cy + b * h - c * w)   # a = sin*0.5, b = cos*0.5 in OpenCV's source
    p1 = (cx + c * h - b * w, cy - b * h - c * w)
    p2 = (2 * cx - p0[0], 2 * cy - p0[1])
    p3 = (2 * cx - p1[0], 2 * cy - p1[1])
    return np.array([p0, p1, p2, p3], np.float32)


def approxPolyDP(curve, epsilon, closed):
    """Douglas-Peucker on the point list (closed curves are split at the two mutually farthest points)."""
    pts = np.asarray(curve).reshape(-1, 2).astype(np.float64)
    n = len(pts)
    if n <= 2:
        return np.asarray(curve).reshape(-1, 1, 2).copy()

    def rdp(lo, hi, keep):
        a, b = pts[lo], pts[hi % n]
        idx = [i % n for i in range(lo + 1, hi)]
        if not idx:
            return
        d = b - a
        ln = math.hypot(d[0], d[1])
        seg = pts[idx]
        dist = np.abs(d[0] * (seg[:, 1] - a[1]) - d[1] * (seg[:, 0] - a[0])) / ln if ln > 0 else np.hypot(seg[:, 0] - a[0], seg[:, 1] - a[1])
        k = int(np.argmax(dist))
        if dist[k] > epsilon:
            m = lo + 1 + k
            keep.add(m % n)
            rdp(lo, m, keep)
            rdp(m, hi, keep)
    keep = set()
    if closed:
        d0 = np.hypot(pts[:, 0] - pts[0, 0], pts[:, 1] - pts[0, 1])
        far = int(np.argmax(d0))
        d1 = np.hypot(pts[:, 0] - pts[far, 0], pts[:, 1] - pts[far, 1])
        start = int(np.argmax(d1))
        keep.update((start, far))
        a, b = sorted((start, far))
        rdp(a, b, keep)
        rdp(b, a + n, keep)
    else:
        keep.update((0, n - 1))
        rdp(0, n - 1, keep)
    out = np.asarray(curve).reshape(-1, 2)[sorted(keep)]
    return out.reshape(-1, 1, 2).copy()


def addWeighted(src1, alpha, src2, beta, gamma, dst=None, dtype=-1):
    """saturate_cast<uchar>(src1*alpha + src2*beta + gamma) with round-half-even (modules/bins.py:20), every step a double.  Two uint8
    images of one shape stay on the device (libvp vp_add_weighted_u8_dev: the same doubles) and the result is computed when something
    reads it - bins.py draws into the overlay only when it has found rectangles, and posts it only when posts are on."""
    from vision.devmat import DeviceMat, defer_enabled, finish_uploads, lazy_enabled
    from vision.utils.helpers import as_mat, device_image
    a, b = as_mat(src1), as_mat(src2)
    on_dev = (lazy_enabled() or isinstance(a, DeviceMat) or isinstance(b, DeviceMat)) and dtype in (-1, CV_8U) and \
        all(isinstance(m, (np.ndarray, DeviceMat)) and m.dtype == np.uint8 and m.ndim in (2, 3) and m.size for m in (a, b)) and \
        tuple(a.shape) == tuple(b.shape)
    if not on_dev:
        acc = np.asarray(a, np.float64) * alpha + np.asarray(b, np.float64) * beta + gamma
        return _into(dst, np.clip(np.rint(acc), 0, 255).astype(np.uint8))
    ctx = _vp.default_context()
    up = []
    try:
        da = device_image(ctx, a, 0, pending=up)
        db = device_image(ctx, b, 0, pending=up)
        fa, fb, fg = float(alpha), float(beta), float(gamma)
        n = int(np.prod(da.shape))

        def run(out, da=da, db=db):
            _vp.check(_vp.lib().vp_add_weighted_u8_dev(ctx.handle, da.dev_ptr, fa, db.dev_ptr, fb, fg, n, out.dev_ptr), ctx.handle)
        if defer_enabled() and not (da.host_escaped or db.host_escaped):
            out = DeviceMat.deferred(ctx, da.shape, np.uint8, False, (da, db), run)
        else:
            out = DeviceMat(ctx, da.shape)
            run(out)
    finally:
        finish_uploads(ctx, up)
    return _into(dst, out)


def add(src1, src2):
    """cv2.add with saturation; one operand may be a scalar (modules/preprocessor.py:90-103 adds a bias to a channel)."""
    a, b = (src1, src2) if isinstance(src1, np.ndarray) else (src2, src1)
    a = np.asarray(a)
    if a.dtype != np.uint8:
        raise error("add: only uint8 arrays are on the accelerated path")
    if np.isscalar(b):
        acc = a.astype(np.float64) + float(b)          # cv2 adds the scalar as a double, then saturate_cast (round half even)
        return np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    return np.clip(a.astype(np.int32) + np.asarray(b, np.int32), 0, 255).astype(np.uint8)


def getRotationMatrix2D(center, angle, scale):
    """imgproc/src/imgwarp.cpp getRotationMatrix2D: 2x3 float64 (modules/preprocessor.py:131-133)."""
    ang = angle * np.pi / 180.0
    a, b = scale * np.cos(ang), scale * np.sin(ang)
    cx, cy = float(np.float32(center[0])), float(np.float32(center[1]))   # the centre is a Point2f
    return np.array([[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy]], np.float64)


def GaussianBlur(src, ksize, sigmaX, dst=None, sigmaY=0, borderType=None):
    """cv2.GaussianBlur on uint8 images (modules/preprocessor.py:110-114): OpenCV's bit-exact fixed-point path on the GPU.
    Positional order as in cv2: (src, ksize, sigmaX, dst, sigmaY, borderType)."""
    return _into(dst, _gaussian_blur(src, ksize, sigmaX, sigmaY, borderType))


def _gaussian_blur(src, ksize, sigmaX, sigmaY, borderType):
    from vision import _vp
    if borderType not in (None, BORDER_DEFAULT):
        raise error("GaussianBlur: only BORDER_DEFAULT (reflect 101) is on the accelerated path")
    src = np.ascontiguousarray(src)
    if src.dtype != np.uint8 or src.ndim not in (2, 3) or src.size == 0:
        raise error("GaussianBlur: only non-empty uint8 images are on the accelerated path")
    cn = 1 if src.ndim == 2 else src.shape[2]
    kw, kh = int(ksize[0]), int(ksize[1])
    if kw <= 0 or kh <= 0 or kw % 2 == 0 or kh % 2 == 0 or kw > 511 or kh > 511 or cn > 4:
        raise error("GaussianBlur: kernel sizes must be odd, 1..511")
    out = np.empty_like(src)
    ctx = _vp.default_context()
    _vp.check(_vp.lib().vp_gaussian_blur_u8(ctx.handle, _vp.ptr(src), src.shape[1], src.shape[0], cn, kw, kh, float(sigmaX), float(sigmaY),
                                            _vp.ptr(out)), ctx.handle)
    return out


INTER_LINEAR = 1
WARP_INVERSE_MAP = 16


def warpAffine(src, M, dsize, dst=None, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0):
    """cv2.warpAffine with bilinear interpolation on uint8 images (modules/preprocessor.py:130-135,145-149): OpenCV's classical
    fixed-point path on the GPU (libvp vp_warp_affine_u8).  Positional order as in cv2: (src, M, dsize, dst, flags, borderMode,
    borderValue)."""
    return _into(dst, _warp_affine(src, M, dsize, flags, borderMode, borderValue))


def _warp_affine(src, M, dsize, flags, borderMode, borderValue):
    from vision import _vp
    if flags is None:
        flags = INTER_LINEAR
    if (flags & ~WARP_INVERSE_MAP) != INTER_LINEAR:
        raise error("warpAffine: only INTER_LINEAR is on the accelerated path")
    if borderMode not in (BORDER_CONSTANT, BORDER_REPLICATE):
        raise error("warpAffine: only BORDER_CONSTANT and BORDER_REPLICATE are on the accelerated path")
    src = np.ascontiguousarray(src)
    if src.dtype != np.uint8 or src.ndim not in (2, 3) or src.size == 0:
        raise error("warpAffine: expected a non-empty uint8 image")
    cn = 1 if src.ndim == 2 else src.shape[2]
    m = np.ascontiguousarray(np.asarray(M, dtype=np.float64))
    dw, dh = int(dsize[0]), int(dsize[1])
    if m.shape != (2, 3) or dw <= 0 or dh <= 0 or cn > 4:
        raise error("warpAffine: M must be 2x3 and the size positive")
    bv = np.zeros(4, np.uint8)
    vals = np.atleast_1d(np.asarray(borderValue, dtype=np.float64))[:4]
    bv[:len(vals)] = np.clip(np.rint(vals), 0, 255).astype(np.uint8)
    out = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.uint8)
    ctx = _vp.default_context()
    _vp.check(_vp.lib().vp_warp_affine_u8(ctx.handle, _vp.ptr(src), src.shape[1], src.shape[0], cn, _vp.ptr(m), int(flags & WARP_INVERSE_MAP),
                                          int(borderMode), _vp.ptr(bv), _vp.ptr(out), dw, dh), ctx.handle)
    return out


def resize(src, dsize, dst=None, fx=None, fy=None, interpolation=INTER_LINEAR):
    """cv2.resize(src, (width, height)) with the default bilinear interpolation, uint8 images (modules/preprocessor.py:136-144).
    Positional order as in cv2: (src, dsize, dst, fx, fy, interpolation); dsize None or (0, 0) takes the size from fx / fy."""
    from vision import _vp
    if interpolation != INTER_LINEAR:
        raise error("resize: only INTER_LINEAR is on the accelerated path")
    src = np.ascontiguousarray(src)
    if dsize is None or tuple(dsize) == (0, 0):
        if not fx or not fy:
            raise error("resize: dsize or both fx and fy are needed")
        dsize = (int(round(src.shape[1] * fx)), int(round(src.shape[0] * fy)))   # cv2: saturate_cast<int>(cols * fx): round half to even
    elif fx or fy:
        pass                                       # cv2 ignores fx / fy when dsize is given
    if src.dtype != np.uint8 or src.ndim not in (2, 3) or src.size == 0:
        raise error("resize: expected a non-empty uint8 image")
    cn = 1 if src.ndim == 2 else src.shape[2]
    dw, dh = int(dsize[0]), int(dsize[1])
    if dw <= 0 or dh <= 0 or cn > 4:
        raise error("resize: bad size")
    out = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.uint8)
    ctx = _vp.default_context()
    _vp.check(_vp.lib().vp_resize_u8(ctx.handle, _vp.ptr(src), src.shape[1], src.shape[0], cn, dw, dh, _vp.ptr(out)), ctx.handle)
    return _into(dst, out)


def Canny(image, threshold1, threshold2, apertureSize=3, L2gradient=False):
    """cv2.Canny on uint8 images with the defaults the reference uses (utils/feature.py:66,101): 3x3 Sobel, L1 gradient magnitude."""
    from vision import _vp
    if apertureSize != 3 or L2gradient:
        raise error("Canny: only apertureSize=3 with the L1 gradient is on the accelerated path")
    image = np.ascontiguousarray(image)
    if image.dtype != np.uint8 or image.ndim not in (2, 3) or image.size == 0:
        raise error("Canny: expected a non-empty uint8 image")
    cn = 1 if image.ndim == 2 else image.shape[2]
    if cn > 4:
        raise error("Canny: at most 4 channels")
    out = np.empty(image.shape[:2], np.uint8)
    ctx = _vp.default_context()
    _vp.check(_vp.lib().vp_canny_u8(ctx.handle, _vp.ptr(image), image.shape[1], image.shape[0], cn, float(threshold1), float(threshold2),
                                    _vp.ptr(out)), ctx.handle)
    return out


def drawContours(image, contours, contourIdx, color, thickness=1):
    sel = contours if contourIdx < 0 else [contours[contourIdx]]
    _draw.draw_contours(image, [np.asarray(c) for c in sel], color, thickness)
    return image


def install():
    """Registers this facade as `cv2` when no real OpenCV is importable.  Returns the module that `import cv2` yields."""
    try:
        import cv2 as real
        return real
    except ImportError:
        sys.modules["cv2"] = sys.modules[__name__]
        return sys.modules[__name__]
