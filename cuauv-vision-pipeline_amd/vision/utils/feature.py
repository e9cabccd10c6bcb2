"""Mirror of the reference utils/feature.py for the hot path.

`connected_components` is the north-star replacement of the cv2.findContours stage
(utils/feature.py:5-40, modules/red_buoy.py:38): cv2.connectedComponentsWithStats(mask, 8, CV_32S)
semantics on the GPU (libvp vp_ccl_u8).  Polygon helpers (`contour_centroid`, `contour_area`,
utils/feature.py:240-265) are float64 scalar arithmetic on a handful of points and stay on the host.
"""
from typing import List, Tuple

import threading

import numpy as np
from collections.abc import Sequence as _SequenceABC

from vision import _vp
from vision.devmat import DeviceMat, to_host_readonly
from vision.utils.helpers import as_mat, device_image


def connected_components(mat: np.ndarray, numbering: int = _vp.CCL_BLOCK2X2, max_labels: int = 4096,
                         want_labels: bool = True):
    """Returns (nlabels, labels int32 (h,w) or None, stats int32 (k,5), centroids float64 (k,2)),
    k = min(nlabels, max_labels); row 0 is the background, like cv2."""
    mat = to_host_readonly(as_mat(mat))
    if not isinstance(mat, np.ndarray) or mat.dtype != np.uint8:
        raise TypeError("expected a uint8 mask")
    if mat.ndim == 3 and mat.shape[2] == 1:
        mat = mat[:, :, 0]
    if mat.ndim != 2 or mat.size == 0:
        raise ValueError("expected a non-empty (h, w) mask")
    if mat.strides[1] != 1:
        mat = np.ascontiguousarray(mat)
    h, w = mat.shape
    labels = np.empty((h, w), np.int32) if want_labels else None
    stats = np.empty((max_labels, 5), np.int32)
    cent = np.empty((max_labels, 2), np.float64)
    n = _vp.C.c_int32(0)
    ctx = _vp.default_context()
    _vp.check(_vp.lib().vp_ccl_u8(ctx.handle, _vp.ptr(mat), mat.strides[0], w, h, int(numbering), _vp.ptr(labels), _vp.ptr(stats),
                                  _vp.ptr(cent), int(max_labels), _vp.C.byref(n)), ctx.handle)
    k = min(n.value, max_labels)
    return n.value, labels, stats[:k].copy(), cent[:k].copy()


_sums_tls = threading.local()


def _green_sums(pts):
    """(a00, a10, a01) of an integer contour as Python ints: one native pass (libvp vp_polygon_sums_i32, host code)."""
    p32 = pts if (pts.dtype == np.int32 and pts.flags.c_contiguous) else np.ascontiguousarray(pts, np.int32)
    t = _sums_tls
    try:
        out, fn = t.out, t.fn
    except AttributeError:                             # per thread: the result buffer and the resolved entry point
        out = t.out = (_vp.C.c_int64 * 3)()
        fn = t.fn = _vp.lib().vp_polygon_sums_i32
    if fn(p32.__array_interface__["data"][0], len(p32), out) != 0:
        raise _vp.VpError("vp_polygon_sums_i32: invalid argument")
    return out[0], out[1], out[2]


def _polygon_moments(contour: np.ndarray):
    """cv2.moments on an (N,1,2) int contour (imgproc/src/moments.cpp contourMoments): Green's theorem,
    float64; m00 made non-negative together with the first moments.  The sums run over integers (coordinates below 2^15, so every
    partial sum stays below 2^53): summed in int64 here, they equal the sequential float64 loop of the C++ code bit for bit."""
    pts = np.asarray(contour).reshape(-1, 2)
    n = len(pts)
    if n == 0:
        return 0.0, 0.0, 0.0
    if not np.issubdtype(pts.dtype, np.integer):
        return _polygon_moments_float(pts.astype(np.float64))
    a00, a10, a01 = (float(v) for v in _green_sums(pts))
    if abs(a00) > 1.1920929e-07:
        if a00 > 0:
            db1_2, db1_6 = 0.5, 1.0 / 6
        else:
            db1_2, db1_6 = -0.5, -1.0 / 6
        return a00 * db1_2, a10 * db1_6, a01 * db1_6
    return 0.0, 0.0, 0.0


def _polygon_moments_float(pts):
    n = len(pts)
    a00 = a10 = a01 = 0.0
    xi_1, yi_1 = pts[n - 1]
    for i in range(n):
        xi, yi = pts[i]
        dxy = xi_1 * yi - xi * yi_1
        a00 += dxy
        a10 += dxy * (xi_1 + xi)
        a01 += dxy * (yi_1 + yi)
        xi_1, yi_1 = xi, yi
    if abs(a00) > 1.1920929e-07:
        if a00 > 0:
            db1_2, db1_6 = 0.5, 1.0 / 6
        else:
            db1_2, db1_6 = -0.5, -1.0 / 6
        return a00 * db1_2, a10 * db1_6, a01 * db1_6
    return 0.0, 0.0, 0.0


def contour_centroid(contour: np.ndarray) -> Tuple[int, int]:
    """utils/feature.py:240-252."""
    m00, m10, m01 = _polygon_moments(contour)
    m00 = max(1e-10, m00)
    return int(m10 / m00), int(m01 / m00)


def contour_area(contour: np.ndarray) -> float:
    """utils/feature.py:255-265 (cv2.contourArea, oriented=False): |shoelace| / 2."""
    if type(contour) is np.ndarray and contour.dtype == np.int32 and contour.size and contour.flags.c_contiguous:   # what find_contours hands out
        return abs(float(_green_sums(contour.reshape(-1, 2))[0]) * 0.5)
    pts = np.asarray(contour).reshape(-1, 2)
    n = len(pts)
    if n == 0:
        return 0.0
    if np.issubdtype(pts.dtype, np.integer):         # integer sums: exact, equal to the sequential float64 loop (see _polygon_moments)
        return abs(float(_green_sums(pts)[0]) * 0.5)
    pts = pts.astype(np.float64)
    a00 = 0.0
    prev = pts[n - 1]
    for i in range(n):
        p = pts[i]
        a00 += prev[0] * p[1] - prev[1] * p[0]
        prev = p
    return abs(a00 * 0.5)


def contour_perimeter(contour: np.ndarray, closed: bool = True) -> float:
    """utils/feature.py:266-278 (cv2.arcLength): host arithmetic on the few points of one contour."""
    from vision import cv2_facade
    return cv2_facade.arcLength(contour, closed)


def contour_approx(contour: np.ndarray, epsilon: float = None, closed: bool = True) -> np.ndarray:
    """utils/feature.py:281-296 (cv2.approxPolyDP; epsilon defaults to a tenth of the perimeter)."""
    from vision import cv2_facade
    if epsilon is None:
        epsilon = 0.1 * contour_perimeter(contour, closed)
    return cv2_facade.approxPolyDP(contour, epsilon, closed)


def min_enclosing_rect(contour: np.ndarray):
    """utils/feature.py:299-310 (cv2.minAreaRect): ((cx, cy), (w, h), angle in degrees), OpenCV >= 4.5.1 angle convention."""
    from vision import cv2_facade
    return cv2_facade.minAreaRect(contour)


def _outside_path(name):
    def _f(*_a, **_k):
        raise NotImplementedError(f"{name}: outside the accelerated path of this build")
    _f.__name__ = name
    return _f


min_enclosing_circle = _outside_path("min_enclosing_circle")
min_enclosing_ellipse = _outside_path("min_enclosing_ellipse")


def canny(mat: np.ndarray, lower: int, upper: int) -> np.ndarray:
    """utils/feature.py:43-66 (cv2.Canny with the default 3x3 aperture and L1 gradient) on the GPU (libvp vp_canny_u8)."""
    from vision import cv2_facade
    return cv2_facade.Canny(mat, lower, upper)


def simple_canny(mat: np.ndarray, sigma: float = 0.33, use_mean: bool = False) -> np.ndarray:
    """utils/feature.py:69-101: thresholds (1 -/+ sigma) x median (or mean) of the image, truncated to int and clamped to [0, 255]."""
    mid = np.mean(mat) if use_mean else np.median(mat)
    lower = int(max(0, (1.0 - sigma) * mid))
    upper = int(min(255, (1.0 + sigma) * mid))
    return canny(mat, lower, upper)
find_corners = _outside_path("find_corners")
find_circles = _outside_path("find_circles")
find_lines = _outside_path("find_lines")
find_line_segments = _outside_path("find_line_segments")


def find_contours(mat: np.ndarray, mode: int = _vp.RETR_EXTERNAL, method: int = _vp.CHAIN_APPROX_SIMPLE, with_holes: bool = False):
    """cv2.findContours(mat, mode, method)[0] on the GPU (libvp vp_find_contours_u8): tuple of (N,1,2) int32 arrays of
    (x, y) points, newest contour first like cv2."""
    mat = as_mat(mat)
    ctx = _vp.default_context()
    dev = None
    if isinstance(mat, DeviceMat):                     # the mask is already in HBM (range_threshold / morphology result): read it there
        if mat.dtype != np.uint8:
            raise TypeError("expected a uint8 mask")
        dev = device_image(ctx, mat, 1)
        h, w = dev.shape
    else:
        if not isinstance(mat, np.ndarray) or mat.dtype != np.uint8:
            raise TypeError("expected a uint8 mask")
        if mat.ndim == 3 and mat.shape[2] == 1:
            mat = mat[:, :, 0]
        if mat.ndim != 2 or mat.size == 0:
            raise ValueError("expected a non-empty (h, w) mask")
        if mat.strides[1] != 1:
            mat = np.ascontiguousarray(mat)
        h, w = mat.shape
    # capacities that served the last call of this size: a mask with more contours than the first guess (speckle: modules/red_buoy.py:38
    # runs on the un-cleaned mask) would otherwise be traced twice at every call, once to learn the sizes and once to keep the result
    max_c, max_p, ttl = _capacity.get((h, w), (256, 1 << 14, 0))
    while True:
        # result arrays of the C call: per thread, kept from call to call (what is handed out below are copies of the used parts)
        sc = getattr(_scratch, "arrays", None)
        if sc is None or sc[0] != (max_c, max_p):
            pts = np.empty((max_p, 2), np.int32)
            counts = np.empty(max_c, np.int32)
            holes = np.empty(max_c, np.uint8)
            sc = _scratch.arrays = ((max_c, max_p), pts, counts, holes)
        _, pts, counts, holes = sc
        nc, npts = _vp.C.c_int32(0), _vp.C.c_int64(0)
        bits = getattr(dev, "_bits", None) if dev is not None else None
        if bits is not None and dev._dev_ok and dev._ctx is ctx:
            # the mask came with its bit plane (range_threshold) and has not been written to since: no packing launch
            _vp.check(_vp.lib().vp_find_contours_bits_dev(ctx.handle, bits.ptr, w, h, int(mode), int(method), _vp.ptr(pts), max_p,
                                                          _vp.ptr(counts), _vp.ptr(holes), max_c, _vp.C.byref(nc), _vp.C.byref(npts)), ctx.handle)
        elif dev is not None:
            _vp.check(_vp.lib().vp_find_contours_dev(ctx.handle, dev.dev_ptr, w, w, h, int(mode), int(method), _vp.ptr(pts), max_p,
                                                     _vp.ptr(counts), _vp.ptr(holes), max_c, _vp.C.byref(nc), _vp.C.byref(npts)), ctx.handle)
        else:
            _vp.check(_vp.lib().vp_find_contours_u8(ctx.handle, _vp.ptr(mat), mat.strides[0], w, h, int(mode), int(method), _vp.ptr(pts), max_p,
                                                    _vp.ptr(counts), _vp.ptr(holes), max_c, _vp.C.byref(nc), _vp.C.byref(npts)), ctx.handle)
        if nc.value <= max_c and npts.value <= max_p:
            break
        max_c = max(max_c, 2 * nc.value)
        max_p = max(max_p, 2 * npts.value)
    k = nc.value
    if k > 256 or npts.value > (1 << 14):
        _capacity[(h, w)] = (max(256, k + k // 2), max(1 << 14, npts.value + npts.value // 2), 32)     # follows the masks up and down ...
    elif ttl > 1:
        _capacity[(h, w)] = (max_c, max_p, ttl - 1)    # ... down only after 32 small results in a row: speckle that comes and goes is not traced twice each time
    else:
        _capacity.pop((h, w), None)
    flat = pts[:npts.value].copy()                     # one block for all contours; the arrays handed out are views of it
    if k > LazyContourList.THRESHOLD:
        out = LazyContourList(flat, counts[:k].copy())
    else:
        cnt = counts[:k].tolist()
        pts3 = flat.reshape(-1, 1, 2)
        out, o = [], 0
        for c in cnt:
            out.append(pts3[o:o + c])
            o += c
        out = ContourList(out)
        out._flat, out._counts = flat, counts[:k].copy()
    return (out, holes[:k].copy()) if with_holes else out


_capacity = {}            # (h, w) -> (contours, points, calls left before shrinking) the result arrays of find_contours start with
_scratch = threading.local()


class ContourList(tuple):
    """The tuple of (N, 1, 2) int32 arrays cv2.findContours returns.  The arrays are views of one point block (`_flat`, in the
    tuple's order), which `draw_contours` hands to the rasteriser as it is; writing to a contour writes to the block, so the two
    cannot disagree."""


class LazyContourList(_SequenceABC):
    """What find_contours returns for a mask with many contours (raw speckle: 17 k at 2 % noise): the same sequence of (N, 1, 2) views
    of one point block, made when they are looked at.  Building seventeen thousand array views up front took 2.2 ms of a 2.5 ms call;
    len(), indexing, slicing, iteration, `max(contours, key=...)` and the overlay (which reads the block itself) behave as on the tuple."""
    THRESHOLD = 512
    __slots__ = ("_flat", "_counts", "_ends", "_pts3")

    def __init__(self, flat, counts):
        self._flat, self._counts = flat, counts
        self._ends = np.cumsum(counts, dtype=np.int64)
        self._pts3 = flat.reshape(-1, 1, 2)

    def __len__(self):
        return len(self._counts)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return tuple(self[j] for j in range(*i.indices(len(self._counts))))
        n = len(self._counts)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("contour index out of range")
        e = int(self._ends[i])
        return self._pts3[e - int(self._counts[i]):e]

    def __iter__(self):
        pts3, o = self._pts3, 0
        for c in self._counts.tolist():
            yield pts3[o:o + c]
            o += c

    def __add__(self, other):
        return tuple(self) + tuple(other)

    def __radd__(self, other):
        return tuple(other) + tuple(self)

    def __repr__(self):
        return f"<{len(self)} contours, {len(self._flat)} points>"


def outer_contours(mat: np.ndarray) -> List[np.ndarray]:
    """utils/feature.py:5-21 (cv2.findContours RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)."""
    return find_contours(mat, _vp.RETR_EXTERNAL, _vp.CHAIN_APPROX_SIMPLE)


def all_contours(mat: np.ndarray) -> List[np.ndarray]:
    """utils/feature.py:25-40 (cv2.findContours RETR_LIST, CHAIN_APPROX_SIMPLE)."""
    return find_contours(mat, _vp.RETR_LIST, _vp.CHAIN_APPROX_SIMPLE)
