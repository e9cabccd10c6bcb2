"""Host-side geometry of the cv2 stand-in (modules/bins.py:60-72 calls cv2.minAreaRect / cv2.boxPoints on every contour): the native
convex hull (libvp vp_convex_hull_i32, exact integers) against the pure-Python monotone chain, and the all-edges-at-once rotating
calipers against the same statements edge by edge.  No GPU involved."""
import numpy as np
import pytest

from vision import cv2_facade as cv2


def _shapes(rng):
    yield np.array([[0, 0]], np.int32)
    yield np.array([[3, 4], [3, 4], [3, 4]], np.int32)
    yield np.array([[0, 0], [10, 0]], np.int32)
    yield np.array([[0, 0], [5, 5], [10, 10], [2, 2]], np.int32)                       # collinear
    yield np.array([[10, 10], [10, 30], [50, 30], [50, 10]], np.int32)                  # axis-aligned box
    yield np.array([[0, 0], [4, 0], [4, 4], [0, 4], [2, 2], [1, 3]], np.int32)         # interior points
    for _ in range(40):
        n = int(rng.integers(3, 400))
        yield rng.integers(-50, 2000, (n, 2)).astype(np.int32)
    for _ in range(20):                                                                 # rotated rectangles as staircase contours
        cx, cy, w, h, t = rng.uniform(100, 900), rng.uniform(100, 500), rng.uniform(5, 300), rng.uniform(5, 150), rng.uniform(0, np.pi)
        u = np.array([np.cos(t), np.sin(t)]); v = np.array([-np.sin(t), np.cos(t)])
        s = np.linspace(-1, 1, 60)
        edge = np.concatenate([np.outer(s, u * w / 2) + v * h / 2, np.outer(s, v * h / 2) + u * w / 2, np.outer(-s, u * w / 2) - v * h / 2, np.outer(-s, v * h / 2) - u * w / 2])
        yield np.rint(edge + (cx, cy)).astype(np.int32)


def test_native_hull_equals_python_hull():
    rng = np.random.default_rng(1)
    for pts in _shapes(rng):
        got = cv2._convex_hull(pts.reshape(-1, 1, 2))
        exp = cv2._convex_hull_py(pts.astype(np.float64))
        assert got.dtype == np.float64 and got.shape == exp.shape and np.array_equal(got, exp), pts[:5]
    assert len(cv2._convex_hull(np.zeros((0, 1, 2), np.int32))) == 0
    f = rng.uniform(0, 100, (30, 2))                                                   # float points keep the Python path
    assert np.array_equal(cv2._convex_hull(f), cv2._convex_hull_py(f))


def test_min_area_rect_equals_the_edge_by_edge_form():
    """Integer points (contours) take libvp's host routine, anything else the all-edges-at-once numpy form: both give exactly what the
    statements give edge by edge."""
    rng = np.random.default_rng(2)
    for pts in _shapes(rng):
        exp = cv2._min_area_rect_loop(pts.reshape(-1, 1, 2))
        assert cv2.minAreaRect(pts.reshape(-1, 1, 2)) == exp, pts[:5]                       # native
        assert cv2.minAreaRect(pts.astype(np.int64)) == exp                                 # native after conversion
        assert cv2.minAreaRect(pts.astype(np.float64).reshape(-1, 1, 2)) == exp, pts[:5]    # numpy form on the same points
    for _ in range(20):                                                                    # points that are not integers
        f = rng.uniform(-50, 900, (int(rng.integers(1, 60)), 2))
        assert cv2.minAreaRect(f) == cv2._min_area_rect_loop(f)
    assert cv2.minAreaRect(np.zeros((0, 1, 2), np.int32)) == ((0.0, 0.0), (0.0, 0.0), 0.0)
    (cx, cy), (w, h), ang = cv2.minAreaRect(np.array([[10, 10], [10, 30], [50, 30], [50, 10]], np.int32))
    assert (cx, cy) == (30.0, 20.0) and sorted((w, h)) == [20.0, 40.0] and ang in (90.0,)
    box = cv2.boxPoints(((30.0, 20.0), (w, h), ang))
    assert sorted(map(tuple, np.rint(box).astype(int).tolist())) == [(10, 10), (10, 30), (50, 10), (50, 30)]
