"""Live comparison with a real OpenCV — the only thing that can ever pin the oracle to the reference's arithmetic.
Skipped wherever `cv2` is not importable (it is absent from the build and GPU images).  CPU-only: oracle vs cv2."""
import numpy as np
import pytest

import frames as F

cv2 = pytest.importorskip("cv2")
if not hasattr(cv2, "connectedComponentsWithStats") or getattr(cv2, "__name__", "") != "cv2" or "vision" in getattr(cv2, "__file__", ""):
    pytest.skip("the cv2 facade of this repo is not a reference", allow_module_level=True)


def test_colour_conversions(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (128, 160, 3), dtype=np.uint8)
    assert np.array_equal(oracle.bgr2lab(img), cv2.cvtColor(img, cv2.COLOR_BGR2LAB))
    assert np.array_equal(oracle.bgr2hsv(img), cv2.cvtColor(img, cv2.COLOR_BGR2HSV))
    assert np.array_equal(oracle.bgr2gray(img), cv2.cvtColor(img, cv2.COLOR_BGR2GRAY))
    b, g = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    for r in range(0, 256, 17):
        full = np.dstack([b, g, np.full_like(b, r)])
        assert np.array_equal(oracle.bgr2lab(full), cv2.cvtColor(full, cv2.COLOR_BGR2LAB)), r


def test_threshold_and_morphology(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (90, 130, 3), dtype=np.uint8)
    assert np.array_equal(oracle.inrange(img, (10, 20, 60), (30, 100, 255)), cv2.inRange(img, np.array([10, 20, 60]), np.array([30, 100, 255])))
    m = F.random_mask(rng, 90, 130, 0.6)
    for shape, size in ((cv2.MORPH_RECT, (5, 5)), (cv2.MORPH_ELLIPSE, (7, 7)), (cv2.MORPH_ELLIPSE, (11, 5)), (cv2.MORPH_RECT, (4, 2))):
        k = cv2.getStructuringElement(shape, size)
        assert np.array_equal(oracle.structuring_element({cv2.MORPH_RECT: 0, cv2.MORPH_ELLIPSE: 2}[shape], *size), k)
        for op, oop in ((cv2.MORPH_ERODE, oracle.ERODE), (cv2.MORPH_DILATE, oracle.DILATE), (cv2.MORPH_OPEN, oracle.OPEN),
                        (cv2.MORPH_CLOSE, oracle.CLOSE), (cv2.MORPH_GRADIENT, oracle.GRADIENT)):
            assert np.array_equal(oracle.morph(oop, m, k), cv2.morphologyEx(m, op, k))
            assert np.array_equal(oracle.morph(oop, img, k, iterations=2), cv2.morphologyEx(img, op, k, iterations=2))


def test_components_and_contours(oracle):
    rng = np.random.default_rng(2)
    for p in (0.2, 0.5, 0.7):
        m = F.random_mask(rng, 60, 90, p)
        n, lab, st, ce = cv2.connectedComponentsWithStats(m, connectivity=8, ltype=cv2.CV_32S)
        on, olab, ost, oce = oracle.ccl(m, 2)
        assert n == on and np.array_equal(lab, olab) and np.array_equal(st, ost) and np.allclose(ce, oce, equal_nan=True)
        for mode, omode in ((cv2.RETR_EXTERNAL, 0), (cv2.RETR_LIST, 1)):
            cs = cv2.findContours(m, mode, cv2.CHAIN_APPROX_SIMPLE)[0]
            exp = oracle.find_contours(m, omode, 2)
            assert len(cs) == len(exp) and all(np.array_equal(a, b) for a, b in zip(cs, exp))
