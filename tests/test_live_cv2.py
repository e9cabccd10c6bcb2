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


def test_more_conversions(oracle):
    """HSV -> BGR (both restated forms are reported), YCrCb (integer, must be exact) and HLS (float32 statement sequence: a vector build of
    OpenCV may move a handful of hues by one; the count is printed, more than 0.01 % fails)."""
    v = np.arange(1 << 24, dtype=np.uint32)
    allc = np.stack([(v & 255), (v >> 8) & 255, v >> 16], axis=1).astype(np.uint8).reshape(4096, 4096, 3)
    assert np.array_equal(oracle.bgr2ycrcb(allc), cv2.cvtColor(allc, cv2.COLOR_BGR2YCrCb))
    hls = cv2.cvtColor(allc, cv2.COLOR_BGR2HLS)
    diff = np.any(oracle.bgr2hls(allc) != hls, axis=2).sum()
    print("HLS triples that differ from this cv2 build:", int(diff))
    assert diff <= 1678
    hsv = allc[:, :, ::-1].copy()
    hsv[:, :, 0] %= 180
    ref = cv2.cvtColor(hsv, cv2.COLOR_HSV2BGR)
    d0, d1 = (np.any(oracle.hsv2bgr(hsv, 0) != ref, axis=2).sum(), np.any(oracle.hsv2bgr(hsv, 1) != ref, axis=2).sum())
    print("HSV2BGR triples that differ: vector form", int(d0), "scalar form", int(d1))
    assert min(d0, d1) == 0


def test_filters_and_warps(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    g = np.ascontiguousarray(img[:, :, 1])
    for k in (3, 5, 9, 21):
        assert np.array_equal(oracle.gaussian_blur(img, (k, k)), cv2.GaussianBlur(img, (k, k), 0))
    assert np.array_equal(oracle.gaussian_blur(g, (7, 3), 1.7, 0.6), cv2.GaussianBlur(g, (7, 3), 1.7, sigmaY=0.6))
    for bs, c in ((3, 0), (15, 2), (17, -3.5), (51, 4.2)):
        assert np.array_equal(oracle.adaptive_threshold_mean(g, 255, False, bs, c), cv2.adaptiveThreshold(g, 255, cv2.ADAPTIVE_THRESH_MEAN_C, cv2.THRESH_BINARY, bs, c))
        assert np.array_equal(oracle.adaptive_threshold_mean(g, 255, True, bs, c), cv2.adaptiveThreshold(g, 255, cv2.ADAPTIVE_THRESH_MEAN_C, cv2.THRESH_BINARY_INV, bs, c))
    for t1, t2 in ((50, 150), (10.9, 30.2), (300, 900)):
        assert np.array_equal(oracle.canny(g, t1, t2), cv2.Canny(g, t1, t2))
        assert np.array_equal(oracle.canny(img, t1, t2), cv2.Canny(img, t1, t2))
    h, w = g.shape
    # integer translations are copies in every OpenCV; general maps follow the classical fixed-point path (releases up to 4.10) —
    # later releases may differ by one grey level in places, which is reported, not failed
    assert np.array_equal(oracle.warp_affine(img, np.float32([[1, 0, 7], [0, 1, -3]]), (w, h)), cv2.warpAffine(img, np.float32([[1, 0, 7], [0, 1, -3]]), (w, h)))
    M = cv2.getRotationMatrix2D((w / 2, h / 2), 12.5, 1)
    assert np.allclose(M, oracle.rotation_matrix_2d((w / 2, h / 2), 12.5, 1), rtol=0, atol=1e-12)
    ref = cv2.warpAffine(img, M, (w, h), borderMode=cv2.BORDER_REPLICATE)
    got = oracle.warp_affine(img, M, (w, h), border="replicate")
    d = np.abs(ref.astype(int) - got.astype(int))
    print("warpAffine: pixels that differ from this cv2 build:", int((d > 0).sum()), "max", int(d.max()), "cv2", cv2.__version__)
    assert d.max() <= 1
    major, minor = (int(x) for x in cv2.__version__.split(".")[:2])
    if (major, minor) <= (4, 10):
        assert d.max() == 0

