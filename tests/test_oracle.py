"""CPU suite: pins the oracle (oracle/vp_oracle.c) against the published OpenCV known answers of
SURVEY Appendix A (tests/golden/known_answers.json), against SciPy as an independent witness and
against float64 analytic colour formulas.  The reference itself has no tests / fixtures."""
import json
import os

import numpy as np
import pytest
import scipy.ndimage as ndi

import frames as F

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")))


def test_known_answers_lab_hsv_gray(oracle):
    for (bgr, exp) in G["bgr2lab"]:
        assert oracle.bgr2lab(np.array([[bgr]], np.uint8))[0, 0].tolist() == exp
    for (bgr, exp) in G["bgr2hsv"]:
        assert oracle.bgr2hsv(np.array([[bgr]], np.uint8))[0, 0].tolist() == exp
    for (bgr, exp) in G["bgr2gray"]:
        assert int(oracle.bgr2gray(np.array([[bgr]], np.uint8))[0, 0]) == exp
    assert oracle.tables()[4].tolist() == G["lab_coeffs"]


def test_bench_colours_separate(oracle):
    # background (150,110,40) -> a=120, buoy (40,45,210) -> a=190: threshold [150,255] splits them
    lab = oracle.bgr2lab(np.array([[[150, 110, 40], [40, 45, 210]]], np.uint8))
    assert lab[0, 0].tolist() == [112, 120, 100] and lab[0, 1].tolist() == [119, 190, 172]


def test_lab_tables_variants(oracle):
    """softfloat-faithful tables (shipped) vs plain float64 tables: gamma identical; the cube-root
    table differs in a handful of entries only — those are the entries a live cv2 must settle."""
    g0, c0, *_ = oracle.tables(0)
    g1, c1, *_ = oracle.tables(1)
    assert np.array_equal(g0, g1)
    diff = np.nonzero(c0[:2041] != c1[:2041])[0]
    assert len(diff) <= 8 and np.all(np.abs(c0.astype(int) - c1.astype(int)) <= 1)
    assert g0[0] == 0 and g0[255] == 2040 and c0[2040] == 32768


def _all_colours(chunk=1 << 20):
    for c0 in range(0, 1 << 24, chunk):
        v = np.arange(c0, c0 + chunk, dtype=np.uint32)
        yield np.stack([v & 255, (v >> 8) & 255, v >> 16], axis=1).astype(np.uint8).reshape(1024, -1, 3)


def test_conversions_against_analytic_float64_all_colours(oracle):
    """Every one of the 2^24 BGR triples against the textbook formulas in float64 (sRGB gamma, D65 XYZ, CIE L*a*b* scaled to 8 bits;
    hexcone HSV with H in half degrees).  HSV: the 8-bit result is the analytic value rounded (never further than 0.65 away).
    LAB: the 8-bit path quantises linear light to 1/2040 and the cube root to a table, which costs up to 2.7 levels in a / b for very
    dark pixels (true of cv2 as well); 99.86 % of all values are within 1, the mean distance is 0.27."""
    M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    worst, within1, total, summed, s_worst, h_worst = 0.0, 0, 0, 0.0, 0.0, 0.0
    for bgr in _all_colours():
        lab = oracle.bgr2lab(bgr).astype(np.float64)
        rgb = bgr[:, :, ::-1].astype(np.float64) / 255
        lin = np.where(rgb <= 0.04045, rgb / 12.92, ((rgb + 0.055) / 1.055) ** 2.4)
        xyz = lin @ M.T / np.array([0.950456, 1.0, 1.088754])
        f = np.where(xyz > 216 / 24389, np.cbrt(xyz), 841 / 108 * xyz + 16 / 116)
        ref = np.stack([(116 * f[..., 1] - 16) * 255 / 100, 500 * (f[..., 0] - f[..., 1]) + 128, 200 * (f[..., 1] - f[..., 2]) + 128], -1)
        err = np.abs(lab - ref)
        worst, within1, total, summed = max(worst, err.max()), within1 + int((err <= 1.0).sum()), total + err.size, summed + err.sum()
        hsv = oracle.bgr2hsv(bgr).astype(np.float64)
        b, g, r = [bgr[..., i].astype(np.float64) for i in range(3)]
        v = np.maximum(np.maximum(b, g), r)
        d = v - np.minimum(np.minimum(b, g), r)
        sat = np.where(v > 0, 255 * d / np.maximum(v, 1), 0)
        dd = np.maximum(d, 1)
        h = np.where(v == r, (g - b) / dd, np.where(v == g, 2 + (b - r) / dd, 4 + (r - g) / dd)) * 30
        h = np.where(d == 0, 0, np.where(h < 0, h + 180, h))
        assert np.array_equal(hsv[..., 2], v)
        assert np.abs(oracle.bgr2gray(bgr) - (0.114 * b + 0.587 * g + 0.299 * r)).max() <= 0.51      # grey: the weighted sum, rounded
        s_worst = max(s_worst, np.abs(hsv[..., 1] - sat).max())
        dh = np.abs(hsv[..., 0] - h)
        h_worst = max(h_worst, np.minimum(dh, 180 - dh).max())
        # YCrCb (Y as grey; Cr / Cb from the rounded Y, hence up to 0.9 away) and HLS (the same hue; lightness and saturation rounded)
        ycc = oracle.bgr2ycrcb(bgr).astype(np.float64)
        yy = 0.299 * r + 0.587 * g + 0.114 * b
        assert np.abs(ycc[..., 0] - yy).max() <= 0.51
        assert np.abs(ycc[..., 1] - np.clip((r - yy) * 0.713 + 128, 0, 255)).max() <= 0.9 and np.abs(ycc[..., 2] - np.clip((b - yy) * 0.564 + 128, 0, 255)).max() <= 0.9
        hls = oracle.bgr2hls(bgr).astype(np.float64)
        hi, lo = v / 255, (v - d) / 255
        sat_l = np.where(d > 0, np.where(hi + lo < 1, (hi - lo) / np.maximum(hi + lo, 1e-12), (hi - lo) / np.maximum(2 - hi - lo, 1e-12)), 0)
        assert np.abs(hls[..., 1] - (hi + lo) / 2 * 255).max() <= 0.501 and np.abs(hls[..., 2] - sat_l * 255).max() <= 0.501
        dh = np.abs(hls[..., 0] - h)
        assert np.minimum(dh, 180 - dh).max() <= 0.7
    assert worst <= 3.0 and within1 / total > 0.998 and summed / total < 0.3, (worst, within1 / total, summed / total)
    assert s_worst <= 0.6 and h_worst <= 0.7, (s_worst, h_worst)


def test_inrange_known(oracle):
    k = G["inrange"]
    v = np.array([k["values"]], np.uint8)
    assert oracle.inrange(v, k["lo"], k["hi"])[0].tolist() == k["expect"]
    assert oracle.inrange(v, 200, 100).max() == 0 and oracle.inrange(v, -10, 300).min() == 255
    assert oracle.inrange(v, 256, 300).max() == 0 and oracle.inrange(v, -10, -1).max() == 0


def test_structuring_elements_known(oracle):
    for name, k in (("ellipse3", 3), ("ellipse5", 5), ("ellipse7", 7)):
        exp = np.array([[int(c) for c in row] for row in G[name]], np.uint8)
        assert np.array_equal(oracle.structuring_element(oracle.MORPH_ELLIPSE, k, k), exp)
    assert np.array_equal(oracle.structuring_element(oracle.MORPH_RECT, 4, 2), np.ones((2, 4), np.uint8))
    cross = oracle.structuring_element(oracle.MORPH_CROSS, 3, 3)
    assert cross.tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]


def test_morph_by_definition(oracle):
    m = np.zeros(G["morph"]["size"], np.uint8)
    y0, x0, hh, ww = G["morph"]["block"]
    m[y0:y0 + hh, x0:x0 + ww] = 255
    k = np.ones((3, 3), np.uint8)
    e = oracle.morph(oracle.ERODE, m, k)
    assert e.sum() == 255 and e[3, 3] == 255
    d = oracle.morph(oracle.DILATE, m, k)
    assert d.sum() == 255 * 25 and d[1:6, 1:6].min() == 255
    full = np.full((5, 9), 255, np.uint8)
    assert np.array_equal(oracle.morph(oracle.ERODE, full, np.ones((5, 5), np.uint8)), full)  # border never wins
    assert oracle.morph(oracle.GRADIENT, full, k).max() == 0


@pytest.mark.parametrize("seed", range(6))
def test_morph_against_scipy(oracle, seed):
    """rect kernels on 0/255 masks: scipy binary_erosion(border_value=1) / binary_dilation(border_value=0)
    implement the same 'outside never wins' rule."""
    rng = np.random.default_rng(seed)
    h, w = rng.integers(5, 60, 2)
    m = F.random_mask(rng, h, w)
    for ky, kx in [(3, 3), (5, 5), (1, 5), (7, 3)]:
        k = np.ones((ky, kx), np.uint8)
        e = ndi.binary_erosion(m > 0, structure=k, border_value=1)
        d = ndi.binary_dilation(m > 0, structure=k, border_value=0)
        assert np.array_equal(oracle.morph(oracle.ERODE, m, k) > 0, e)
        assert np.array_equal(oracle.morph(oracle.DILATE, m, k) > 0, d)
        assert np.array_equal(oracle.morph(oracle.ERODE, m, k, fast=True), oracle.morph(oracle.ERODE, m, k))
        assert np.array_equal(oracle.morph(oracle.CLOSE, m, k, fast=True), oracle.morph(oracle.CLOSE, m, k))
    g = rng.integers(0, 256, (h, w), dtype=np.uint8)
    el = oracle.structuring_element(oracle.MORPH_ELLIPSE, 5, 5)
    assert np.array_equal(oracle.morph(oracle.ERODE, g, el), ndi.grey_erosion(g, footprint=el, mode="constant", cval=255))
    assert np.array_equal(oracle.morph(oracle.DILATE, g, el), ndi.grey_dilation(g, footprint=el, mode="constant", cval=0))


def test_morph_iterations_and_duality(oracle):
    rng = np.random.default_rng(7)
    m = F.random_mask(rng, 31, 45, 0.6)
    k3 = np.ones((3, 3), np.uint8)
    assert np.array_equal(oracle.morph(oracle.ERODE, m, k3, iterations=2), oracle.morph(oracle.ERODE, m, np.ones((5, 5), np.uint8)))
    two = oracle.morph(oracle.ERODE, oracle.morph(oracle.ERODE, m, k3), k3)
    assert np.array_equal(oracle.morph(oracle.ERODE, m, k3, iterations=2), two)
    assert np.array_equal(255 - oracle.morph(oracle.ERODE, m, k3), oracle.morph(oracle.DILATE, 255 - m, k3))
    o = oracle.morph(oracle.OPEN, m, k3)
    assert np.array_equal(oracle.morph(oracle.OPEN, o, k3), o)  # idempotent
    assert np.array_equal(oracle.morph(oracle.ERODE, m, None), oracle.morph(oracle.ERODE, m, k3))
    assert np.array_equal(oracle.morph(oracle.ERODE, m, k3, iterations=0), m)


@pytest.mark.parametrize("seed", range(8))
def test_ccl_partition_against_scipy(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    h, w = rng.integers(1, 50, 2)
    m = F.random_mask(rng, h, w)
    ref, nref = ndi.label(m, structure=np.ones((3, 3)))
    for block in (1, 2):
        n, lab, st, ce = oracle.ccl(m, block)
        assert n == nref + 1
        assert len(set(zip(lab.ravel().tolist(), ref.ravel().tolist()))) == n  # same partition
        for l in range(1, n):
            ys, xs = np.nonzero(lab == l)
            assert st[l].tolist() == [xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1, len(xs)]
            assert ce[l, 0] == xs.sum() / len(xs) and ce[l, 1] == ys.sum() / len(ys)


def test_ccl_numbering_known(oracle):
    k = G["ccl_numbering"]
    m = np.zeros(k["size"], np.uint8)
    m[tuple(k["A"])] = 255
    m[tuple(k["B"])] = 255
    for block, key in ((2, "block2x2"), (1, "pixel")):
        n, lab, _, _ = oracle.ccl(m, block)
        assert n == 3 and lab[tuple(k["A"])] == k[key]["A"] and lab[tuple(k["B"])] == k[key]["B"]


def test_ccl_background_row_conventions(oracle):
    n, lab, st, ce = oracle.ccl(np.full((3, 3), 255, np.uint8), 2)
    assert n == 2 and st[1].tolist() == [0, 0, 3, 3, 9] and st[0, 4] == 0 and np.isnan(ce[0]).all()
    n, lab, st, ce = oracle.ccl(np.zeros((3, 4), np.uint8), 2)
    assert n == 1 and st[0].tolist() == [0, 0, 4, 3, 12] and ce[0].tolist() == [1.5, 1.0]


def test_chain_equals_composition(oracle):
    f = F.s1_buoy(0, 128, 72)
    out = oracle.chain(f, oracle.MODE_LAB, (0, 150, 0), (255, 255, 255), [oracle.OPEN, oracle.CLOSE], 5, 5, 2, 512)
    th = oracle.inrange(np.ascontiguousarray(oracle.bgr2lab(f)[:, :, 1]), 150, 255)
    k = np.ones((5, 5), np.uint8)
    cl = oracle.morph(oracle.CLOSE, oracle.morph(oracle.OPEN, th, k), k)
    assert np.array_equal(out["threshed"], th) and np.array_equal(out["cleaned"], cl)
    n, lab, st, ce = oracle.ccl(cl, 2)
    assert out["nlabels"] == n and np.array_equal(out["labels"], lab) and np.array_equal(out["stats"], st)
    assert n >= 2


def test_known_answers_second_batch(oracle):
    """tests/golden/known_answers.json entries added with the colour balance / blur / contour work."""
    for hsv, bgr in G["hsv2bgr"]:
        assert oracle.hsv2bgr(np.array([[hsv]], np.uint8))[0, 0].tolist() == bgr
    for n, taps in G["gaussian_taps"].items():
        assert oracle.gaussian_kernel_fixed(int(n)).tolist() == taps
    c = G["contour_rect"]
    m = np.zeros(c["size"], np.uint8)
    y0, x0, y1, x1 = c["rect"]
    m[y0:y1, x0:x1] = 255
    (cont,) = oracle.find_contours(m, 0, 2)
    assert cont.reshape(-1, 2).tolist() == c["points"]


def test_hsv2bgr_known_answers_and_variants(oracle):
    """cv2.cvtColor(COLOR_HSV2BGR), 8-bit: primaries / greys (the inverses of SURVEY A2's known answers), agreement of the two
    arithmetic forms of OpenCV's float kernel except on a small counted set, round trip through BGR2HSV within the
    quantisation of H (2 degrees) and S."""
    px = np.array([[[0, 255, 255], [60, 255, 255], [120, 255, 255], [0, 0, 255], [0, 0, 0], [15, 76, 200], [30, 255, 255], [90, 255, 128]]], np.uint8)
    exp = [[0, 0, 255], [0, 255, 0], [255, 0, 0], [255, 255, 255], [0, 0, 0], [140, 170, 200], [0, 255, 255], [128, 128, 0]]
    for variant in (0, 1):
        assert oracle.hsv2bgr(px, variant).reshape(-1, 3).tolist() == exp
    h, s, v = np.meshgrid(np.arange(180), np.arange(0, 256, 3), np.arange(256), indexing="ij")
    hsv = np.stack([h, s, v], -1).astype(np.uint8).reshape(180, -1, 3)
    a, b = oracle.hsv2bgr(hsv, 0), oracle.hsv2bgr(hsv, 1)
    d = np.abs(a.astype(int) - b.astype(int)).max(-1)
    assert d.max() <= 1 and (d > 0).mean() < 2e-4
    rng = np.random.default_rng(0)
    bgr = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    rt = oracle.hsv2bgr(oracle.bgr2hsv(bgr))
    assert np.abs(rt.astype(int) - bgr.astype(int)).max() <= 6


def test_color_balance_restatement(oracle):
    """utils/color_correction/color_balance.cpp process_frame: hand-checkable cases of the restatement."""
    f = F.s1_buoy(0, 160, 90)
    # every stage off: identity
    assert np.array_equal(oracle.color_balance(f, equalize_rgb=False, hsv_contrast_correct=False, rgb_extrema_clipping=False), f)
    # extrema clipping only: each channel clipped to its 0.2 % / 99.8 % histogram bounds (cpp:111-139)
    out = oracle.color_balance(f, equalize_rgb=False, hsv_contrast_correct=False)
    n = f.shape[0] * f.shape[1]
    for c in range(3):
        ch = np.sort(f[:, :, c].ravel())
        lo = ch[int(np.float32(0.002) * np.float32(n))]
        hi = ch[n - 1 - (n - int(np.float32(0.998) * np.float32(n)))]
        assert out[:, :, c].min() == lo and out[:, :, c].max() == hi
        assert np.array_equal(out[:, :, c], np.clip(f[:, :, c], lo, hi))
    # equalisation only, blue cast (B mean largest): R and G are lifted by mean(B)/mean(R), mean(B)/mean(G), truncated; B untouched
    out = oracle.color_balance(f, hsv_contrast_correct=False, rgb_extrema_clipping=False, mean_mode=1)
    m = f.reshape(-1, 3).astype(np.float64).mean(0)
    assert m[0] > m[1] and m[0] > m[2]
    assert np.array_equal(out[:, :, 0], f[:, :, 0])
    for c in (1, 2):
        assert np.array_equal(out[:, :, c], np.minimum(f[:, :, c].astype(np.float64) * (m[0] / m[c]), 255).astype(np.uint8))
    # the running mean of the reference and the exact mean give the same frame here
    assert np.array_equal(out, oracle.color_balance(f, hsv_contrast_correct=False, rgb_extrema_clipping=False, mean_mode=0))
    # HSV stretch: afterwards S and V of the result span the full range (up to the conversion's quantisation)
    out = oracle.color_balance(f)
    hsv = oracle.bgr2hsv(out)
    assert hsv[:, :, 2].max() >= 250 and hsv[:, :, 2].min() <= 5
    # HSI stage alone: hue is kept (grey pixels stay grey), intensity spans the range afterwards
    g = np.repeat(np.arange(0, 250, 1, dtype=np.uint8)[None, :, None], 40, 0).repeat(3, 2)      # a grey ramp
    out = oracle.color_balance(g, equalize_rgb=False, hsv_contrast_correct=False, rgb_extrema_clipping=False, hsi_contrast_correct=True)
    assert (out[:, :, 0] == out[:, :, 1]).all() and (out[:, :, 1] == out[:, :, 2]).all()
    assert out.min() == 0 and out.max() >= 254 and (np.diff(out[0, :, 0].astype(int)) >= 0).all()


def test_gaussian_blur_restatement(oracle):
    """cv2.GaussianBlur, 8-bit fixed-point path: the published small kernels (1-2-1, 1-4-6-4-1, ...) in 8.8 fixed point, taps that
    always sum to 256, a constant image stays constant, an impulse reproduces the outer product of the taps, and the 3x3 case is
    the binomial average with half-up rounding and mirrored (101) borders."""
    assert oracle.gaussian_kernel_fixed(3).tolist() == [64, 128, 64]
    assert oracle.gaussian_kernel_fixed(5).tolist() == [16, 64, 96, 64, 16]
    assert oracle.gaussian_kernel_fixed(7).tolist() == [8, 28, 56, 72, 56, 28, 8]
    assert oracle.gaussian_kernel_fixed(9).tolist() == [4, 13, 30, 51, 60, 51, 30, 13, 4]
    for n in range(1, 202, 2):
        for sigma in (0.0, 0.5, 3.7):
            t = oracle.gaussian_kernel_fixed(n, sigma)
            assert int(t.sum()) == 256 and np.array_equal(t, t[::-1])      # the centre takes the diffused rounding error, so it
            assert abs(int(t[n // 2]) - int(t.max())) <= 1                     # may sit one count under its neighbours
    flat = np.full((9, 12, 3), 77, np.uint8)
    assert np.array_equal(oracle.gaussian_blur(flat, (11, 7)), flat)
    imp = np.zeros((21, 21), np.uint8)
    imp[10, 10] = 255
    t = oracle.gaussian_kernel_fixed(5).astype(np.int64)
    exp = (np.outer(t, t) * 255 + 32768) >> 16
    assert np.array_equal(oracle.gaussian_blur(imp, (5, 5))[8:13, 8:13], exp)
    rng = np.random.default_rng(4)
    g = rng.integers(0, 256, (6, 8), dtype=np.uint8).astype(np.int64)
    p = np.pad(g, 1, mode="reflect")                       # reflect = BORDER_REFLECT_101
    k = np.array([1, 2, 1])
    s = sum(k[i] * k[j] * p[i:i + 6, j:j + 8] for i in range(3) for j in range(3))
    assert np.array_equal(oracle.gaussian_blur(g.astype(np.uint8), (3, 3)), ((s * 4096 + 32768) >> 16).astype(np.uint8))


def test_warp_affine_restatement(oracle):
    """cv2.warpAffine, classical 8-bit bilinear path: identity and integer shifts are copies, a half-pixel shift is the half-up
    average of neighbours, a quarter turn is a permutation, a general map agrees with an independent numpy restatement of the
    22.10 / 5-bit / 15-bit fixed-point rule, and both border modes behave as documented."""
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    assert np.array_equal(oracle.warp_affine(img, [[1, 0, 0], [0, 1, 0]], (50, 40)), img)
    out = oracle.warp_affine(img, [[1, 0, 7], [0, 1, -3]], (50, 40), value=(1, 2, 3))
    exp = np.empty_like(img)
    exp[:] = (1, 2, 3)
    exp[0:37, 7:50] = img[3:40, 0:43]
    # the row just below the copied block blends source row 39 with the border value at weight 0 -> still the source row... which is
    # outside (sy = 40 is fully out), so it is the plain border value
    assert np.array_equal(out, exp)
    g = rng.integers(0, 256, (8, 16), dtype=np.uint8)
    a = g.astype(int)
    left = np.concatenate([a[:, :1], a[:, :-1]], 1)
    assert np.array_equal(oracle.warp_affine(g, [[1, 0, 0.5], [0, 1, 0]], (16, 8), border="replicate"), ((a + left + 1) >> 1).astype(np.uint8))
    sq = rng.integers(0, 256, (32, 32), dtype=np.uint8)
    M = oracle.rotation_matrix_2d((16, 16), 90)
    assert np.allclose(M, [[0, 1, 0], [-1, 0, 32]])
    q = oracle.warp_affine(sq, M, (32, 32), border="replicate")
    assert np.array_equal(q, np.array([[sq[X, min(32 - Y, 31)] for X in range(32)] for Y in range(32)], np.uint8))

    def witness(src, Minv, dw, dh, replicate, cval):
        sh, sw = src.shape
        x = np.arange(dw, dtype=np.float64)
        y = np.arange(dh, dtype=np.float64)
        ad = np.rint(Minv[0, 0] * x * 1024).astype(np.int64)
        bd = np.rint(Minv[1, 0] * x * 1024).astype(np.int64)
        X0 = np.rint((Minv[0, 1] * y + Minv[0, 2]) * 1024).astype(np.int64) + 16
        Y0 = np.rint((Minv[1, 1] * y + Minv[1, 2]) * 1024).astype(np.int64) + 16
        X = (X0[:, None] + ad[None, :]) >> 5
        Y = (Y0[:, None] + bd[None, :]) >> 5
        sx, sy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31
        acc = np.zeros((dh, dw), np.int64)
        for dy, dx, wgt in ((0, 0, (32 - fx) * (32 - fy)), (0, 1, fx * (32 - fy)), (1, 0, (32 - fx) * fy), (1, 1, fx * fy)):
            xs, ys = sx + dx, sy + dy
            inside = (xs >= 0) & (xs < sw) & (ys >= 0) & (ys < sh)
            v = src[np.clip(ys, 0, sh - 1), np.clip(xs, 0, sw - 1)].astype(np.int64)
            if not replicate:
                v = np.where(inside, v, cval)
            acc += v * wgt * 32
        return ((acc + (1 << 14)) >> 15).astype(np.uint8)

    src = rng.integers(0, 256, (33, 47), dtype=np.uint8)
    for M in (np.array([[0.9, 0.3, -4.0], [-0.2, 1.1, 6.5]]), oracle.rotation_matrix_2d((23.5, 16.5), 33.3, 1.2)):
        Minv = np.linalg.inv(np.vstack([M, [0, 0, 1]]))[:2]
        for rep in (False, True):
            got = oracle.warp_affine(src, Minv, (60, 41), inverse_map=True, border="replicate" if rep else "constant", value=99)
            assert np.array_equal(got, witness(src, Minv, 60, 41, rep, 99))
        # the forward form inverts the matrix itself; away from rounding ties both give the same picture
        fw = oracle.warp_affine(src, M, (60, 41), border="replicate")
        iv = oracle.warp_affine(src, Minv, (60, 41), inverse_map=True, border="replicate")
        assert np.mean(fw != iv) < 0.02 and np.abs(fw.astype(int) - iv.astype(int)).max() <= 12


def test_ycrcb_hls_known_answers(oracle):
    """COLOR_BGR2YCrCb / COLOR_BGR2HLS, 8-bit: the published primaries / secondaries / greys; Y equals BGR2GRAY; greys have no chroma
    and no saturation; HLS lightness is the rounded mean of the extreme channels."""
    import json
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")))
    for bgr, exp in G["bgr2ycrcb"]:
        assert oracle.bgr2ycrcb(np.array([[bgr]], np.uint8))[0, 0].tolist() == exp
    for bgr, exp in G["bgr2hls"]:
        assert oracle.bgr2hls(np.array([[bgr]], np.uint8))[0, 0].tolist() == exp
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    ycc, hls = oracle.bgr2ycrcb(img), oracle.bgr2hls(img)
    assert np.array_equal(ycc[:, :, 0], oracle.bgr2gray(img))
    mx, mn = img.max(axis=2).astype(np.float64), img.min(axis=2).astype(np.float64)
    assert np.abs(hls[:, :, 1] - (mx + mn) / 2).max() <= 0.5 + 1e-3
    assert hls[:, :, 0].max() <= 180     # a hue just under 360 degrees rounds up to 180: the float path has no wrap after rounding
    grey = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)
    assert np.array_equal(oracle.bgr2ycrcb(grey)[0], np.stack([np.arange(256), np.full(256, 128), np.full(256, 128)], 1))
    assert np.array_equal(oracle.bgr2hls(grey)[0], np.stack([np.zeros(256), np.arange(256), np.zeros(256)], 1))
    # hue agrees with the integer HSV hue to within the two roundings
    hsv = oracle.bgr2hsv(img)
    dh = np.abs(hls[:, :, 0].astype(int) - hsv[:, :, 0].astype(int))
    assert np.minimum(dh, 180 - dh).max() <= 1


def test_canny_restatement(oracle):
    """cv2.Canny (3x3, L1): a vertical step marks the column on the dark side of the step only (m > left, m >= right); a flat image has
    no edges; thresholds are ordered and floored; and hysteresis equals 'components of the low-threshold survivors that hold a
    high-threshold survivor' (scipy labelling as the independent witness of the stack flood)."""
    img = np.zeros((12, 20), np.uint8)
    img[:, 10:] = 200
    e = oracle.canny(img, 50, 100)
    assert e[:, 9].min() == 255 and np.count_nonzero(e) == 12
    assert np.array_equal(oracle.canny(img.T.copy(), 50, 100), e.T)
    assert not oracle.canny(np.full((9, 9), 77, np.uint8), 0, 0).any()
    assert not oracle.canny(img, 801, 900).any() and oracle.canny(img, 799.9, 799.9).any()     # the step's gradient magnitude is 800
    rng = np.random.default_rng(21)
    for cn in (1, 3):
        base = ndi.gaussian_filter(rng.normal(0, 1, (90, 130) + ((cn,) if cn > 1 else ())), (3, 3) + ((0,) if cn > 1 else ()))
        g = np.clip(128 + 900 * base, 0, 255).astype(np.uint8)
        lo, hi = 150, 420
        e = oracle.canny(g, lo, hi)
        assert np.array_equal(e, oracle.canny(g, hi, lo)) and np.array_equal(e, oracle.canny(g, lo + 0.7, hi + 0.2))
        weak, strong = oracle.canny(g, lo, lo) > 0, oracle.canny(g, hi, hi) > 0
        assert (strong <= weak).all() and 0 < strong.sum() < weak.sum()
        lab, n = ndi.label(weak, structure=np.ones((3, 3)))
        keep = np.zeros(n + 1, bool)
        keep[np.unique(lab[strong])] = True
        keep[0] = False
        assert np.array_equal(e > 0, keep[lab])


def test_adaptive_threshold_mean_restatement(oracle):
    """cv2.adaptiveThreshold (mean): OpenCV's three roundings of the box mean — Q23 reciprocal on 16-bit sums (windows up to 256 pixels),
    float32 product, double product — coincide with the exact nearest integer for every possible sum and every odd block size up to
    151 (so one arithmetic can be restated); the oracle agrees with a direct numpy evaluation; bias and type behave as documented."""
    import math
    for bs in list(range(3, 17, 2)) + [17, 31, 51, 99, 151]:
        d = bs * bs
        s = np.arange(0, 255 * d + 1, dtype=np.int64)
        exact = (2 * s + d) // (2 * d)
        assert np.array_equal(np.rint(s * (1.0 / d)).astype(np.int64), exact)
        assert np.array_equal(np.rint((s.astype(np.float32) * np.float32(1.0 / d)).astype(np.float64)).astype(np.int64), exact)
        if d <= 256:
            scalef = (1 << 23) / d
            div_scale, div_delta = math.floor(scalef), d // 2
            if scalef - div_scale < 0.5:
                div_delta += 1
            else:
                div_scale += 1
            assert np.array_equal(((s + div_delta) * div_scale) >> 23, exact)
    rng = np.random.default_rng(17)
    img = rng.integers(0, 256, (31, 45), dtype=np.uint8)
    for bs, c in ((3, 0), (5, 2), (11, -3.5), (7, 4.2)):
        r = bs // 2
        p = np.pad(img.astype(np.int64), r, mode="edge")
        ssum = sum(p[i:i + 31, j:j + 45] for i in range(bs) for j in range(bs))
        mean = (2 * ssum + bs * bs) // (2 * bs * bs)
        diff = img.astype(np.int64) - mean
        assert np.array_equal(oracle.adaptive_threshold_mean(img, 255, False, bs, c), np.where(diff > -math.ceil(c), 255, 0).astype(np.uint8))
        assert np.array_equal(oracle.adaptive_threshold_mean(img, 200.4, True, bs, c), np.where(diff <= -math.floor(c), 200, 0).astype(np.uint8))
    flat = np.full((9, 9), 90, np.uint8)
    assert not oracle.adaptive_threshold_mean(flat, 255, False, 3, 0).any() and oracle.adaptive_threshold_mean(flat, 255, False, 3, 1).all()
    assert not oracle.adaptive_threshold_mean(flat, -1, True, 3, 0).any()


def test_fold_bound_holds(oracle):
    """|running mean - exact mean| stays below the bound the kernels rely on (cb_fold_bound in csrc/vp_balance.hip), on long
    sequences of every kind: the bound is a proof, this is a check that it was typed correctly."""
    rng = np.random.default_rng(1)
    for n, gen in ((2_073_600, lambda n: rng.integers(0, 256, n)), (2_073_600, lambda n: np.full(n, 255)), (500_000, lambda n: np.arange(n) % 256),
                   (300_000, lambda n: np.r_[np.zeros(n // 2), np.full(n - n // 2, 255)]), (7, lambda n: rng.integers(0, 256, n))):
        x = gen(n).astype(np.float64)
        avg = 0.0
        k = np.arange(1, n + 1, dtype=np.float64)
        for i in range(n):                          # the fold, literally (float64 scalar arithmetic = the C doubles)
            avg += (x[i] - avg) / k[i]
        exact = float(np.float64(int(x.sum())) / np.float64(n))
        bound = (510.0 + 127.5 * (n + 1.0)) * 2.0 ** -53 * 1.001
        assert abs(avg - exact) <= bound, (n, abs(avg - exact), bound)


def _numpy1_cases():
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "numpy1_color_distance.npz"))
    k = 0
    while f"c{k}_split" in z:
        a = z[f"c{k}_args"]
        yield {"split": [np.ascontiguousarray(p) for p in z[f"c{k}_split"]], "color": tuple(a[:3]), "weights": tuple(a[3:6]),
               "percentile": None if a[6] < 0 else float(a[6]), "distance_arg": float(a[7]), "ignore": [int(i) for i in z[f"c{k}_ignore"]],
               "weights_cp": z[f"c{k}_weights_cp"], "dists": z[f"c{k}_dists"], "distance": float(z[f"c{k}_distance"][0]), "sq": z[f"c{k}_sq"]}
        k += 1
    assert k >= 6 and str(z["numpy_version"][0]).startswith("1.")


def test_color_distance_equals_numpy1_vectors(oracle):
    """The numpy statements of thresh_color_distance (reference utils/color.py:91-103) as numpy 1.26.4 evaluates them
    (tests/golden/make_numpy1_vectors.py, run under the image's conda interpreter): the oracle's float32 restatement reproduces the
    distance image and its uint8 square root bit for bit, i.e. the promotion rules were read correctly."""
    for c in _numpy1_cases():
        wn = np.array([0 if i in c["ignore"] else c["weights"][i] for i in range(3)], np.float64) / np.linalg.norm(c["weights"])
        assert np.array_equal(wn, c["weights_cp"])
        skip = sum(1 << i for i in c["ignore"])
        d2, sq = oracle.color_distance(c["split"], c["color"], wn.astype(np.float32), skip)
        assert np.array_equal(d2.view(np.uint32), c["dists"].view(np.uint32))
        assert np.array_equal(sq, c["sq"])
        if c["percentile"] is not None:                # numpy 2 on the very same float32 image gives the same threshold
            assert min(np.percentile(c["dists"], c["percentile"]), c["distance_arg"] ** 2) == c["distance"]


def _all_4x4_patterns(pitch=5):
    n = 256
    m = np.zeros((n * pitch + 3, n * pitch + 7), np.uint8)
    v = np.arange(1 << 16, dtype=np.uint32).reshape(n, n)
    for b in range(16):
        m[(b // 4):(b // 4) + n * pitch:pitch, (b % 4):(b % 4) + n * pitch:pitch][:n, :n] = ((v >> b) & 1).astype(np.uint8) * 255
    return m


def _check_labelling_against_scipy(oracle, m):
    """Labels, statistics, centroids and the numbering of both scans against SciPy's labelling and plain array operations."""
    ref, nref = ndi.label(m, structure=np.ones((3, 3)))
    for block in (1, 2):
        n, lab, st, ce = oracle.ccl(m, block)
        assert n == nref + 1
        a, b = lab.ravel(), ref.ravel()                                   # one-to-one in both directions: the same partition
        fwd, back = np.zeros(n, np.int64), np.zeros(n, np.int64)
        fwd[a] = b
        back[b] = a
        assert np.array_equal(fwd[a], b) and np.array_equal(back[b], a)
        area = np.bincount(lab.ravel(), minlength=n)
        assert np.array_equal(st[1:, 4], area[1:])
        # numbering, stated without the scan: components in the order of their first 2x2 block in block-raster order (two foreground
        # pixels of one block are always connected, so a block belongs to one component) - or of their first pixel for the pixel scan
        yy, xx = np.nonzero(lab)
        key = ((yy >> 1) * ((m.shape[1] + 1) >> 1) + (xx >> 1)) if block == 2 else (yy * m.shape[1] + xx)
        firstkey = np.full(n, np.iinfo(np.int64).max, np.int64)
        np.minimum.at(firstkey, lab[yy, xx], key)
        assert (np.diff(firstkey[1:]) > 0).all()
        lo_x, lo_y = np.full(n, 1 << 30), np.full(n, 1 << 30)
        hi_x, hi_y = np.full(n, -1), np.full(n, -1)
        np.minimum.at(lo_x, lab[yy, xx], xx); np.minimum.at(lo_y, lab[yy, xx], yy)
        np.maximum.at(hi_x, lab[yy, xx], xx); np.maximum.at(hi_y, lab[yy, xx], yy)
        assert np.array_equal(st[1:, 0], lo_x[1:]) and np.array_equal(st[1:, 1], lo_y[1:])
        assert np.array_equal(st[1:, 2], (hi_x - lo_x + 1)[1:]) and np.array_equal(st[1:, 3], (hi_y - lo_y + 1)[1:])
        l = lab[yy, xx]
        assert np.array_equal(ce[1:, 0], (np.bincount(l, xx, n) / np.maximum(area, 1))[1:]) and np.array_equal(ce[1:, 1], (np.bincount(l, yy, n) / np.maximum(area, 1))[1:])
    return ref, nref


def _check_contours_against_scipy(oracle, m, ref, nref, apart):
    """Contour lists against statements that need no tracing: counts from SciPy's labellings of foreground and background, the union
    of all border points, start pixels and list order, the SIMPLE lists as a filter of the full lists."""
    bg, nbg = ndi.label(m == 0, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    touching = np.unique(np.concatenate([bg[0], bg[-1], bg[:, 0], bg[:, -1]]))
    holes = nbg - len(touching[touching > 0])
    cs, hole_flags = oracle.find_contours(m, 1, 1, with_holes=True)
    assert len(cs) == nref + holes and int(hole_flags.sum()) == holes
    outer = oracle.find_contours(m, 0, 1)
    assert len(outer) <= nref and (not apart or len(outer) == nref)      # apart: no pattern lies inside another's hole
    # the points of all borders together (no approximation) are exactly the foreground pixels with a 4-neighbour in the background or
    # outside the frame - stated with array shifts, no tracing
    f = m > 0
    pad = np.pad(f, 1)
    inner = pad[:-2, 1:-1] & pad[2:, 1:-1] & pad[1:-1, :-2] & pad[1:-1, 2:]
    allp = np.concatenate([c.reshape(-1, 2) for c in cs])
    traced = np.zeros(m.shape, bool)
    traced[allp[:, 1], allp[:, 0]] = True
    assert np.array_equal(traced, f & ~inner)
    # outer borders start at their component's first pixel in raster order, and the list runs from the last start to the first
    first = np.full(nref + 1, -1, np.int64)
    flat = ref.ravel()
    idx = np.nonzero(flat)[0]
    first[flat[idx][::-1]] = idx[::-1]                                    # (reverse assignment: the smallest index wins)
    starts = np.array([int(c[0, 0, 1]) * m.shape[1] + int(c[0, 0, 0]) for c, hflag in zip(cs, hole_flags) if not hflag])
    assert np.array_equal(np.sort(starts), np.sort(first[1:]))
    allstarts = np.array([int(c[0, 0, 1]) * m.shape[1] + int(c[0, 0, 0]) for c in cs])
    assert (np.diff(allstarts) < 0).all()
    # CHAIN_APPROX_SIMPLE = the full point list without the points a border runs straight through (cyclically)
    simple = oracle.find_contours(m, 1, 2)
    assert len(simple) == len(cs)
    for full, short in zip(cs, simple):
        p = full.reshape(-1, 2).astype(np.int64)
        if len(p) > 1:
            p = p[((np.roll(p, -1, axis=0) - p) != (p - np.roll(p, 1, axis=0))).any(axis=1)]
        assert np.array_equal(p, short.reshape(-1, 2))
    for c in cs[:2000]:                                                   # every border pixel is foreground; a border is 8-connected and closed
        p = c.reshape(-1, 2)
        assert (m[p[:, 1], p[:, 0]] > 0).all()
        d = np.abs(np.diff(np.vstack([p, p[:1]]), axis=0))
        assert d.max(initial=0) <= 1


@pytest.mark.parametrize("pitch", [5, 4])
def test_every_4x4_pattern_against_scipy(oracle, pitch):
    """All 65,536 binary 4x4 patterns in one image (apart, and packed edge to edge), SciPy as the witness: the oracle's labelling is
    the same partition with exact statistics, its contour lists hold one outer border per 8-connected component and one hole border
    per 4-connected background region that does not reach the frame, and its rectangle / cross morphology is SciPy's."""
    m = _all_4x4_patterns(pitch)
    ref, nref = _check_labelling_against_scipy(oracle, m)
    _check_contours_against_scipy(oracle, m, ref, nref, apart=pitch == 5)
    for k in (np.ones((3, 3), np.uint8), np.ones((2, 3), np.uint8), np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)):
        # cv2 anchors a kernel at size // 2 for both operators; SciPy's erosion does the same, its dilation mirrors the structure
        # about that centre, which an even axis answers with origin -1
        assert np.array_equal(oracle.morph(oracle.ERODE, m, k) > 0, ndi.binary_erosion(m > 0, structure=k, border_value=1))
        assert np.array_equal(oracle.morph(oracle.DILATE, m, k) > 0,
                              ndi.binary_dilation(m > 0, structure=k, border_value=0, origin=tuple(-1 if s % 2 == 0 else 0 for s in k.shape)))


def test_labelling_of_full_frames_against_scipy(oracle):
    """1080p: the threshold mask of an S1 frame (blobs + salt), the same after OPEN / CLOSE, uniform noise at 50 %."""
    img = F.s1_buoy(0)
    th = oracle.inrange(np.ascontiguousarray(oracle.bgr2lab(img)[:, :, 1]), 150, 255)
    k = np.ones((5, 5), np.uint8)
    masks = [th, oracle.morph(oracle.CLOSE, oracle.morph(oracle.OPEN, th, k), k)]
    g = F.s3_noise(1)[:, :, 0]
    masks += [oracle.inrange(np.ascontiguousarray(g), 128, 255)]
    for i, m in enumerate(masks):
        ref, nref = _check_labelling_against_scipy(oracle, m)
        if i < 2:                                        # (the noise mask has 10^5 borders: its contour lists are checked at a smaller size below)
            _check_contours_against_scipy(oracle, m, ref, nref, apart=False)
    rng = np.random.default_rng(4)
    for p in (0.1, 0.5, 0.9):
        m = F.random_mask(rng, 200, 300, p)
        ref, nref = ndi.label(m, structure=np.ones((3, 3)))
        _check_contours_against_scipy(oracle, m, ref, nref, apart=False)
