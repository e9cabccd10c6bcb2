"""GPU parity: every operator of the hot path, called through the C ABI (libvp.so via the `vision`
mirror), bit-exact against the CPU oracle on the same seeded inputs.  Integer / byte / index work
=> exact equality; the only float op (colour distance, float32) is also compared exactly because
both sides perform the same IEEE operations in the same order."""
import numpy as np
import pytest

import frames as F

pytestmark = pytest.mark.gpu

SIZES = [(1, 1), (3, 5), (7, 64), (17, 63), (33, 65), (64, 128), (48, 200), (101, 257)]


def _rand_bgr(rng, h, w):
    return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)


def test_tables_match_oracle(vp, oracle):
    g, c, s, hd, lc = vp.get_tables()
    og, oc, os_, oh, olc = oracle.tables()
    assert np.array_equal(g, og) and np.array_equal(c, oc)
    assert np.array_equal(s, os_) and np.array_equal(hd, oh) and np.array_equal(lc, olc)


@pytest.mark.parametrize("h,w", SIZES)
def test_cvt_color(vp, oracle, h, w):
    from vision.utils import color
    rng = np.random.default_rng(h * 1000 + w)
    img = _rand_bgr(rng, h, w)
    lab, (l, a, b) = color.bgr_to_lab(img)
    ref = oracle.bgr2lab(img)
    assert np.array_equal(lab, ref)
    assert np.array_equal(np.dstack([l, a, b]), ref)
    hsv, planes = color.bgr_to_hsv(img)
    ref = oracle.bgr2hsv(img)
    assert np.array_equal(hsv, ref) and np.array_equal(np.dstack(planes), ref)
    gray, (g0,) = color.bgr_to_gray(img)
    assert np.array_equal(gray, oracle.bgr2gray(img)) and np.array_equal(g0, gray)
    back, _ = color.gray_to_bgr(gray)
    assert np.array_equal(back, oracle.gray2bgr(gray))
    ycc, planes = color.bgr_to_ycrcb(img)
    assert np.array_equal(ycc, oracle.bgr2ycrcb(img)) and np.array_equal(np.dstack(planes), ycc)
    hls, planes = color.bgr_to_hls(img)
    assert np.array_equal(hls, oracle.bgr2hls(img)) and np.array_equal(np.dstack(planes), hls)


def test_cvt_color_exhaustive_slices(vp, oracle):
    """every (b,g) x 7 r-values (test_cvt_color_all_colours below runs all 2^24 triples)"""
    from vision.utils import color
    b, g = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    for r in (0, 1, 37, 128, 200, 254, 255):
        img = np.dstack([b, g, np.full_like(b, r)])
        assert np.array_equal(color.bgr_to_lab(img)[0], oracle.bgr2lab(img))
        assert np.array_equal(color.bgr_to_hsv(img)[0], oracle.bgr2hsv(img))
        assert np.array_equal(color.bgr_to_ycrcb(img)[0], oracle.bgr2ycrcb(img))
        assert np.array_equal(color.bgr_to_hls(img)[0], oracle.bgr2hls(img))


def test_cvt_color_all_colours(vp, oracle):
    """every one of the 2^24 BGR triples, for the conversions whose oracle is fast enough (float32 HLS is the one where a fused or
    reordered operation on the device would show)"""
    from vision.utils import color
    v = np.arange(1 << 24, dtype=np.uint32)
    img = np.stack([(v & 255), (v >> 8) & 255, v >> 16], axis=1).astype(np.uint8).reshape(4096, 4096, 3)
    assert np.array_equal(color.bgr_to_hls(img)[0], oracle.bgr2hls(img))
    assert np.array_equal(color.bgr_to_ycrcb(img)[0], oracle.bgr2ycrcb(img))
    assert np.array_equal(color.bgr_to_hsv(img)[0], oracle.bgr2hsv(img))
    olab = oracle.bgr2lab(img)
    assert np.array_equal(color.bgr_to_lab(img)[0], olab)
    assert np.array_equal(color.bgr_to_gray(img)[0], oracle.bgr2gray(img))
    # the fused convert + threshold kernels of the batched chain over the same 2^24 colours: every single-channel form of the LAB
    # kernel (it computes only the channels the bounds constrain), the three-channel form, HSV and grey
    from vision import _vp
    from vision.utils import chain
    ohsv, ogray = oracle.bgr2hsv(img), oracle.bgr2gray(img)
    for mode, conv, lo, hi in ((_vp.BGR2LAB, olab, (60, 0, 0), (200, 255, 255)), (_vp.BGR2LAB, olab, (0, 150, 0), (255, 255, 255)),
                               (_vp.BGR2LAB, olab, (0, 0, 90), (255, 255, 140)), (_vp.BGR2LAB, olab, (20, 110, 100), (240, 160, 170)),
                               (_vp.BGR2HSV, ohsv, (10, 20, 60), (30, 100, 255)), (_vp.BGR2HSV, ohsv, (170, 0, 0), (179, 255, 255))):
        out = chain.run_chain(img[None], mode, lo, hi, [], ccl=0, want=("threshed",))
        assert np.array_equal(out["threshed"][0], oracle.inrange(conv, lo, hi)), (mode, lo, hi)
    out = chain.run_chain(img[None], _vp.BGR2GRAY, (100, 0, 0), (180, 255, 255), [], ccl=0, want=("threshed",))
    assert np.array_equal(out["threshed"][0], oracle.inrange(ogray, 100, 180))


def test_cvt_color_strided_view(vp, oracle):
    from vision.utils import color
    rng = np.random.default_rng(5)
    big = _rand_bgr(rng, 40, 90)
    view = big[3:30, 7:71]
    assert np.array_equal(color.bgr_to_lab(view)[0], oracle.bgr2lab(np.ascontiguousarray(view)))


@pytest.mark.parametrize("h,w", SIZES)
def test_inrange(vp, oracle, h, w):
    from vision.utils import color
    rng = np.random.default_rng(h * 77 + w)
    img = _rand_bgr(rng, h, w)
    for lo, hi in [((10, 20, 60), (30, 100, 255)), ((0, 0, 0), (255, 255, 255)), ((200, 0, 0), (100, 255, 255)),
                   ((-5, 10, 10), (300, 260, 250))]:
        assert np.array_equal(color.range_threshold(img, np.array(lo), np.array(hi)), oracle.inrange(img, lo, hi))
    ch = img[:, :, 1]
    for lo, hi in [(150, 255), (0, 255), (0, 0), (255, 255), (200, 100), (-3, 90), (90, 1000), (149.5, 150.5)]:
        exp = oracle.inrange(np.ascontiguousarray(ch), int(np.rint(lo)), int(np.rint(hi)))
        assert np.array_equal(color.range_threshold(ch, lo, hi), exp)
    f = rng.normal(0, 100, (h, w)).astype(np.float32)
    assert np.array_equal(color.range_threshold(f, 0, 55.5), oracle.inrange(f, 0.0, 55.5))


@pytest.mark.parametrize("h,w", [(5, 7), (64, 64), (33, 130)])
def test_thresh_color_distance(vp, oracle, h, w):
    from vision.utils import color
    rng = np.random.default_rng(w)
    img = _rand_bgr(rng, h, w)
    split = [np.ascontiguousarray(img[:, :, c]) for c in range(3)]
    for ignore, weights in [([], (1, 1, 1)), ([0], (1, 2, 3)), ([1, 2], (0.5, 1, 1))]:
        mask, dist = color.thresh_color_distance(split, (100, 150, 60), 70.0, ignore_channels=ignore, weights=weights)
        wn = np.array([0 if i in ignore else weights[i] for i in range(3)], np.float64) / np.linalg.norm(weights)
        skip = sum(1 << i for i in ignore)
        d2, sq = oracle.color_distance(split, (100, 150, 60), wn.astype(np.float32), skip)
        assert np.array_equal(dist, sq)
        assert np.array_equal(mask, oracle.inrange(d2, 0.0, float(np.float32(70.0 ** 2))))


def test_thresh_color_distance_equals_numpy1_vectors(vp, oracle):
    """The HIP path against what numpy 1.26.4 itself computes for the reference's statements (tests/golden/numpy1_color_distance.npz):
    distance image -> mask through inRange, uint8 square root, the percentile-derived threshold."""
    from test_oracle import _numpy1_cases
    from vision.utils import color
    for c in _numpy1_cases():
        mask, dist = color.thresh_color_distance(c["split"], c["color"], c["distance_arg"], auto_distance_percentile=c["percentile"],
                                                 ignore_channels=c["ignore"], weights=c["weights"])
        assert np.array_equal(dist, c["sq"])
        assert np.array_equal(mask, oracle.inrange(c["dists"], 0.0, float(np.float32(c["distance"]))))


def test_structuring_elements(vp, oracle):
    from vision.utils import transform
    for k in (1, 3, 5, 7, 9, 11, 21, 51, 101):
        assert np.array_equal(transform.elliptic_kernel(k), oracle.structuring_element(oracle.MORPH_ELLIPSE, k, k))
    assert np.array_equal(transform.elliptic_kernel(9, 5), oracle.structuring_element(oracle.MORPH_ELLIPSE, 9, 5))
    assert np.array_equal(transform.rect_kernel(4, 3), np.ones((3, 4), np.uint8))
    with pytest.raises(ValueError):
        transform.elliptic_kernel(4)
    with pytest.raises(ValueError):
        transform.rect_kernel(0)


@pytest.mark.parametrize("h,w", SIZES)
def test_morph_binary_rect(vp, oracle, h, w):
    """0/255 masks + rect kernels: the bit-plane LDS path."""
    from vision.utils import transform as T
    rng = np.random.default_rng(h * 31 + w)
    for p in (0.2, 0.5, 0.9):
        m = F.random_mask(rng, h, w, p)
        for kx, ky in [(5, 5), (3, 3), (1, 7), (4, 2), (9, 3)]:
            k = T.rect_kernel(kx, ky)
            assert np.array_equal(T.erode(m, k), oracle.morph(oracle.ERODE, m, k)), (kx, ky, "erode")
            assert np.array_equal(T.dilate(m, k), oracle.morph(oracle.DILATE, m, k)), (kx, ky, "dilate")
            assert np.array_equal(T.morph_remove_noise(m, k), oracle.morph(oracle.OPEN, m, k)), (kx, ky, "open")
            assert np.array_equal(T.morph_close_holes(m, k), oracle.morph(oracle.CLOSE, m, k)), (kx, ky, "close")
            assert np.array_equal(T.morph_borders(m, k), oracle.morph(oracle.GRADIENT, m, k)), (kx, ky, "gradient")
        k = T.rect_kernel(3)
        assert np.array_equal(T.erode(m, k, iterations=3), oracle.morph(oracle.ERODE, m, k, iterations=3))
        assert np.array_equal(T.morph_remove_noise(m, k, iterations=2), oracle.morph(oracle.OPEN, m, k, iterations=2))


def test_morph_binary_large_kernels(vp, oracle):
    from vision.utils import transform as T
    rng = np.random.default_rng(9)
    m = F.random_mask(rng, 150, 300, 0.97)
    for kx, ky in [(71, 5), (5, 71), (101, 101), (129, 3)]:
        k = T.rect_kernel(kx, ky)
        assert np.array_equal(T.erode(m, k), oracle.morph(oracle.ERODE, m, k, fast=True))
        assert np.array_equal(T.morph_close_holes(255 - m, k), oracle.morph(oracle.CLOSE, 255 - m, k, fast=True))


@pytest.mark.parametrize("h,w", [(9, 11), (40, 70), (64, 64)])
def test_morph_grey_and_ellipse(vp, oracle, h, w):
    """grey-level images, ellipse kernels, 3 channels (modules/preprocessor.py:120-129): generic path."""
    from vision.utils import transform as T
    rng = np.random.default_rng(h + w)
    g = rng.integers(0, 256, (h, w), dtype=np.uint8)
    c3 = _rand_bgr(rng, h, w)
    for img in (g, c3):
        for k in (T.elliptic_kernel(3), T.elliptic_kernel(7), T.elliptic_kernel(11, 5), T.rect_kernel(5), T.rect_kernel(2, 3)):
            assert np.array_equal(T.erode(img, k), oracle.morph(oracle.ERODE, img, k))
            assert np.array_equal(T.dilate(img, k), oracle.morph(oracle.DILATE, img, k))
        k = T.elliptic_kernel(5)
        assert np.array_equal(T.erode(img, k, iterations=2), oracle.morph(oracle.ERODE, img, k, iterations=2))
        assert np.array_equal(T.morph_remove_noise(img, k), oracle.morph(oracle.OPEN, img, k))
        assert np.array_equal(T.morph_borders(img, k), oracle.morph(oracle.GRADIENT, img, k))
    assert np.array_equal(T.erode(g, None), oracle.morph(oracle.ERODE, g, None))


def test_morph_span_form_large_and_irregular_elements(vp, oracle):
    """The span form of the generic operator (running min/max tables): the preprocessor's largest ellipse (side 2*50+1,
    modules/preprocessor.py:120-129), elements wider than the image, irregular elements with several runs per row, off-centre
    anchors, iterations, 1/3/4 channels, image widths around the table's power-of-two windows."""
    from vision.utils import transform as T
    from vision import _vp
    rng = np.random.default_rng(7)
    imgs = [rng.integers(0, 256, (97, 130), dtype=np.uint8), _rand_bgr(rng, 120, 161), rng.integers(0, 256, (33, 65, 4), dtype=np.uint8)]
    for img in imgs:
        for k in (T.elliptic_kernel(101), T.elliptic_kernel(51, 9), T.elliptic_kernel(5, 77), T.elliptic_kernel(255, 3)):
            assert np.array_equal(T.erode(img, k), oracle.morph(oracle.ERODE, img, k))
            assert np.array_equal(T.dilate(img, k), oracle.morph(oracle.DILATE, img, k))
    img = imgs[1]
    for t in range(6):
        kh, kw = int(rng.integers(1, 12)), int(rng.integers(2, 40))
        k = (rng.random((kh, kw)) < 0.7).astype(np.uint8)          # several runs per row
        k[kh // 2, kw // 2] = 1
        assert np.array_equal(T.erode(img, k), oracle.morph(oracle.ERODE, img, k)), (kh, kw)
        assert np.array_equal(T.dilate(img, k, iterations=2), oracle.morph(oracle.DILATE, img, k, iterations=2)), (kh, kw)
    cross = np.zeros((21, 21), np.uint8); cross[10, :] = 1; cross[:, 10] = 1
    assert np.array_equal(T.morph_borders(img, cross), oracle.morph(oracle.GRADIENT, img, cross))
    # off-centre anchor through the C ABI
    k = T.elliptic_kernel(31, 15)
    g = imgs[0]
    out = np.empty_like(g)
    ctx = _vp.default_context()
    _vp.check(_vp.lib().vp_morph_u8(ctx.handle, _vp.MORPH_ERODE, _vp.ptr(g), g.shape[1], g.shape[0], 1, _vp.ptr(k), 31, 15, 3, 12, 1, _vp.ptr(out)), ctx.handle)
    assert np.array_equal(out, oracle.morph(oracle.ERODE, g, k, anchor=(3, 12)))


def _check_ccl(vp, oracle, m, numbering):
    from vision.utils import feature
    n, lab, st, ce = feature.connected_components(m, numbering=numbering, max_labels=m.size // 1 + 2)
    on, olab, ost, oce = oracle.ccl(m, block=numbering)
    assert n == on
    assert np.array_equal(lab, olab)
    assert np.array_equal(st, ost)
    assert np.array_equal(ce.view(np.uint64), oce.view(np.uint64))  # bitwise, NaN included


@pytest.mark.parametrize("h,w", SIZES + [(2, 2), (1, 130), (130, 1), (65, 129)])
@pytest.mark.parametrize("numbering", [2, 1])
def test_ccl_random(vp, oracle, h, w, numbering):
    rng = np.random.default_rng(h * 13 + w + numbering)
    for p in (0.05, 0.3, 0.5, 0.6, 0.95):
        _check_ccl(vp, oracle, F.random_mask(rng, h, w, p), numbering)


def test_ccl_edge_cases(vp, oracle):
    for m in (np.zeros((10, 70), np.uint8), np.full((10, 70), 255, np.uint8), np.full((64, 64), 7, np.uint8)):
        _check_ccl(vp, oracle, m, 2)
    # checkerboards: the worst case for component count, and diagonal 8-connectivity
    yy, xx = np.mgrid[0:37, 0:131]
    _check_ccl(vp, oracle, (((yy + xx) % 2) * 255).astype(np.uint8), 2)
    _check_ccl(vp, oracle, (((yy % 2 == 0) & (xx % 2 == 0)) * 255).astype(np.uint8), 2)
    # spiral / U shapes force late merges
    m = np.zeros((40, 140), np.uint8)
    m[2:38, 5] = 255; m[2:38, 130] = 255; m[37, 5:131] = 255; m[2, 20:110] = 255; m[2:30, 20] = 255
    _check_ccl(vp, oracle, m, 2)
    _check_ccl(vp, oracle, m, 1)
    # numbering differs between block and pixel order: A first appears at (1,0), B at (0,10)
    m = np.zeros((4, 16), np.uint8); m[1, 0] = 255; m[0, 10] = 255
    _check_ccl(vp, oracle, m, 2)
    _check_ccl(vp, oracle, m, 1)


def test_ccl_blobs(vp, oracle):
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:270, 0:480]
    m = np.zeros((270, 480), np.uint8)
    for _ in range(25):
        cx, cy, r = rng.uniform(0, 480), rng.uniform(0, 270), rng.uniform(3, 40)
        m[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 255
    _check_ccl(vp, oracle, m, 2)


def test_ccl_max_labels_truncation(vp, oracle):
    from vision.utils import feature
    rng = np.random.default_rng(4)
    m = F.random_mask(rng, 50, 90, 0.2)
    n, lab, st, ce = feature.connected_components(m, max_labels=5)
    on, olab, ost, oce = oracle.ccl(m, block=2)
    assert n == on and n > 5 and st.shape == (5, 5)
    assert np.array_equal(lab, olab) and np.array_equal(st, ost[:5]) and np.array_equal(ce, oce[:5])


CHAINS = [
    ("red_buoy", 0, (0, 150, 0), (255, 255, 255), [(2, 5, 5), (3, 5, 5)]),
    ("bins", 1, (10, 20, 60), (30, 100, 255), [(2, 5, 5)]),
    ("gray", 2, (90, 0, 0), (160, 255, 255), [(3, 3, 3), (0, 3, 1)]),
    ("lab_all", 0, (20, 100, 90), (200, 160, 170), []),
]


def _oracle_chain(oracle, frame, mode, lo, hi, morph, block):
    th = None
    conv = {0: oracle.bgr2lab, 1: oracle.bgr2hsv}.get(mode)
    if mode == 2:
        th = oracle.inrange(oracle.bgr2gray(frame), lo[0], hi[0])
    else:
        th = oracle.inrange(conv(frame), lo, hi)
    cl = th
    for op, kw, kh in morph:
        cl = oracle.morph(op, cl, np.ones((kh, kw), np.uint8), fast=True)
    n, lab, st, ce = oracle.ccl(cl, block=block)
    return th, cl, n, lab, st, ce


@pytest.mark.parametrize("name,mode,lo,hi,morph", CHAINS)
@pytest.mark.parametrize("h,w", [(72, 128), (45, 100), (64, 192)])
def test_chain_small(vp, oracle, name, mode, lo, hi, morph, h, w):
    """fused batched chain vs the oracle, flat (w % 64 == 0) and generic-row kernels."""
    from vision.utils.chain import run_chain
    gen = {"red_buoy": F.s1_buoy, "bins": F.s2_bins}.get(name, F.s3_noise)
    frames = np.stack([gen(i, w, h) for i in range(3)] + [F.s3_noise(9, w, h)])
    out = run_chain(frames, mode, lo, hi, morph, ccl=1, numbering=2, max_labels=h * w)
    for i in range(len(frames)):
        th, cl, n, lab, st, ce = _oracle_chain(oracle, frames[i], mode, lo, hi, morph, 2)
        assert np.array_equal(out["threshed"][i], th), (name, i)
        assert np.array_equal(out["cleaned"][i], cl), (name, i)
        assert out["nlabels"][i] == n
        assert np.array_equal(out["labels"][i], lab)
        assert np.array_equal(out["stats"][i][:n], st)
        assert np.array_equal(out["centroids"][i][:n].view(np.uint64), ce.view(np.uint64))


def test_chain_ccl_on_threshold_mask(vp, oracle):
    """modules/red_buoy.py:38 runs its contour stage on `threshed`, not `cleaned`: ccl = 2."""
    from vision.utils.chain import run_chain
    frames = np.stack([F.s1_buoy(i, 128, 72) for i in range(2)])
    out = run_chain(frames, 0, (0, 150, 0), (255, 255, 255), [(2, 5, 5), (3, 5, 5)], ccl=2, max_labels=128 * 72)
    for i in range(2):
        th, cl, _, _, _, _ = _oracle_chain(oracle, frames[i], 0, (0, 150, 0), (255, 255, 255), [(2, 5, 5), (3, 5, 5)], 2)
        n, lab, st, ce = oracle.ccl(th, block=2)
        assert np.array_equal(out["cleaned"][i], cl)
        assert out["nlabels"][i] == n and np.array_equal(out["labels"][i], lab)


def test_chain_full_size_properties(vp, oracle):
    """1080p (BASELINE config 2): one frame against the oracle end to end, plus size-independent
    properties on a batch: open/close idempotence, cleaned == operator-API result, label image
    consistent with stats (areas, bounding boxes), flat frames."""
    from vision.utils.chain import run_chain
    from vision.utils import transform as T
    frames = np.stack([F.s1_buoy(0), F.s1_buoy(1), F.s4_flat(0), F.s4_flat(255)])
    morph = [(2, 5, 5), (3, 5, 5)]
    out = run_chain(frames, 0, (0, 150, 0), (255, 255, 255), morph, ccl=1, max_labels=4096)
    th, cl, n, lab, st, ce = _oracle_chain(oracle, frames[0], 0, (0, 150, 0), (255, 255, 255), morph, 2)
    assert np.array_equal(out["threshed"][0], th) and np.array_equal(out["cleaned"][0], cl)
    assert out["nlabels"][0] == n and np.array_equal(out["labels"][0], lab)
    assert np.array_equal(out["stats"][0][:n], st) and np.array_equal(out["centroids"][0][:n], ce)
    assert n > 3
    k = T.rect_kernel(5)
    for i in range(len(frames)):
        c = out["cleaned"][i]
        assert np.array_equal(T.morph_close_holes(c, k), c)          # closing is idempotent
        l = out["labels"][i]
        nl = int(out["nlabels"][i])
        assert np.array_equal(l > 0, c > 0)
        assert l.max() == nl - 1
        areas = np.bincount(l.ravel(), minlength=nl)
        assert np.array_equal(areas, out["stats"][i][:nl, 4])
    assert out["nlabels"][2] == 1 and out["nlabels"][3] == 1   # all-zero / all-255 LAB-a frames select nothing


def test_device_resident_chain_matches_host_chain(vp):
    """vp_chain_run (device pointers, what bench.py times) == vp_chain_run_host."""
    import ctypes as C
    from vision.utils.chain import run_chain
    frames = np.stack([F.s1_buoy(i, 256, 144) for i in range(5)])
    n, h, w, _ = frames.shape
    morph = [(2, 5, 5), (3, 5, 5)]
    ref = run_chain(frames, 0, (0, 150, 0), (255, 255, 255), morph, max_labels=64)
    ctx = vp.default_context()
    L = vp.lib()

    def dev(nbytes):
        p = C.c_void_p()
        vp.check(L.vp_dev_alloc(ctx.handle, nbytes, C.byref(p)))
        return p
    npx = n * h * w
    bufs = vp.ChainBuffers()
    bufs.bgr = dev(npx * 3).value
    bufs.threshed, bufs.cleaned, bufs.labels = dev(npx).value, dev(npx).value, dev(npx * 4).value
    bufs.stats, bufs.centroids, bufs.nlabels = dev(n * 64 * 20).value, dev(n * 64 * 16).value, dev(n * 4).value
    vp.check(L.vp_memcpy_h2d(ctx.handle, bufs.bgr, frames.ctypes.data, npx * 3))
    desc = vp.make_chain_desc(w, h, 0, (0, 150, 0), (255, 255, 255), morph, 1, 2, 64)
    for _ in range(2):  # twice: workspace reuse must not leak state between steps
        ctx.chain_run(desc, bufs, n)
    ctx.synchronize()
    lab = np.empty((n, h, w), np.int32)
    st = np.empty((n, 64, 5), np.int32)
    cl = np.empty((n, h, w), np.uint8)
    vp.check(L.vp_memcpy_d2h(ctx.handle, lab.ctypes.data, bufs.labels, npx * 4))
    vp.check(L.vp_memcpy_d2h(ctx.handle, st.ctypes.data, bufs.stats, n * 64 * 20))
    vp.check(L.vp_memcpy_d2h(ctx.handle, cl.ctypes.data, bufs.cleaned, npx))
    assert np.array_equal(lab, ref["labels"]) and np.array_equal(st, ref["stats"]) and np.array_equal(cl, ref["cleaned"])
    assert L.vp_chain_algorithmic_bytes(C.byref(desc), C.byref(bufs), n) == npx * 9
    for p in (bufs.bgr, bufs.threshed, bufs.cleaned, bufs.labels, bufs.stats, bufs.centroids, bufs.nlabels):
        vp.check(L.vp_dev_free(ctx.handle, p))


def test_lab_float32_within_1e_4(vp):
    """North star: "LAB floats match within 1e-4".  Float path vs float64 analytic CIE L*a*b* (tolerance 1e-4 absolute
    on L in [0,100] / a,b in [-127,127]; measured error is ~2e-5)."""
    from vision.utils import color
    rng = np.random.default_rng(0)
    img = rng.random((120, 200, 3)).astype(np.float32)
    img[0, :8] = np.array([[0, 0, 0], [1, 1, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.04045, 0.04045, 0.04045], [0.003, 0.002, 0.001], [0.5, 0.5, 0.5]], np.float32)
    lab, (L, a, b) = color.bgr_to_lab_f32(img)
    rgb = img[:, :, ::-1].astype(np.float64)
    lin = np.where(rgb <= 0.04045, rgb / 12.92, ((rgb + 0.055) / 1.055) ** 2.4)
    M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    xyz = lin @ M.T / np.array([0.950456, 1.0, 1.088754])
    f = np.where(xyz > 216 / 24389, np.cbrt(xyz), 841 / 108 * xyz + 16 / 116)
    ref = np.stack([116 * f[..., 1] - 16, 500 * (f[..., 0] - f[..., 1]), 200 * (f[..., 1] - f[..., 2])], -1)
    err = np.abs(lab.astype(np.float64) - ref).max()
    assert err <= 1e-4, err
    assert np.array_equal(L, lab[:, :, 0]) and abs(lab[0, 1, 0] - 100.0) < 1e-4 and abs(lab[0, 0]).max() < 1e-4


def test_percentile_and_auto_distance(vp, oracle):
    """utils/color.py:98: distance = min(np.percentile(dists, p), distance**2); the order statistics come from the GPU."""
    from vision.utils import color
    rng = np.random.default_rng(2)
    for n in (1, 2, 7, 1000, 70001):
        a = (rng.normal(0, 50, n) ** 2).astype(np.float32)
        if n > 5:
            a[:3] = [-4.0, 0.0, -0.0]
        for q in (0, 0.5, 25, 50, 90, 99.9, 100):
            assert color.percentile_f32(a, q) == float(np.percentile(a.astype(np.float64), q)), (n, q)
    img = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    split = [np.ascontiguousarray(img[:, :, c]) for c in range(3)]
    mask, dist = color.thresh_color_distance(split, (100, 150, 60), 500.0, auto_distance_percentile=20)
    wn = (np.ones(3) / np.linalg.norm((1, 1, 1))).astype(np.float32)
    d2, sq = oracle.color_distance(split, (100, 150, 60), wn, 0)
    thr = min(np.percentile(d2.astype(np.float64), 20), 500.0 ** 2)
    assert np.array_equal(mask, oracle.inrange(d2, 0.0, float(np.float32(thr)))) and 0.18 < (mask > 0).mean() < 0.22


def test_chain_streams_option_is_result_neutral(vp, oracle):
    """VP_OPT_CHAIN_STREAMS splits a batch over internal streams (fork/join): identical results for 1..4."""
    from vision.utils.chain import run_chain
    frames = np.stack([F.s1_buoy(i, 256, 144) for i in range(13)])
    morph = [(2, 5, 5), (3, 5, 5)]
    ctx = vp.default_context()
    ref = None
    try:
        for s in (1, 2, 3, 4):
            ctx.set_option(vp.OPT_CHAIN_STREAMS, s)
            out = run_chain(frames, 0, (0, 150, 0), (255, 255, 255), morph, max_labels=64)
            if ref is None:
                ref = out
                th, cl, n, lab, st, ce = _oracle_chain(oracle, frames[12], 0, (0, 150, 0), (255, 255, 255), morph, 2)
                assert np.array_equal(out["labels"][12], lab) and out["nlabels"][12] == n
            else:
                for k in ("threshed", "cleaned", "labels", "stats", "nlabels"):
                    assert np.array_equal(out[k], ref[k]), (s, k)
    finally:
        ctx.set_option(vp.OPT_CHAIN_STREAMS, 1)
    with pytest.raises(vp.VpError):
        ctx.set_option(vp.OPT_CHAIN_STREAMS, 9)


def test_very_wide_image(vp, oracle):
    """8192 px wide = 128 words per row: beyond the 64-KB LDS budget of the specialised morphology kernels (runtime-plan
    kernel with a larger LDS carve-out) and close to the strip-local labelling limit."""
    from vision.utils import feature, transform as T
    rng = np.random.default_rng(8)
    m = F.random_mask(rng, 70, 8192, 0.7)
    k = T.rect_kernel(5)
    opened = T.morph_remove_noise(m, k)
    assert np.array_equal(opened, oracle.morph(oracle.OPEN, m, k, fast=True))
    n, lab, st, ce = feature.connected_components(opened, max_labels=20000)
    on, olab, ost, oce = oracle.ccl(opened, 2)
    assert n == on and np.array_equal(lab, olab) and np.array_equal(st, ost[:len(st)])
    got = feature.find_contours(opened, 1, 2)
    exp = oracle.find_contours(opened, 1, 2)
    assert len(got) == len(exp) and all(np.array_equal(a, b) for a, b in zip(got, exp))


def test_c_abi_argument_checks(vp):
    """Every entry point answers bad arguments with VP_ERR_INVALID (-1) and a message — never a crash or an exception."""
    import ctypes as C
    L, ctx = vp.lib(), vp.default_context()
    img = np.zeros((4, 4, 3), np.uint8)
    out = np.zeros((4, 4), np.uint8)
    lo = np.zeros(3, np.int32)
    assert L.vp_cvt_color_u8(ctx.handle, 99, vp.ptr(img), 12, 4, 4, vp.ptr(img), None) == -1
    assert L.vp_cvt_color_u8(ctx.handle, 0, None, 12, 4, 4, vp.ptr(img), None) == -1
    assert L.vp_cvt_color_u8(ctx.handle, 0, vp.ptr(img), 5, 4, 4, vp.ptr(img), None) == -1      # stride < row bytes
    assert L.vp_inrange_u8(ctx.handle, vp.ptr(img), 12, 4, 4, 2, vp.ptr(lo), vp.ptr(lo), vp.ptr(out)) == -1
    assert L.vp_morph_u8(ctx.handle, 7, vp.ptr(out), 4, 4, 1, None, 0, 0, -1, -1, 1, vp.ptr(out)) == -1
    assert L.vp_morph_u8(ctx.handle, 0, vp.ptr(out), 4, 4, 1, vp.ptr(out), 3, 3, 5, 5, 1, vp.ptr(out)) == -1   # anchor outside the kernel
    n = C.c_int32()
    assert L.vp_ccl_u8(ctx.handle, vp.ptr(out), 4, 4, 4, 3, None, None, None, 8, C.byref(n)) == -1             # unknown numbering
    assert L.vp_ccl_u8(ctx.handle, vp.ptr(out), 4, 4, 4, 2, None, None, None, 0, C.byref(n)) == -1             # max_labels < 1
    desc = vp.make_chain_desc(0, 4, 0, (0, 0, 0), (255, 255, 255))
    bufs = vp.ChainBuffers()
    assert L.vp_chain_run(ctx.handle, C.byref(desc), C.byref(bufs), 1) == -1
    desc = vp.make_chain_desc(4, 4, 0, (0, 0, 0), (255, 255, 255))
    assert L.vp_chain_run(ctx.handle, C.byref(desc), C.byref(bufs), 1) == -1                                   # no input buffer
    assert b"bgr" in L.vp_last_error(ctx.handle)
    assert L.vp_chain_run(None, C.byref(desc), C.byref(bufs), 1) == -1
    assert L.vp_set_option(ctx.handle, 77, 1) == -1
    # entry points added after the first pass: colour balance, contours in the chain, detector pre / post steps, resize
    assert L.vp_color_balance_u8(ctx.handle, None, 4, 4, vp.CB_DEFAULT, 1, 1, vp.ptr(img)) == -1
    assert L.vp_color_balance_u8(ctx.handle, vp.ptr(img), 4, 4, vp.CB_DEFAULT, 0, 1, vp.ptr(img)) == -1
    assert L.vp_color_balance_u8(ctx.handle, vp.ptr(img), 4, 4, vp.CB_DEFAULT, 3, 1, vp.ptr(img)) == -4      # tiles do not divide the frame
    assert L.vp_color_balance_dev(ctx.handle, None, None, 4, 4, 1, vp.CB_DEFAULT, 1, 1) == -1
    cd = vp.make_contour_desc("cleaned", 0, 2, 8, 64)
    cb = vp.ContourBuffers()
    bufs.bgr = img.ctypes.data
    assert L.vp_chain_run_contours_host(ctx.handle, C.byref(desc), C.byref(bufs), C.byref(cd), C.byref(cb), 1) == -1   # no contour buffers
    cd.mode = 5
    assert L.vp_chain_run_contours_host(ctx.handle, C.byref(desc), C.byref(bufs), C.byref(cd), C.byref(cb), 1) == -1
    f32 = np.zeros((3, 8, 8), np.float32)
    assert L.vp_letterbox_u8_f32(ctx.handle, vp.ptr(img), 4, 4, 0, 8, 114, vp.ptr(f32), None) == -1
    assert L.vp_letterbox_u8_f32(ctx.handle, vp.ptr(img), 4, 4, 8, 8, 300, vp.ptr(f32), None) == -1
    assert L.vp_resize_u8(ctx.handle, vp.ptr(img), 4, 4, 5, 8, 8, vp.ptr(img)) == -1
    boxes = np.zeros((4, 4), np.float32)
    keep = np.zeros(4, np.int32)
    assert L.vp_nms_f32(ctx.handle, vp.ptr(boxes), vp.ptr(boxes), -1, C.c_float(0.5), 0, 4, vp.ptr(keep), C.byref(n)) == -1
    assert L.vp_nms_f32(ctx.handle, None, None, 4, C.c_float(0.5), 0, 4, vp.ptr(keep), C.byref(n)) == -1
    big = np.zeros((20000, 4), np.float32)
    assert L.vp_nms_f32(ctx.handle, vp.ptr(big), vp.ptr(big), 20000, C.c_float(0.5), 0, 4, vp.ptr(keep), C.byref(n)) == -4   # > 16384 candidates
    assert L.vp_nms_f32(ctx.handle, vp.ptr(boxes), vp.ptr(boxes), 0, C.c_float(0.5), 0, 4, vp.ptr(keep), C.byref(n)) == 0 and n.value == 0
    with pytest.raises(TypeError):
        from vision.utils import color
        color.bgr_to_lab(np.zeros((4, 4, 3), np.float64))
    with pytest.raises(ValueError):
        from vision.utils import color
        color.bgr_to_lab(np.zeros((4, 4), np.uint8))


def test_chain_runner_pinned_staging(vp, oracle):
    """Host-fed chain through page-locked staging buffers (vp_host_alloc): same results as the oracle, reusable across runs."""
    from vision.utils.chain import ChainRunner
    frames = np.stack([F.s1_buoy(i, 256, 144) for i in range(4)])
    morph = [(2, 5, 5), (3, 5, 5)]
    r = ChainRunner(4, 144, 256, 0, (0, 150, 0), (255, 255, 255), morph, max_labels=64, want=("cleaned", "labels", "stats"))
    for rep in range(2):
        r.input[:] = frames if rep == 0 else frames[::-1]
        out = r.run()
        src = frames if rep == 0 else frames[::-1]
        for i in range(4):
            th, cl, n, lab, st, ce = _oracle_chain(oracle, src[i], 0, (0, 150, 0), (255, 255, 255), morph, 2)
            assert np.array_equal(out["cleaned"][i], cl) and out["nlabels"][i] == n
            assert np.array_equal(out["labels"][i], lab) and np.array_equal(out["stats"][i][:n], st)
            assert np.array_equal(out["centroids"][i][:n].view(np.uint64), ce.view(np.uint64))
    assert "threshed" not in out


def test_gaussian_blur_u8(vp, oracle):
    """cv2.GaussianBlur on 8-bit images (modules/preprocessor.py:110-114): the fixed-point taps (host) and the two integer passes
    (GPU) against the oracle; kernel sizes from the preprocessor's range, explicit sigmas, rectangular kernels, kernels wider than
    the image (the reflection wraps more than once), 1 / 3 / 4 channels, single-row and single-column images."""
    from vision import cv2_facade as cv2
    from vision.utils import transform as T
    rng = np.random.default_rng(9)
    for n in (1, 3, 5, 7, 9, 11, 31, 101, 201):
        taps = oracle.gaussian_kernel_fixed(n)
        assert int(taps.sum()) == 256 and np.array_equal(taps, taps[::-1])
    imgs = [rng.integers(0, 256, (67, 130), dtype=np.uint8), _rand_bgr(rng, 90, 161), rng.integers(0, 256, (33, 65, 4), dtype=np.uint8),
            rng.integers(0, 256, (1, 40, 3), dtype=np.uint8), rng.integers(0, 256, (40, 1), dtype=np.uint8), rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)]
    for img in imgs:
        for (kw, kh), s1, s2 in [((3, 3), 0, 0), ((5, 5), 0, 0), ((7, 7), 0, 0), ((9, 9), 0, 0), ((11, 11), 0, 0), ((31, 31), 0, 0), ((5, 5), 1.3, 0),
                                 ((21, 3), 4.0, 0.7), ((1, 9), 0, 0), ((101, 101), 0, 0), ((201, 201), 0, 0)]:
            got = cv2.GaussianBlur(img, (kw, kh), s1, sigmaY=s2)      # positional order is cv2's: the 4th positional is dst
            exp = oracle.gaussian_blur(img, (kw, kh), s1, s2)
            assert got.shape == img.shape and np.array_equal(got, exp), (img.shape, kw, kh, s1, s2)
    assert np.array_equal(T.simple_gaussian_blur(imgs[1], 7, 2.0), oracle.gaussian_blur(imgs[1], (7, 7), 2.0))
    with pytest.raises(ValueError):
        T.simple_gaussian_blur(imgs[1], 4, 1.0)
    # independent witness: a float Gaussian with mirrored borders agrees to the quantisation of the 8.8 taps
    import scipy.ndimage as ndi
    g = imgs[0]
    ref = ndi.gaussian_filter(g.astype(np.float64), 0.15 * 31 + 0.35, mode="mirror", truncate=15 / (0.15 * 31 + 0.35))
    assert np.abs(cv2.GaussianBlur(g, (31, 31), 0).astype(np.float64) - ref).max() <= 2.0
    f = F.s1_buoy(0)
    assert np.array_equal(cv2.GaussianBlur(f, (11, 11), 0), oracle.gaussian_blur(f, (11, 11)))


def test_threshold_family(vp):
    """cv2.threshold variants of utils/color.py:124-199 on 8-bit data.  The definition is elementary (compare with floor(thresh)),
    so the expectation is written out in numpy: fractional, negative and > 255 thresholds, single- and 3-channel images."""
    from vision.utils import color
    rng = np.random.default_rng(2)
    for img in (rng.integers(0, 256, (37, 53), dtype=np.uint8), rng.integers(0, 256, (20, 31, 3), dtype=np.uint8)):
        for t in (-3, 0, 0.5, 99, 99.9, 127, 254, 255, 300):
            it = int(np.floor(t))
            above = img.astype(np.int32) > it
            assert np.array_equal(color.max_threshold(img, t), np.where(above, np.clip(it, 0, 255), img).astype(np.uint8)), t
            assert np.array_equal(color.above_threshold(img, t), np.where(above, img, 0).astype(np.uint8)), t
            assert np.array_equal(color.below_threshold(img, t), np.where(above, 0, img).astype(np.uint8)), t
    g = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    assert np.array_equal(color.binary_threshold(g, 100), np.where(g > 100, 255, 0).astype(np.uint8))
    assert np.array_equal(color.binary_threshold_inv(g, 100), np.where(g > 100, 0, 255).astype(np.uint8))


def test_otsu_threshold(vp):
    """cv2.threshold(..., THRESH_OTSU) (utils/color.py:204-217): the histogram comes from the GPU, the scan is OpenCV's
    getThreshVal_Otsu_8u; the expectation restates that scan in numpy float64 and, independently, maximises the between-class
    variance by brute force."""
    from vision.utils import color
    rng = np.random.default_rng(5)
    for img in (np.clip(np.concatenate([rng.normal(60, 12, 4000), rng.normal(180, 20, 6000)]), 0, 255).astype(np.uint8).reshape(100, 100),
                rng.integers(0, 256, (77, 91), dtype=np.uint8), np.full((8, 8), 7, np.uint8), np.repeat(np.array([[10, 200]], np.uint8), 6, 0)):
        t, out = color.otsu_threshold(img)
        h = np.bincount(img.ravel(), minlength=256).astype(np.float64)
        scale = 1.0 / img.size
        mu = float((np.arange(256) * h).sum()) * scale
        mu1 = q1 = 0.0
        best, arg = 0.0, 0.0
        for i in range(256):
            p = h[i] * scale
            mu1 *= q1
            q1 += p
            q2 = 1.0 - q1
            if min(q1, q2) < 1.1920929e-07 or max(q1, q2) > 1.0 - 1.1920929e-07:
                continue
            mu1 = (mu1 + i * p) / q1
            mu2 = (mu - q1 * mu1) / q2
            s = q1 * q2 * (mu1 - mu2) ** 2
            if s > best:
                best, arg = s, float(i)
        assert t == arg
        assert np.array_equal(out, np.where(img > t, 255, 0).astype(np.uint8))
        if h[h > 0].size > 1:   # brute force: the chosen threshold maximises the between-class variance
            var = []
            for k in range(256):
                w0 = h[:k + 1].sum(); w1 = h[k + 1:].sum()
                if w0 == 0 or w1 == 0:
                    var.append(0.0); continue
                m0 = (np.arange(k + 1) * h[:k + 1]).sum() / w0; m1 = (np.arange(k + 1, 256) * h[k + 1:]).sum() / w1
                var.append(w0 * w1 * (m0 - m1) ** 2)
            assert var[int(t)] >= max(var) * (1 - 1e-12)


@pytest.mark.parametrize("h,w", [(40, 2050), (50, 2112), (70, 4100), (33, 3838)])
def test_ccl_wide_frames_both_strip_heights(vp, oracle, h, w):
    """Frames wider than 2048 px use 16-row strips for the strip-local union-find when ceil(w/2) is even and 32-row strips
    otherwise (the root-bitmap slices must not share a word): both cases, both numberings, blobs crossing many strips."""
    rng = np.random.default_rng(w)
    m = F.random_mask(rng, h, w, 0.55)
    m[:, w // 3: w // 3 + 70] = 255                      # a tall blob through every strip
    m[h // 2, :] = 255                                    # and a line through every word of a row
    for numbering in (2, 1):
        _check_ccl(vp, oracle, m, numbering)


# ---- warpAffine (modules/preprocessor.py:130-135,145-149; utils/transform.py rotate / translate) ---------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_warp_affine_u8(vp, oracle, cn):
    from vision import cv2_facade as cvf
    rng = np.random.default_rng(77 + cn)
    for (h, w) in ((37, 53), (120, 200), (1, 9), (64, 1)):
        img = rng.integers(0, 256, (h, w) if cn == 1 else (h, w, cn), dtype=np.uint8)
        mats = [np.array([[1, 0, 0], [0, 1, 0]], np.float64), np.float32([[1, 0, 7], [0, 1, -3]]), np.float32([[1, 0, -2.25], [0, 1, 5.5]]),
                cvf.getRotationMatrix2D((w / 2, h / 2), 30, 1), cvf.getRotationMatrix2D((w / 2, h / 2), -117.3, 1), cvf.getRotationMatrix2D((w / 3, h / 5), 90, 0.7),
                np.array([[0.3, -1.9, 11.2], [2.2, 0.4, -30.0]]), np.array([[0, 0, 0], [0, 0, 0]], np.float64), rng.normal(0, 1.5, (2, 3))]
        for M in mats:
            for (dw, dh) in ((w, h), (w + 13, max(1, h - 5))):
                for border, bname in ((cvf.BORDER_CONSTANT, "constant"), (cvf.BORDER_REPLICATE, "replicate")):
                    for inv in (0, cvf.WARP_INVERSE_MAP):
                        val = (9, 200, 31, 77)[:cn] if border == cvf.BORDER_CONSTANT else 0
                        got = cvf.warpAffine(img, M, (dw, dh), flags=cvf.INTER_LINEAR | inv, borderMode=border, borderValue=val)
                        exp = oracle.warp_affine(img, M, (dw, dh), inverse_map=bool(inv), border=bname, value=val)
                        assert got.shape == exp.shape and np.array_equal(got, exp), (h, w, M, dw, dh, bname, inv)


@pytest.mark.gpu
def test_rotate_translate_utils(vp, oracle):
    from vision.utils import transform
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    t = transform.translate(img, 40, -25)
    exp = np.zeros_like(img)
    exp[:1080 - 25, 40:] = img[25:, :1920 - 40]
    assert np.array_equal(t, exp)                                   # an integer shift is a copy, whatever the interpolation arithmetic
    r = transform.rotate(img, 12.5)
    M = oracle.rotation_matrix_2d((1920 / 2, 1080 / 2), 12.5, 1)
    assert np.array_equal(r, oracle.warp_affine(img, M, (1920, 1080), border="replicate"))
    sq = rng.integers(0, 256, (64, 64), dtype=np.uint8)              # a quarter turn about the centre is a permutation of the pixels
    q = transform.rotate(sq, 90)
    exp = np.array([[sq[X, min(64 - Y, 63)] for X in range(64)] for Y in range(64)], np.uint8)
    assert np.array_equal(q, exp)


@pytest.mark.gpu
def test_warp_affine_argument_checks(vp):
    from vision import _vp
    ctx = _vp.default_context()
    img = np.zeros((4, 4), np.uint8); out = np.zeros((4, 4), np.uint8)
    M = np.eye(2, 3); bad = np.full((2, 3), np.nan)
    L = _vp.lib()
    assert L.vp_warp_affine_u8(ctx.handle, _vp.ptr(img), 4, 4, 1, _vp.ptr(M), 0, 0, None, _vp.ptr(out), 4, 4) == 0
    assert L.vp_warp_affine_u8(ctx.handle, _vp.ptr(img), 4, 4, 1, _vp.ptr(bad), 0, 0, None, _vp.ptr(out), 4, 4) != 0
    assert L.vp_warp_affine_u8(ctx.handle, _vp.ptr(img), 4, 4, 5, _vp.ptr(M), 0, 0, None, _vp.ptr(out), 4, 4) != 0
    assert L.vp_warp_affine_u8(ctx.handle, _vp.ptr(img), 4, 4, 1, _vp.ptr(M), 1, 0, None, _vp.ptr(out), 4, 4) != 0     # INTER_* bits are not flags here
    assert L.vp_warp_affine_u8(ctx.handle, _vp.ptr(img), 4, 4, 1, _vp.ptr(M), 0, 4, None, _vp.ptr(out), 4, 4) != 0     # BORDER_REFLECT_101
    assert L.vp_warp_affine_u8(ctx.handle, None, 4, 4, 1, _vp.ptr(M), 0, 0, None, _vp.ptr(out), 4, 4) != 0


# ---- Canny (utils/feature.py:43-101) ----------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_canny_u8(vp, oracle):
    from vision.utils import feature
    import scipy.ndimage as ndi
    rng = np.random.default_rng(31)
    cases = []
    for (h, w) in ((1, 1), (1, 40), (37, 1), (2, 2), (63, 65), (120, 257), (200, 320)):
        for cn in (1, 3):
            shape = (h, w) if cn == 1 else (h, w, cn)
            cases.append(rng.integers(0, 256, shape, dtype=np.uint8))                                        # noise: every pixel a candidate
            sm = ndi.gaussian_filter(rng.normal(0, 1, shape), (2, 2) + ((0,) if cn > 1 else ()))
            cases.append(np.clip(128 + 700 * sm, 0, 255).astype(np.uint8))                                   # smooth: long edge chains
    cases.append(F.s1_buoy(3, 640, 360))
    cases.append(np.ascontiguousarray(F.s1_buoy(4, 640, 360)[:, :, 2]))
    cases.append(rng.integers(0, 256, (40, 50, 4), dtype=np.uint8))
    for img in cases:
        for (t1, t2) in ((50, 150), (450, 150), (0, 0), (10.9, 30.2), (300, 900), (5000, 6000), (0, 2039)):
            got = feature.canny(img, t1, t2)
            assert got.shape == img.shape[:2] and np.array_equal(got, oracle.canny(img, t1, t2)), (img.shape, t1, t2)
    big = F.s1_buoy(0, 1920, 1080)
    assert np.array_equal(feature.canny(big, 40, 120), oracle.canny(big, 40, 120))
    g = np.ascontiguousarray(big[:, :, 1])
    mid = np.median(g)
    assert np.array_equal(feature.simple_canny(g), oracle.canny(g, int(max(0, 0.67 * mid)), int(min(255, 1.33 * mid))))
    with pytest.raises(Exception):
        feature.canny(np.zeros((4, 4), np.float32), 1, 2)


@pytest.mark.gpu
def test_adaptive_threshold_mean_u8(vp, oracle):
    from vision.utils import color
    rng = np.random.default_rng(41)
    for (h, w) in ((1, 1), (1, 33), (40, 1), (37, 53), (200, 1500), (1080, 1920)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        for bs, c in ((3, 0), (5, 2), (15, -3.5), (17, 4.2), (151, 0.5)):
            if h * w > 500000 and bs > 17:
                continue
            assert np.array_equal(color.adaptive_threshold_mean(img, bs, c), oracle.adaptive_threshold_mean(img, 255, False, bs, c)), (h, w, bs, c)
            assert np.array_equal(color.adaptive_threshold_mean_inv(img, bs, c), oracle.adaptive_threshold_mean(img, 255, True, bs, c)), (h, w, bs, c)
    img = rng.integers(0, 256, (20, 20), dtype=np.uint8)
    for bad in (4, 1, 0, -3, 153):
        with pytest.raises(vp.VpError):
            color.adaptive_threshold_mean(img, bad)
    labels = rng.integers(0, 3, (6, 7))
    masks = color.mask_from_labels(labels, np.zeros((3, 3)))
    assert len(masks) == 3 and all(np.array_equal(m, np.where(labels == i, 255, 0)) for i, m in enumerate(masks))


def test_fused_threshold_kernels_over_many_ranges_all_colours(vp, oracle):
    """The fused convert + threshold kernels compare Lab and HSV ranges as intervals of the integers the channels are made from (the host
    turns each range into such an interval): every one of the 2^24 colours against the oracle's convert-then-inRange for ranges at the
    edges of that translation - single values, the ends of the type, bounds outside it, empty ranges, every channel alone and together."""
    from vision import _vp
    from vision.utils import chain
    v = np.arange(1 << 24, dtype=np.uint32)
    img = np.stack([(v & 255), (v >> 8) & 255, v >> 16], axis=1).astype(np.uint8).reshape(4096, 4096, 3)
    olab, ohsv = oracle.bgr2lab(img), oracle.bgr2hsv(img)
    lab_ranges = [((0, 0, 0), (255, 255, 255)), ((0, 0, 0), (0, 255, 255)), ((255, 0, 0), (255, 255, 255)), ((1, 0, 0), (254, 255, 255)),
                  ((128, 0, 0), (128, 255, 255)), ((0, 128, 0), (255, 128, 255)), ((0, 0, 128), (255, 255, 128)), ((0, 127, 0), (255, 129, 255)),
                  ((0, 0, 0), (255, 0, 255)), ((0, 255, 0), (255, 255, 255)), ((0, 0, 255), (255, 255, 255)), ((0, 0, 0), (255, 255, 0)),
                  ((0, 42, 0), (255, 41, 255)), ((-5, -1, -300), (300, 400, 256)), ((0, 256, 0), (255, 300, 255)), ((0, -9, 0), (255, -1, 255)),
                  ((97, 131, 77), (98, 200, 201)), ((0, 100, 100), (255, 150, 150)), ((50, 0, 120), (60, 255, 136)), ((254, 127, 127), (255, 129, 129))]
    hsv_ranges = [((0, 0, 0), (179, 255, 255)), ((0, 0, 0), (0, 255, 255)), ((179, 0, 0), (179, 255, 255)), ((1, 0, 0), (178, 255, 255)),
                  ((150, 0, 0), (179, 255, 255)), ((0, 0, 0), (29, 255, 255)), ((30, 0, 0), (90, 255, 255)), ((90, 0, 0), (150, 255, 255)),
                  ((0, 0, 0), (179, 0, 255)), ((0, 255, 0), (179, 255, 255)), ((0, 1, 0), (179, 254, 255)), ((0, 0, 0), (179, 255, 0)),
                  ((0, 0, 255), (179, 255, 255)), ((0, 0, 1), (179, 255, 254)), ((60, 0, 0), (59, 255, 255)), ((0, 90, 0), (179, 89, 255)),
                  ((-3, -1, -2), (200, 300, 256)), ((180, 0, 0), (255, 255, 255)), ((0, 0, 0), (-1, 255, 255)), ((10, 20, 60), (30, 100, 255)),
                  ((120, 128, 128), (121, 129, 129)), ((0, 0, 200), (179, 30, 255))]
    for mode, conv, ranges in ((_vp.BGR2LAB, olab, lab_ranges), (_vp.BGR2HSV, ohsv, hsv_ranges)):
        for lo, hi in ranges:
            out = chain.run_chain(img[None], mode, lo, hi, [], ccl=0, want=("threshed",))
            c = conv.astype(np.int32)                                                           # inRange on the oracle's conversion, bounds as given
            exp = (np.all((c >= np.array(lo)) & (c <= np.array(hi)), axis=2) * 255).astype(np.uint8)
            assert np.array_equal(out["threshed"][0], exp), (mode, lo, hi, int((out["threshed"][0] != exp).sum()))
