"""GPU parity for contour extraction (utils/feature.py:5-40): vision.utils.feature.find_contours / outer_contours /
all_contours through the C ABI vs the oracle's sequential restatement of OpenCV's border following — same contours,
same points, same order, same start points."""
import numpy as np
import pytest

import frames as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["one_block", "launches"])
def bookkeeping_form(request, monkeypatch):
    """The single-image entries run the cycle bookkeeping between the two follower passes either in one block (the form a module's mask
    takes) or as launches over the chip (the form chosen after a speckled mask); which one is a guess from the context's last call and
    must not show in the results: every test of this file runs with each form forced (VP_CT_MANY)."""
    monkeypatch.setenv("VP_CT_MANY", "0" if request.param == "one_block" else "1")


def _same(a, b):
    return len(a) == len(b) and all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(a, b))


def _check(vp, oracle, m, mode, method=2):
    from vision.utils import feature
    got, gh = feature.find_contours(m, mode, method, with_holes=True)
    exp, eh = oracle.find_contours(m, mode, method, with_holes=True)
    assert len(got) == len(exp), (len(got), len(exp))
    assert _same(got, exp)
    assert np.array_equal(gh, eh)
    assert all(c.dtype == np.int32 and c.ndim == 3 and c.shape[1:] == (1, 2) for c in got)


def test_known_shapes(vp, oracle):
    from vision.utils import feature
    m = np.zeros((8, 10), np.uint8)
    m[2:6, 3:8] = 255
    (c,) = feature.outer_contours(m)
    assert c.reshape(-1, 2).tolist() == [[3, 2], [3, 5], [7, 5], [7, 2]]      # rectangle: TL, BL, BR, TR
    m[3:5, 4:7] = 0
    cs = feature.all_contours(m)
    assert len(cs) == 2 and cs[1].reshape(-1, 2).tolist() == [[3, 2], [3, 5], [7, 5], [7, 2]]   # newest (the hole) first
    assert len(feature.outer_contours(m)) == 1
    m = np.zeros((5, 5), np.uint8)
    m[1, 1] = 255
    m[3, 3] = 255
    assert [c.reshape(-1, 2).tolist() for c in feature.outer_contours(m)] == [[[3, 3]], [[1, 1]]]
    assert feature.outer_contours(np.zeros((4, 4), np.uint8)) == ()
    (c,) = feature.outer_contours(np.full((3, 4), 255, np.uint8))              # touching the frame on all sides
    assert c.reshape(-1, 2).tolist() == [[0, 0], [0, 2], [3, 2], [3, 0]]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("method", [2, 1])
def test_shapes_vs_oracle(vp, oracle, mode, method):
    yy, xx = np.mgrid[0:90, 0:200]
    m = np.zeros((90, 200), np.uint8)
    m[((xx - 50) ** 2 + (yy - 45) ** 2 <= 40 ** 2) & ((xx - 50) ** 2 + (yy - 45) ** 2 >= 25 ** 2)] = 255   # ring
    m[(xx - 50) ** 2 + (yy - 45) ** 2 <= 10 ** 2] = 255                                                     # disc inside the ring
    m[10:80, 120:190] = 255
    m[20:70, 130:180] = 0
    m[30:60, 140:170] = 255
    m[40:50, 150:160] = 0                                                                                   # nested squares
    m[5, 100:110] = 255                                                                                     # 1-px line
    m[0:3, 0:3] = 255
    m[87:90, 197:200] = 255                                                                                 # corners
    _check(vp, oracle, m, mode, method)
    _check(vp, oracle, np.ascontiguousarray(m.T), mode, method)
    _check(vp, oracle, 255 - m, mode, method)


@pytest.mark.parametrize("h,w", [(1, 1), (1, 70), (70, 1), (2, 2), (17, 63), (33, 65), (64, 128), (48, 200)])
def test_list_mode_random(vp, oracle, h, w):
    """RETR_LIST has no dependence on OpenCV's marks: exact on any mask, including noise."""
    rng = np.random.default_rng(h * 7 + w)
    for p in (0.1, 0.4, 0.5, 0.6, 0.9):
        m = F.random_mask(rng, h, w, p)
        _check(vp, oracle, m, 1, 2)
        _check(vp, oracle, m, 1, 1)


def test_external_mode_blobs_and_cleaned_masks(vp, oracle):
    """RETR_EXTERNAL on the masks the path actually produces: blobs, and threshold masks after OPEN/CLOSE."""
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:270, 0:480]
    m = np.zeros((270, 480), np.uint8)
    for _ in range(30):
        cx, cy, r = rng.uniform(0, 480), rng.uniform(0, 270), rng.uniform(3, 40)
        m[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 255
    for _ in range(10):
        cx, cy, r = rng.uniform(0, 480), rng.uniform(0, 270), rng.uniform(2, 12)
        m[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 0
    _check(vp, oracle, m, 0)
    _check(vp, oracle, m, 1)
    k = np.ones((5, 5), np.uint8)
    for i in range(3):
        th = oracle.inrange(np.ascontiguousarray(oracle.bgr2lab(F.s1_buoy(i, 640, 360))[:, :, 1]), 150, 255)
        cl = oracle.morph(oracle.CLOSE, oracle.morph(oracle.OPEN, th, k, fast=True), k, fast=True)
        _check(vp, oracle, th, 0)          # modules/red_buoy.py:38 runs outer_contours on the raw threshold mask
        _check(vp, oracle, cl, 0)
        _check(vp, oracle, cl, 1)


def test_external_mode_equals_mark_rule_on_noise_and_thin_walls(vp, oracle):
    """OpenCV decides RETR_EXTERNAL nesting from the sign of the last border mark left of a start pixel (and re-tests a rejected
    component at every later left edge); the GPU uses the topological statement (the background left of the component's first
    pixel reaches the frame).  Every crack is swept by exactly one border, so a mark is negative exactly when the pixel's east
    neighbour lies in the traced border's own background region, which makes the two rules agree; checked here on noise and on
    one-pixel walls with clutter inside (400 more masks: tools/exp_external_rule.py)."""
    rng = np.random.default_rng(11)
    for t in range(60):
        h, w = int(rng.integers(5, 80)), int(rng.integers(5, 100))
        m = F.random_mask(rng, h, w, rng.uniform(0.3, 0.7))
        if t % 3 == 0:
            m = np.zeros((h, w), np.uint8)
            for _ in range(5):
                y0, x0 = int(rng.integers(0, h - 3)), int(rng.integers(0, w - 3))
                y1, x1 = int(rng.integers(y0 + 2, h)), int(rng.integers(x0 + 2, w))
                m[y0, x0:x1 + 1] = 255; m[y1, x0:x1 + 1] = 255; m[y0:y1 + 1, x0] = 255; m[y0:y1 + 1, x1] = 255
            m |= (rng.random((h, w)) < 0.08).astype(np.uint8) * 255
        _check(vp, oracle, m, 0, 2)


def test_contour_helpers_on_gpu_contours(vp, oracle):
    from vision.utils import feature
    m = np.zeros((60, 80), np.uint8)
    m[10:40, 20:50] = 255
    m[30:55, 40:70] = 255
    (c,) = feature.outer_contours(m)
    mo = oracle.contour_moments(c)
    assert feature.contour_area(c) == mo["area"]
    assert feature.contour_centroid(c) == (int(mo["m10"] / max(1e-10, mo["m00"])), int(mo["m01"] / max(1e-10, mo["m00"])))


def test_thin_and_periodic_structures(vp, oracle):
    """Patterns that stress the segment-parallel follower: states that sweep several cracks at once (line tips), states that
    sweep none (inner corners), flat edges longer than the 8-column head spacing at every alignment, word boundaries."""
    yy, xx = np.mgrid[0:96, 0:200]
    pats = [
        ((xx + yy) % 2 == 0),                                  # checkerboard: one diagonal-connected component with many holes
        ((xx % 4 == 0) | (yy % 4 == 0)),                       # 1-px grid
        ((xx + yy) % 7 == 0),                                  # diagonal lines
        ((xx - yy) % 5 == 0) | (yy == 48),                     # anti-diagonals crossed by a line
        (yy % 3 == 0) & (xx > 2) & (xx < 197),                 # long horizontal 1-px lines
        (xx % 3 == 0) & (yy > 2) & (yy < 93),                  # long vertical 1-px lines
        (np.abs(xx - 100) + np.abs(yy - 48) <= 40) & (np.abs(xx - 100) + np.abs(yy - 48) >= 38),   # diamond outline
        ((xx // 2 + yy // 2) % 2 == 0),                        # 2x2 checkerboard
    ]
    for p in pats:
        m = np.where(p, 255, 0).astype(np.uint8)
        for mode in (0, 1):
            _check(vp, oracle, m, mode, 2)
            _check(vp, oracle, m, mode, 1)
        _check(vp, oracle, 255 - m, 1, 2)
    for x0 in range(57, 73):                                    # rectangles at every alignment around a word boundary
        for wd in (1, 2, 7, 8, 9, 17):
            m = np.zeros((12, 160), np.uint8)
            m[3:9, x0:x0 + wd] = 255
            m[5:7, x0 + 2:x0 + wd - 2] = 0
            _check(vp, oracle, m, 1, 1)
            _check(vp, oracle, m, 0, 2)


def test_many_heads_global_jump_path_and_capacity_retry(vp, oracle):
    """More than 8192 heads per frame (pointer jumping in global memory instead of LDS) and more contours / points than the
    mirror's first-guess capacities (the call is repeated with the reported totals)."""
    rng = np.random.default_rng(21)
    m = F.random_mask(rng, 200, 300, 0.5)
    _check(vp, oracle, m, 1, 2)
    _check(vp, oracle, m, 1, 1)
    m = F.random_mask(rng, 150, 700, 0.35)
    _check(vp, oracle, m, 1, 1)


def test_full_size_frame_properties(vp, oracle):
    """1080p: noise in RETR_LIST mode against the oracle, plus properties that need no oracle: every contour point is a
    foreground pixel, CHAIN_APPROX_NONE neighbours are 8-adjacent, SIMPLE is a subsequence of NONE with the same start."""
    from vision.utils import feature
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:1080, 0:1920]
    m = np.zeros((1080, 1920), np.uint8)
    for _ in range(60):
        cx, cy, r = rng.uniform(0, 1920), rng.uniform(0, 1080), rng.uniform(5, 200)
        m[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 255
    for _ in range(25):
        cx, cy, r = rng.uniform(0, 1920), rng.uniform(0, 1080), rng.uniform(3, 60)
        m[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 0
    m[500:503, :] = 255                                         # a line across the whole frame, touching both sides
    for mode in (0, 1):
        _check(vp, oracle, m, mode, 2)
    none = feature.find_contours(m, 1, 1)
    simple = feature.find_contours(m, 1, 2)
    assert len(none) == len(simple)
    for a, b in zip(none, simple):
        a, b = a.reshape(-1, 2), b.reshape(-1, 2)
        assert (m[a[:, 1], a[:, 0]] == 255).all()
        if len(a) > 1:
            d = np.abs(np.diff(np.vstack([a, a[:1]]), axis=0))
            assert d.max() == 1
        assert tuple(a[0]) == tuple(b[0])
        ia = {tuple(p) for p in a}
        assert all(tuple(p) in ia for p in b)
    noise = F.random_mask(rng, 1080, 1920, 0.5)
    _check(vp, oracle, noise, 1, 2)
    # RETR_EXTERNAL on speckle (140 k components, chains of "same answer as the component to my left" across the frame)
    _check(vp, oracle, F.random_mask(rng, 1080, 1920, 0.1), 0, 2)


def test_4k_frame(vp, oracle):
    """3840 x 2160 (BASELINE config 4's frame size): discs with holes and a frame-wide bar, both retrieval modes."""
    rng = np.random.default_rng(9)
    yy, xx = np.mgrid[0:2160, 0:3840]
    m = np.zeros((2160, 3840), np.uint8)
    for _ in range(40):
        cx, cy, r = rng.uniform(0, 3840), rng.uniform(0, 2160), rng.uniform(10, 400)
        m[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 255
    for _ in range(20):
        cx, cy, r = rng.uniform(0, 3840), rng.uniform(0, 2160), rng.uniform(5, 120)
        m[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 0
    m[1000:1004, :] = 255
    for mode in (0, 1):
        _check(vp, oracle, m, mode, 2)


@pytest.mark.parametrize("source,mode", [("cleaned", 0), ("threshed", 1)])
def test_chain_with_contours_batch(vp, oracle, source, mode, monkeypatch):
    """vp_chain_run_contours_host: the whole red_buoy body (modules/red_buoy.py:21-38) for a batch - per frame the same
    contours as the oracle's chain followed by its border following; the scratch budget is set so low that the batch takes
    several contour passes."""
    from vision import _vp
    monkeypatch.setenv("VP_CT_SCRATCH_MB", "6")
    from vision.utils import chain
    n, h, w = 19, 144, 256
    frames = np.stack([F.s1_buoy(i, w, h) if i % 3 else F.s2_bins(i, w, h) for i in range(n)])
    morph = ((_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5))
    out = chain.run_chain(frames, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), morph, ccl=1, want=("cleaned", "threshed", "stats"),
                          contours=dict(source=source, mode=mode, method=2, max_contours=4, max_points=64))   # forces a capacity retry
    assert len(out["contours"]) == n
    k = np.ones((5, 5), np.uint8)
    for f in range(n):
        th = oracle.inrange(np.ascontiguousarray(oracle.bgr2lab(frames[f])[:, :, 1]), 150, 255)
        cl = oracle.morph(oracle.CLOSE, oracle.morph(oracle.OPEN, th, k, fast=True), k, fast=True)
        assert np.array_equal(out["threshed"][f], th) and np.array_equal(out["cleaned"][f], cl)
        exp, eh = oracle.find_contours(cl if source == "cleaned" else th, mode, 2, with_holes=True)
        got, gh = out["contours"][f]
        assert _same(got, exp), f
        assert np.array_equal(gh, eh)
        feat = out["contour_features"][f]                      # computed on the device; exact (integer sums)
        assert feat.shape == (len(exp), 8)
        for c, row in zip(exp, feat):
            mo = oracle.contour_moments(c)
            assert (row[0], row[1], row[2], row[3]) == (mo["m00"], mo["m10"], mo["m01"], mo["area"]), (f, row, mo)
            pts = c.reshape(-1, 2)
            assert row[4:].tolist() == [pts[:, 0].min(), pts[:, 1].min(), np.ptp(pts[:, 0]) + 1, np.ptp(pts[:, 1]) + 1]
    # without CCL and without mask outputs the contours are the same
    out2 = chain.run_chain(frames, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), morph, ccl=0, want=(),
                           contours=dict(source=source, mode=mode, method=2, max_contours=256, max_points=1 << 14))
    for f in range(n):
        assert _same(out2["contours"][f][0], out["contours"][f][0])


def test_chain_runner_contours_pinned(vp, oracle):
    from vision import _vp
    from vision.utils import chain
    n, h, w = 3, 144, 256
    r = chain.ChainRunner(n, h, w, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), ((_vp.MORPH_OPEN, 3, 3),), ccl=0, want=(),
                          contours=dict(source="cleaned", mode=0, method=1, max_contours=128, max_points=1 << 14))
    for i in range(n):
        r.input[i] = F.s1_buoy(i, w, h)
    out = r.run()
    k = np.ones((3, 3), np.uint8)
    for f in range(n):
        th = oracle.inrange(np.ascontiguousarray(oracle.bgr2lab(r.input[f])[:, :, 1]), 150, 255)
        cl = oracle.morph(oracle.OPEN, th, k, fast=True)
        assert _same(out["contours"][f][0], oracle.find_contours(cl, 0, 1))


def test_contour_tuple_shares_one_point_block(vp, oracle):
    """The contours of one call are views of one block that the overlay rasteriser reads directly: they are ordinary writable arrays,
    and a change made through one of them is what gets drawn."""
    from vision.utils import feature
    from vision.utils.draw import draw_contours
    m = np.zeros((120, 200), np.uint8)
    m[20:50, 30:80] = 255
    m[70:100, 120:180] = 255
    cs = feature.outer_contours(m)
    assert isinstance(cs, tuple) and len(cs) == 2 and all(type(c) is np.ndarray and c.flags.writeable and c.flags.c_contiguous for c in cs)
    assert _same(cs, oracle.find_contours(m, 0, 2))
    cs[0][:, 0, 0] -= 100                                          # move the newest contour (the lower right box) to the left
    a, b = np.zeros((120, 200, 3), np.uint8), np.zeros((120, 200, 3), np.uint8)
    draw_contours(a, cs, thickness=3)
    draw_contours(b, [c.copy() for c in cs], thickness=3)          # a plain list of copies takes the general path
    assert np.array_equal(a, b) and a[70:100, 20:80].any() and not a[70:100, 120:181].any()
    assert feature.contour_area(cs[0]) == 59.0 * 29.0


@pytest.mark.parametrize("n", [2, 5, 9])
def test_batches_at_1080p_every_strip_height(vp, oracle, n):
    """The strip height of the union-finds inside contour extraction follows the batch (8 rows up to 3 frames of 1080p, 16 up to 7,
    32 beyond): the same contours whichever is in use."""
    from vision import _vp
    from vision.utils import chain
    frames = np.stack([F.s1_buoy(i) if i % 2 else F.s2_bins(i) for i in range(n)])
    out = chain.run_chain(frames, _vp.BGR2LAB, (0, 140, 0), (255, 255, 255), ((_vp.MORPH_OPEN, 3, 3),), ccl=0, want=("cleaned",),
                          contours=dict(source="cleaned", mode=1, method=2, max_contours=4096, max_points=1 << 17))
    k = np.ones((3, 3), np.uint8)
    for f in range(n):
        th = oracle.inrange(np.ascontiguousarray(oracle.bgr2lab(frames[f])[:, :, 1]), 140, 255)
        cl = oracle.morph(oracle.OPEN, th, k, fast=True)
        assert np.array_equal(out["cleaned"][f], cl)
        exp, eh = oracle.find_contours(cl, 1, 2, with_holes=True)
        got, gh = out["contours"][f]
        assert _same(got, exp) and np.array_equal(gh, eh), (n, f, len(got), len(exp))


def _all_4x4_patterns(pitch=5):
    """Every one of the 2^16 binary 4x4 patterns once, in a 256 x 256 grid of cells `pitch` pixels apart (a background gap keeps the
    patterns apart; the pitch of 5 puts them at every alignment against the 64-bit words and the strips of the kernels)."""
    n = 256
    m = np.zeros((n * pitch + 3, n * pitch + 7), np.uint8)
    v = np.arange(1 << 16, dtype=np.uint32).reshape(n, n)
    for b in range(16):
        m[(b // 4):(b // 4) + n * pitch:pitch, (b % 4):(b % 4) + n * pitch:pitch][:n, :n] = ((v >> b) & 1).astype(np.uint8) * 255
    return m


def test_every_4x4_pattern(vp, oracle):
    """Exhaustive over local shapes: all 65,536 patterns of 4x4 pixels (rings with holes, diagonal contacts, single pixels, full
    blocks) in one image - contours in both retrieval modes and approximations, and the component labelling in both numberings."""
    from vision.utils import feature
    m = _all_4x4_patterns()
    for mode, method in ((1, 1), (0, 2), (1, 2)):
        got, gh = feature.find_contours(m, mode, method, with_holes=True)
        exp, eh = oracle.find_contours(m, mode, method, with_holes=True)
        assert len(got) == len(exp) > 60000, (mode, method, len(got), len(exp))
        assert np.array_equal(gh, eh)
        assert np.array_equal(np.concatenate([c.reshape(-1, 2) for c in got]), np.concatenate([c.reshape(-1, 2) for c in exp]))
        assert [len(c) for c in got] == [len(c) for c in exp]
    for numbering in (2, 1):
        n, lab, st, ce = feature.connected_components(m, numbering=numbering, max_labels=200000)
        on, olab, ost, oce = oracle.ccl(m, block=numbering)
        assert n == on and np.array_equal(lab, olab) and np.array_equal(st[:on], ost) and np.array_equal(ce[:on].view(np.uint64), oce.view(np.uint64))
    # a pitch of 4 packs the patterns edge to edge into one texture (components span cells)
    m4 = _all_4x4_patterns(4)
    got = feature.find_contours(m4, 1, 2)
    exp = oracle.find_contours(m4, 1, 2)
    assert len(got) == len(exp) and all(np.array_equal(a, b) for a, b in zip(got, exp))


def test_every_4x4_pattern_through_morphology(vp, oracle):
    """The same exhaustive image through the bit-plane morphology (rectangles, all four operators and the gradient) and the generic
    path (cross, ellipse), both pitches."""
    from vision import _vp
    from vision.utils import transform
    for pitch in (5, 4):
        m = _all_4x4_patterns(pitch)
        for shape, k in ((_vp.SHAPE_RECT, (3, 3)), (_vp.SHAPE_RECT, (5, 3)), (_vp.SHAPE_RECT, (2, 4)), (_vp.SHAPE_CROSS, (3, 3)), (_vp.SHAPE_ELLIPSE, (5, 5))):
            se = transform._structuring_element(shape, k[0], k[1])
            for name, op in (("erode", oracle.ERODE), ("dilate", oracle.DILATE), ("morph_remove_noise", oracle.OPEN), ("morph_close_holes", oracle.CLOSE),
                             ("morph_borders", oracle.GRADIENT)):
                got = getattr(transform, name)(m, se)
                assert np.array_equal(got, oracle.morph(op, m, se)), (pitch, shape, k, name)


@pytest.mark.parametrize("max_labels,numbering", [(256, 2), (3, 1), (2, 2)])
def test_chain_contours_from_the_chains_own_labelling(vp, oracle, max_labels, numbering):
    """With the label image and the statistics among the outputs, the contour pass of the same mask takes every component's first pixel
    from them instead of running its foreground union-find; frames with more labels than the statistics table holds keep the union-find.
    A batch that mixes both kinds (blobs, empty, full, speckle): contours, hole flags and labels as the oracle's, for every table size."""
    from vision import _vp
    from vision.utils import chain
    h, w = 200, 320
    rng = np.random.default_rng(7)
    frames = np.stack([F.s1_buoy(1, w, h), F.s4_flat(0, w, h), F.s1_buoy(2, w, h), F.s4_flat(255, w, h), F.s2_bins(3, w, h), F.s1_buoy(5, w, h)])
    morph = ((_vp.MORPH_OPEN, 3, 3),)
    for mode in (0, 1):
        out = chain.run_chain(frames, _vp.BGR2LAB, (0, 140, 0), (255, 255, 255), morph, ccl=1, numbering=numbering, max_labels=max_labels,
                              want=("cleaned", "labels", "stats"), contours=dict(source="cleaned", mode=mode, method=2, max_contours=512, max_points=1 << 15))
        k = np.ones((3, 3), np.uint8)
        for f in range(len(frames)):
            th = oracle.inrange(np.ascontiguousarray(oracle.bgr2lab(frames[f])[:, :, 1]), 140, 255)
            cl = oracle.morph(oracle.OPEN, th, k, fast=True)
            on, olab, ost, _ = oracle.ccl(cl, block=numbering)
            assert int(out["nlabels"][f]) == on and np.array_equal(out["labels"][f], olab)
            exp, eh = oracle.find_contours(cl, mode, 2, with_holes=True)
            got, gh = out["contours"][f]
            assert _same(got, exp) and np.array_equal(gh, eh), (max_labels, mode, f, len(got), len(exp))


def test_many_contours_come_back_lazily_and_are_traced_once(vp, oracle):
    """A mask with thousands of contours (raw speckle: modules/red_buoy.py:38 runs findContours on the un-cleaned threshold mask): the
    result is the same sequence of (N, 1, 2) views as the tuple for small masks - len, indexing, negative indices, slices, iteration,
    max(key=) - made on demand, and from the second call on the buffers are large enough the first time (no second trace)."""
    from vision.utils import feature
    rng = np.random.default_rng(21)
    mask = (rng.random((270, 480)) < 0.03).astype(np.uint8) * 255
    exp = oracle.find_contours(mask, oracle.RETR_EXTERNAL, oracle.CHAIN_APPROX_SIMPLE)
    assert len(exp) > feature.LazyContourList.THRESHOLD
    feature._capacity.pop(mask.shape, None)
    got = feature.outer_contours(mask)
    assert isinstance(got, feature.LazyContourList) and len(got) == len(exp)
    assert all(np.array_equal(a, b) for a, b in zip(got, exp))
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[-1], exp[-1]) and np.array_equal(got[len(exp) // 2], exp[len(exp) // 2])
    assert got[0].shape[1:] == (1, 2) and got[0].dtype == np.int32
    assert all(np.array_equal(a, b) for a, b in zip(got[3:9], exp[3:9])) and isinstance(got[3:9], tuple)
    with pytest.raises(IndexError):
        got[len(exp)]
    big = max(got, key=feature.contour_area)
    assert feature.contour_area(big) == max(feature.contour_area(c) for c in exp)
    cap = feature._capacity[mask.shape]
    assert cap[0] >= len(exp) and cap[1] >= sum(len(c) for c in exp)
    calls = []
    real = vp.lib().vp_find_contours_u8

    class Counting:
        def __getattr__(self, name):
            if name == "vp_find_contours_u8":
                def f(*a):
                    calls.append(1)
                    return real(*a)
                return f
            return getattr(vp.lib(), name)
    import unittest.mock as mock
    with mock.patch.object(feature._vp, "lib", lambda: Counting()):
        again = feature.outer_contours(mask)
    assert len(calls) == 1 and len(again) == len(exp)
    # the overlay reads the point block of either kind of list
    from vision.utils.draw import draw_contours
    a, b = np.zeros((270, 480, 3), np.uint8), np.zeros((270, 480, 3), np.uint8)
    draw_contours(a, got, thickness=1)
    draw_contours(b, exp, thickness=1)
    assert np.array_equal(a, b)
    # a small mask afterwards: the tuple again; the memory of the big one is kept for 32 small results in a row (speckle that comes and
    # goes is not traced twice each time), then dropped
    small = np.zeros((270, 480), np.uint8)
    small[10:20, 10:20] = 255
    out = feature.outer_contours(small)
    assert isinstance(out, tuple) and len(out) == 1 and mask.shape in feature._capacity
    with mock.patch.object(feature._vp, "lib", lambda: Counting()):
        del calls[:]
        assert len(feature.outer_contours(mask)) == len(exp) and len(calls) == 1        # still one trace
    for _ in range(32):
        feature.outer_contours(small)
    assert mask.shape not in feature._capacity


def test_external_contours_of_nested_components_from_the_chains_labelling(vp, oracle):
    """RETR_EXTERNAL when the chain has labelled the same mask: frames in which no component's box lies strictly inside another's skip
    the background half of the contour pass (every component is external then); frames WITH such a pair resolve it as before - a blob
    inside the hole of a ring is not external, a blob inside the BOX of an L-shaped component but outside its pixels is.  Both kinds in
    one batch, with holes that are never asked for; against the oracle."""
    from vision import _vp
    from vision.utils import chain
    h, w = 160, 256
    red, black = (0, 0, 255), (0, 0, 0)                      # LAB a = 208 / 128: foreground / background of a[150, 255]

    def frame(draw):
        m = np.zeros((h, w), bool)
        draw(m)
        f = np.zeros((h, w, 3), np.uint8)
        f[m] = red
        return f

    def ring_with_blob(m):                                    # nested: the inner blob is NOT external
        m[20:120, 30:200] = True; m[35:105, 45:185] = False; m[60:80, 100:130] = True

    def ring_with_ring_with_blob(m):                          # two levels: only the outermost ring is external
        m[10:150, 10:240] = True; m[20:140, 20:230] = False
        m[40:120, 60:200] = True; m[50:110, 70:190] = False; m[70:90, 120:140] = True

    def l_shape_and_blob(m):                                  # the blob's box lies inside the L's box, the blob itself outside the L: external
        m[20:140, 20:40] = True; m[120:140, 20:220] = True; m[40:70, 100:160] = True

    def apart(m):                                             # no nested boxes: the background half is skipped (holes exist, nobody asks)
        m[10:60, 10:90] = True; m[25:45, 30:60] = False; m[90:150, 120:240] = True; m[100:110, 130:140] = False; m[70:75, 5:9] = True

    def touching_frame(m):
        m[0:30, 0:50] = True; m[h - 20:h, w - 60:w] = True; m[60:100, 100:160] = True; m[70:90, 115:145] = False; m[76:84, 125:135] = True

    frames = np.stack([frame(d) for d in (ring_with_blob, apart, ring_with_ring_with_blob, l_shape_and_blob, touching_frame, apart)])
    for max_labels in (64, 3):
        out = chain.run_chain(frames, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), (), ccl=1, max_labels=max_labels, want=("threshed", "labels", "stats"),
                              contours=dict(source="threshed", mode=0, method=2, max_contours=64, max_points=1 << 13))
        for f in range(len(frames)):
            th = oracle.inrange(np.ascontiguousarray(oracle.bgr2lab(frames[f])[:, :, 1]), 150, 255)
            assert np.array_equal(out["threshed"][f], th)
            exp, eh = oracle.find_contours(th, 0, 2, with_holes=True)
            got, gh = out["contours"][f]
            assert _same(got, exp) and np.array_equal(gh, eh), (max_labels, f, len(got), len(exp))
    assert len(out["contours"][0][0]) == 1 and len(out["contours"][2][0]) == 1 and len(out["contours"][3][0]) == 2 and len(out["contours"][1][0]) == 3


def test_threshold_masks_carry_their_bit_plane_to_the_contour_pass(vp, oracle):
    """range_threshold of a device image with w % 64 == 0 leaves the mask's bit-packed form with the mask; outer_contours of that very
    mask takes it (vp_find_contours_bits_dev: no packing launch) - same contours as the oracle's; the plane is dropped by anything that
    writes to the mask (an in-place draw on the device, a host-side write), after which the contours follow the new contents; other
    widths and aliases never carry one."""
    from vision.devmat import DeviceMat
    from vision.utils import color, feature
    from vision.utils.draw import draw_contours
    ctx = vp.default_context()
    calls = []
    real = vp.lib().vp_find_contours_bits_dev

    def count_bits_entry():
        import unittest.mock as mock

        class Counting:
            def __getattr__(self, name):
                if name == "vp_find_contours_bits_dev":
                    def f(*a):
                        calls.append(1)
                        return real(*a)
                    return f
                return getattr(vp.lib(), name)
        return mock.patch.object(feature._vp, "lib", lambda: Counting())

    for w, h in ((1920, 1080), (640, 360), (64, 5)):
        frame = F.s1_buoy(3, w, h)
        a = color.bgr_to_lab(DeviceMat.from_host(ctx, frame))[1][1]
        th = color.range_threshold(a, 150, 255)
        assert isinstance(th, DeviceMat) and th._bits is not None and th._host is None
        exp_mask = oracle.inrange(np.ascontiguousarray(oracle.bgr2lab(frame)[:, :, 1]), 150, 255)
        del calls[:]
        with count_bits_entry():
            got = feature.outer_contours(th)
        assert calls == [1], "the contour pass did not take the bit plane"
        assert _same(got, oracle.find_contours(exp_mask, 0, 2)) and np.array_equal(np.asarray(th), exp_mask)
        # three-channel inRange makes one too
        hsv = color.bgr_to_hsv(DeviceMat.from_host(ctx, frame))[0]
        m3 = color.range_threshold(hsv, (0, 30, 60), (180, 255, 255))
        assert m3._bits is not None
        assert _same(feature.all_contours(m3), oracle.find_contours(oracle.inrange(oracle.bgr2hsv(frame), (0, 30, 60), (180, 255, 255)), 1, 2))
    # an in-place device write drops the plane; the contours are those of the new contents
    frame = F.s1_buoy(1, 640, 360)
    th = color.range_threshold(color.bgr_to_lab(DeviceMat.from_host(ctx, frame))[1][1], 150, 255)
    box = [np.array([[100, 100], [300, 100], [300, 250], [100, 250]], np.int32).reshape(-1, 1, 2)]
    draw_contours(th, box, color=255, thickness=3)
    assert th._bits is None and th._host is None
    now = np.asarray(th).copy()
    assert now[100, 200] == 255
    th2 = DeviceMat.from_host(ctx, now, binary=True)
    assert _same(feature.outer_contours(th), oracle.find_contours(now, 0, 2)) and _same(feature.outer_contours(th2), oracle.find_contours(now, 0, 2))
    # a host-side write likewise
    th = color.range_threshold(color.bgr_to_lab(DeviceMat.from_host(ctx, frame))[1][1], 150, 255)
    th[50:60, 50:70] = 255
    assert th._bits is None
    assert _same(feature.outer_contours(th), oracle.find_contours(np.asarray(th), 0, 2))
    # widths that are not a multiple of 64, and reshaped aliases: no plane, the ordinary entry
    th = color.range_threshold(color.bgr_to_lab(DeviceMat.from_host(ctx, F.s1_buoy(2, 250, 100)))[1][1], 150, 255)
    assert th._bits is None
    th = color.range_threshold(color.bgr_to_lab(DeviceMat.from_host(ctx, frame))[1][1], 150, 255)
    assert th.reshaped((360, 640, 1))._bits is None


def test_batch_with_speckled_frames_takes_either_form(vp, oracle, monkeypatch):
    """vp_chain_run_contours on a batch that holds raw-noise frames (tens of thousands of border segments each): the first call does the
    bookkeeping in one block per frame (tables in global memory for such frames), the prefix kernel leaves the head counts in pinned
    memory, and the next call - unforced - launches the bookkeeping over the chip.  Same contours as the oracle either way."""
    from vision import _vp
    from vision.utils import chain
    h, w = 200, 320
    frames = np.stack([F.s3_noise(0, w, h), F.s1_buoy(1, w, h), F.s3_noise(2, w, h), F.s4_flat(255, w, h)])
    exp = []
    for f in range(len(frames)):
        th = oracle.inrange(oracle.bgr2gray(frames[f]), 128, 255)
        exp.append((th,) + tuple(oracle.find_contours(th, 1, 2, with_holes=True)))
    cap = max(len(e[1]) for e in exp) + 8
    pts = max(sum(len(c) for c in e[1]) for e in exp) + 64

    def run():
        out = chain.run_chain(frames, _vp.BGR2GRAY, (128, 0, 0), (255, 255, 255), (), ccl=1, want=("threshed", "stats"),
                              contours=dict(source="threshed", mode=1, method=2, max_contours=cap, max_points=pts))
        for f, (th, ec, eh) in enumerate(exp):
            assert np.array_equal(out["threshed"][f], th)
            got, gh = out["contours"][f]
            assert _same(got, ec) and np.array_equal(gh, eh), f
    run()                                            # the form the fixture forces
    monkeypatch.delenv("VP_CT_MANY")
    ctx = vp.default_context()
    run()
    run()                                            # by now the hint of the call before has arrived
    assert vp.lib().vp_contours_last_heads(ctx.handle) > 8192


def test_unexpected_speckle_is_repeated_as_launches(vp, oracle, monkeypatch):
    """A single-image call whose mask turns out to hold more border segments than the one block's LDS tables (no such frame was
    expected: the call before saw a clean mask) is answered by "too many" in place of a result and repeated in the launches form inside
    the same call - same contours as the oracle; the call after it starts in the launches form, and a clean mask brings the block back."""
    from vision.utils import feature
    monkeypatch.delenv("VP_CT_MANY")
    ctx = vp.default_context()
    rng = np.random.default_rng(41)
    clean = np.zeros((360, 640), np.uint8)
    clean[100:200, 200:400] = 255
    noise = F.random_mask(rng, 360, 640, 0.3)
    for m in (clean, noise, noise, clean, clean, noise):
        for mode in (0, 1):
            _check(vp, oracle, m, mode, 2)
        if m is noise:                              # (the diagnostic also remembers the last batched pass: nothing to assert for a clean mask)
            assert vp.lib().vp_contours_last_heads(ctx.handle) > 8192
