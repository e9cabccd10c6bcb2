"""The labelling has two forms behind one entry point: the two-level path (strip components merged by one block per frame) and the
one-level kernels, which also finish every frame the two-level path hands over as crowded.  All of them must produce cv2's result:
every mask here goes through (a) the default, (b) the one-level kernels only, (c) the two-level path with the merge capacity at 0,
so that every frame with foreground is handed over to the one-level kernels, (d) a small capacity, so that a batch mixes both kinds.
Checked against the oracle (labels, stats, centroids bitwise, both numberings)."""
import numpy as np
import pytest

import frames as F

pytestmark = pytest.mark.gpu

CONFIGS = [("two-level", 2, -1), ("one-level", 1, -1), ("hand-over", 2, 0), ("mixed", 2, 6)]


@pytest.fixture(params=CONFIGS, ids=[c[0] for c in CONFIGS])
def ccl_ctx(vp, request):
    ctx = vp.default_context()
    _, levels, cap = request.param
    ctx.set_option(vp.OPT_CCL_LEVELS, levels)
    ctx.set_option(vp.OPT_CCL_MERGE_CAP, cap)
    yield ctx
    ctx.set_option(vp.OPT_CCL_LEVELS, 2)
    ctx.set_option(vp.OPT_CCL_MERGE_CAP, -1)


def _check(oracle, m, numbering, max_labels=None):
    from vision.utils import feature
    ml = m.size + 2 if max_labels is None else max_labels
    n, lab, st, ce = feature.connected_components(m, numbering=numbering, max_labels=ml)
    on, olab, ost, oce = oracle.ccl(m, block=numbering)
    assert n == on
    assert np.array_equal(lab, olab)
    k = min(on, ml)
    assert np.array_equal(st[:k], ost[:k])
    assert np.array_equal(ce[:k].view(np.uint64), oce[:k].view(np.uint64))  # bitwise, NaN included


def _shapes():
    out = {}
    yy, xx = np.mgrid[0:150, 0:300]
    rng = np.random.default_rng(11)
    m = np.zeros((150, 300), np.uint8)
    for _ in range(14):
        cx, cy, r = rng.uniform(0, 300), rng.uniform(0, 150), rng.uniform(3, 45)
        m[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 255
    out["discs over five strips"] = m
    m = np.zeros((150, 300), np.uint8)
    m[5:145, 10] = 255; m[5:145, 290] = 255; m[144, 10:291] = 255; m[5, 40:260] = 255; m[5:100, 40] = 255; m[99, 40:200] = 255
    out["nested U: merges found late, across strips"] = m
    m = np.zeros((150, 300), np.uint8)
    for k in range(0, 300, 6):
        m[:, k] = 255                       # 50 full-height bars: 50 components, each in every strip
    out["bars through every strip"] = m
    m2 = m.copy(); m2[149, :] = 255         # joined in the last row only: one component, its root is the first bar
    out["bars joined at the bottom"] = m2
    m = np.zeros((150, 300), np.uint8)
    m[31, 5:50] = 255; m[32, 49:120] = 255; m[63, 119:180] = 255; m[64, 181:250] = 255   # contacts exactly on strip boundaries (diagonal too)
    out["contacts on the strip boundaries"] = m
    m = np.full((150, 300), 255, np.uint8)
    out["full"] = m
    m = m.copy(); m[70:80, 100:200] = 0
    out["full with a hole"] = m
    m = np.zeros((150, 300), np.uint8)
    m[::2, ::2] = 255                       # isolated pixels: every strip far beyond the table of a strip
    out["isolated pixels"] = m
    m = ((yy + xx) % 2 * 255).astype(np.uint8)
    out["checkerboard"] = m
    m = np.zeros((150, 300), np.uint8)
    m[0:32:2, 0:200:3] = 255                # one crowded strip above an ordinary frame
    m[60:140, 50:250] = 255
    out["one crowded strip"] = m
    return out


@pytest.mark.parametrize("numbering", [2, 1])
def test_shapes(ccl_ctx, oracle, numbering):
    for name, m in _shapes().items():
        try:
            _check(oracle, m, numbering)
        except AssertionError as e:
            raise AssertionError(f"{name}: {e}") from None


# (40, 2500) and (70, 4096): more than 32 words per row - the row masks of the strip-local pass come from two ballots, strips are 16 rows
@pytest.mark.parametrize("h,w", [(1, 1), (7, 3), (33, 65), (64, 64), (97, 257), (130, 1), (1, 130), (200, 420), (40, 2500), (70, 4096), (35, 2049)])
def test_random(ccl_ctx, oracle, h, w):
    rng = np.random.default_rng(h * 31 + w)
    for p in (0.02, 0.2, 0.5, 0.62, 0.97):
        for numbering in (2, 1):
            _check(oracle, F.random_mask(rng, h, w, p), numbering)


def test_wide_frames_with_blobs(ccl_ctx, oracle):
    """Blobs that cross word 32 of a row and several strips, full rows, runs that span all 64 words."""
    yy, xx = np.mgrid[0:90, 0:4096]
    m = np.zeros((90, 4096), np.uint8)
    rng = np.random.default_rng(21)
    for _ in range(30):
        cx, cy, rx, ry = rng.uniform(0, 4096), rng.uniform(0, 90), rng.uniform(5, 400), rng.uniform(3, 30)
        m[((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1] = 255
    m[40, :] = 255                                       # one run through all 64 words
    m[60:62, 1000:3100] = 255
    for numbering in (2, 1):
        _check(oracle, m, numbering)
    m2 = np.full((48, 3840), 255, np.uint8)
    m2[10:20, 2040:2060] = 0
    _check(oracle, m2, 2)


def test_truncated_tables(ccl_ctx, oracle):
    rng = np.random.default_rng(5)
    _check(oracle, F.random_mask(rng, 80, 200, 0.1), 2, max_labels=7)
    _check(oracle, _shapes()["bars through every strip"], 2, max_labels=7)
    _check(oracle, _shapes()["discs over five strips"], 2, max_labels=1)


def test_chain_batch_mixing_both_kinds(ccl_ctx, oracle):
    """One vp_chain_run over frames of different kinds: blobs (resolved by the merge), noise with no morphology (crowded), flat."""
    from vision import _vp
    from vision.utils import chain
    w, h = 320, 200
    frames = np.stack([F.s1_buoy(1, w, h), F.s3_noise(2, w, h), F.s4_flat(0, w, h), F.s1_buoy(3, w, h), F.s4_flat(255, w, h), F.s3_noise(4, w, h)])
    for numbering in (_vp.CCL_BLOCK2X2, _vp.CCL_PIXEL):
        out = chain.run_chain(frames, _vp.BGR2GRAY, (100, 0, 0), (255, 255, 255), [], ccl=1, numbering=numbering, max_labels=w * h + 2)
        for f in range(len(frames)):
            th = oracle.inrange(oracle.bgr2gray(frames[f]), 100, 255)
            on, olab, ost, oce = oracle.ccl(th, block=numbering)
            assert int(out["nlabels"][f]) == on, f
            assert np.array_equal(out["labels"][f], olab), f
            assert np.array_equal(out["stats"][f][:on], ost), f
            assert np.array_equal(out["centroids"][f][:on].view(np.uint64), oce.view(np.uint64)), f


def test_1080p_frame(ccl_ctx, oracle):
    from vision import _vp
    from vision.utils import chain
    fr = F.s1_buoy(0)[None]
    out = chain.run_chain(fr, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), [(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)], ccl=1, max_labels=256)
    ref = oracle.chain(fr[0], oracle.MODE_LAB, (0, 150, 0), (255, 255, 255), [oracle.OPEN, oracle.CLOSE], 5, 5, 2, 256)
    n = ref["nlabels"]
    assert int(out["nlabels"][0]) == n
    assert np.array_equal(out["labels"][0], ref["labels"])
    assert np.array_equal(out["stats"][0][:n], ref["stats"])
    # the threshold mask without morphology: salt pixels and ragged edges, hundreds of small components
    out = chain.run_chain(fr, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), [], ccl=1, max_labels=8192)
    th = oracle.inrange(oracle.bgr2lab(fr[0]), (0, 150, 0), (255, 255, 255))
    on, olab, ost, oce = oracle.ccl(th, block=2)
    assert int(out["nlabels"][0]) == on
    assert np.array_equal(out["labels"][0], olab)
    assert np.array_equal(out["stats"][0][:on], ost)
    assert np.array_equal(out["centroids"][0][:on].view(np.uint64), oce.view(np.uint64))


_NOISE_REF = {}


def _noise_ref(oracle, lo, numbering):
    key = (lo, numbering)
    if key not in _NOISE_REF:
        frames = [F.s3_noise(i) for i in range(2)]
        _NOISE_REF[key] = (frames, [oracle.ccl(oracle.inrange(oracle.bgr2gray(fr), lo, 255), block=numbering) for fr in frames])
    return _NOISE_REF[key]


@pytest.mark.parametrize("lo,numbering", [(230, 2), (190, 1), (128, 2)], ids=["2pc", "10pc", "50pc"])
def test_1080p_noise_frames(ccl_ctx, oracle, lo, numbering):
    """Raw noise at 2 % / 10 % / 50 % density straight into the labelling at 1080p (17k / 143k / 7k components per frame, every strip
    crowded), several copies in one batch next to an ordinary frame: labels, statistics and centroids of every copy equal the oracle's."""
    from vision import _vp
    from vision.utils import chain
    frames, refs = _noise_ref(oracle, lo, numbering)
    batch = np.stack([frames[0], frames[1], F.s4_flat(0), frames[0], frames[1]])
    ml = 1 << 18
    out = chain.run_chain(batch, _vp.BGR2GRAY, (lo, 0, 0), (255, 255, 255), [], ccl=1, numbering=numbering, max_labels=ml)
    for f, k in ((0, 0), (1, 1), (3, 0), (4, 1)):
        on, olab, ost, oce = refs[k]
        assert int(out["nlabels"][f]) == on, (f, int(out["nlabels"][f]), on)
        assert np.array_equal(out["labels"][f], olab), f
        assert np.array_equal(out["stats"][f][:on], ost), f
        assert np.array_equal(out["centroids"][f][:on].view(np.uint64), oce.view(np.uint64)), f
        assert not out["stats"][f][on:].any()
    assert int(out["nlabels"][2]) == 1 and not out["labels"][2].any()


@pytest.mark.parametrize("lo,numbering", [(190, 2), (128, 1)], ids=["10pc", "50pc"])
def test_4k_noise_frame(ccl_ctx, oracle, lo, numbering):
    """The crowded-frame kernels at the other plan 1080p does not take: 3840 columns = 60 words per row, 8-row strips of 15,360 ids (270
    strips per frame), one noise frame beside an empty one."""
    from vision import _vp
    from vision.utils import chain
    fr = F.s3_noise(3, 3840, 2160)
    on, olab, ost, oce = oracle.ccl(oracle.inrange(oracle.bgr2gray(fr), lo, 255), block=numbering)
    batch = np.stack([np.zeros_like(fr), fr])
    out = chain.run_chain(batch, _vp.BGR2GRAY, (lo, 0, 0), (255, 255, 255), [], ccl=1, numbering=numbering, max_labels=1 << 20)
    assert int(out["nlabels"][1]) == on, (int(out["nlabels"][1]), on)
    assert np.array_equal(out["labels"][1], olab)
    assert np.array_equal(out["stats"][1][:on], ost)
    assert np.array_equal(out["centroids"][1][:on].view(np.uint64), oce.view(np.uint64))
    assert not out["stats"][1][on:].any()
    assert int(out["nlabels"][0]) == 1 and not out["labels"][0].any()
