"""Seeded differential sweep: random chain descriptors (colour mode, bounds, morphology sequences with odd and even kernels and
iterations, labelled mask, numbering, frame sizes) through vp_chain_run_host / vp_chain_run_contours_host against the oracle run
step by step.  Catches interactions the targeted tests do not enumerate."""
import numpy as np
import pytest

import frames as F

pytestmark = pytest.mark.gpu


def _oracle_chain(oracle, frame, mode, lo, hi, morph):
    conv = {0: oracle.bgr2lab, 1: oracle.bgr2hsv}.get(mode)
    if mode == 2:
        th = oracle.inrange(oracle.bgr2gray(frame), lo[0], hi[0])
    else:
        th = oracle.inrange(conv(frame), lo, hi)
    cl = th
    for op, kw, kh, it in morph:
        k = np.ones((kh, kw), np.uint8)
        cl = oracle.morph({0: oracle.ERODE, 1: oracle.DILATE, 2: oracle.OPEN, 3: oracle.CLOSE}[op], cl, k, iterations=it, fast=True)
    return th, cl


@pytest.mark.parametrize("seed", range(24))
def test_random_chain(vp, oracle, seed):
    from vision import _vp
    from vision.utils import chain
    rng = np.random.default_rng(1000 + seed)
    h, w = int(rng.integers(1, 200)), int(rng.integers(1, 420))
    if seed % 6 == 0:
        w = int(rng.choice([64, 128, 192, 256, 320]))            # widths the flat colour kernel takes
    n = int(rng.integers(1, 4))
    gen = [F.s1_buoy, F.s2_bins, F.s3_noise][seed % 3]
    frames = np.stack([gen(seed * 10 + i, w, h) for i in range(n)])
    mode = int(rng.integers(0, 3))
    lo = [int(v) for v in rng.integers(0, 200, 3)]
    hi = [int(min(255, l + rng.integers(0, 160))) for l in lo]
    if rng.random() < 0.15:
        lo, hi = [0, 0, 0], [255, 255, 255]
    if mode == 1:
        lo[0], hi[0] = lo[0] % 180, min(hi[0], 179)
    morph = []
    for _ in range(int(rng.integers(0, 4))):
        morph.append((int(rng.integers(0, 4)), int(rng.integers(1, 10)), int(rng.integers(1, 10)), int(rng.integers(1, 3))))
    ccl = int(rng.integers(1, 3))
    numbering = int(rng.choice([_vp.CCL_BLOCK2X2, _vp.CCL_PIXEL]))
    max_labels = int(rng.choice([4, 64, 4096]))
    cmode, cmethod = int(rng.integers(0, 2)), int(rng.integers(1, 3))
    src = "cleaned" if rng.random() < 0.7 else "threshed"
    out = chain.run_chain(frames, mode, lo, hi, morph, ccl=ccl, numbering=numbering, max_labels=max_labels,
                          contours=dict(source=src, mode=cmode, method=cmethod, max_contours=8, max_points=256))
    for f in range(n):
        th, cl = _oracle_chain(oracle, frames[f], mode, lo, hi, morph)
        assert np.array_equal(out["threshed"][f], th), (seed, "threshed")
        assert np.array_equal(out["cleaned"][f], cl), (seed, "cleaned")
        m = cl if ccl == 1 else th
        on, olab, ost, oce = oracle.ccl(m, block=numbering)
        assert int(out["nlabels"][f]) == on, (seed, "nlabels")
        assert np.array_equal(out["labels"][f], olab), (seed, "labels")
        k = min(on, max_labels)
        assert np.array_equal(out["stats"][f][:k], ost[:k]) and np.array_equal(out["centroids"][f][:k].view(np.uint64), oce[:k].view(np.uint64))
        if True:                                                   # both retrieval modes are exact on any mask (see test_gpu_contours)
            exp, eh = oracle.find_contours(cl if src == "cleaned" else th, cmode, cmethod, with_holes=True)
            got, gh = out["contours"][f]
            assert len(got) == len(exp) and all(np.array_equal(a, b) for a, b in zip(got, exp)), (seed, "contours")
            assert np.array_equal(gh, eh)


@pytest.mark.parametrize("block", range(4))
def test_random_single_image_contours(vp, oracle, block):
    """Seeded sweep of single-image contour extraction (the form every module's outer_contours / all_contours call takes): random
    sizes - widths that allow 8-, 16- and only 32-row strips in the union-finds -, densities from isolated pixels to nearly full,
    rings, blocks; both retrieval modes and approximations; host and device-resident masks.  tools/fuzz_contours.py runs more seeds."""
    from vision.utils import feature
    from vision.utils.color import range_threshold
    for seed in range(1000 + 12 * block, 1000 + 12 * (block + 1)):
        rng = np.random.default_rng(seed)
        h = int(rng.integers(1, 260))
        w = int(rng.choice([rng.integers(1, 500), 8 * rng.integers(1, 60), 64 * rng.integers(1, 8), 2 * rng.integers(1, 200) + 1]))
        kind = seed % 4
        if kind == 0:
            m = (rng.random((h, w)) < rng.choice([0.01, 0.1, 0.3, 0.5, 0.7, 0.95])).astype(np.uint8) * 255
        elif kind == 1:
            yy, xx = np.mgrid[0:h, 0:w]
            m = np.zeros((h, w), np.uint8)
            for _ in range(int(rng.integers(1, 10))):
                cx, cy, rx, ry = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(1, max(2, w / 3)), rng.uniform(1, max(2, h / 3))
                ring = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2
                m[(ring <= 1) & (ring >= rng.choice([0.0, 0.3, 0.7]))] = 255
        elif kind == 2:
            m = np.full((h, w), 255, np.uint8)
            m[rng.random((h, w)) < 0.02] = 0
        else:
            m = np.ascontiguousarray(np.kron((rng.random(((h + 7) // 8, (w + 7) // 8)) < 0.5).astype(np.uint8) * 255, np.ones((8, 8), np.uint8))[:h, :w])
        mode, method = int(rng.integers(0, 2)), int(rng.integers(1, 3))
        src = range_threshold(m, 128, 255) if seed % 3 == 0 else m
        got, gh = feature.find_contours(src, mode, method, with_holes=True)
        exp, eh = oracle.find_contours(m, mode, method, with_holes=True)
        assert len(got) == len(exp) and all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(got, exp)) and np.array_equal(gh, eh), \
            (seed, h, w, kind, mode, method, len(got), len(exp))
