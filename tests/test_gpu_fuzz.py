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
