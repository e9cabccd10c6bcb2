"""GPU parity for the colour balance (utils/color_correction/color_balance.cpp process_frame, modules/color_balance.py
balance) and the 8-bit HSV -> BGR conversion it uses, against the oracle's statement-by-statement restatement."""
import numpy as np
import pytest

import frames as F

pytestmark = pytest.mark.gpu


def test_hsv2bgr_exhaustive(vp, oracle):
    """All 180 x 256 x 256 (h, s, v) triples plus the out-of-range hues 180..255: bit-exact with the oracle's vector form."""
    from vision.utils import color
    h, s, v = np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij")
    hsv = np.stack([h, s, v], -1).astype(np.uint8).reshape(256, 65536, 3)
    got, planes = color.hsv_to_bgr(hsv)
    exp = oracle.hsv2bgr(hsv, 0)
    assert np.array_equal(got, exp)
    assert all(np.array_equal(planes[c], got[:, :, c]) for c in range(3))
    # primaries (SURVEY A2 known answers, inverted)
    px = np.array([[[0, 255, 255], [60, 255, 255], [120, 255, 255], [0, 0, 255], [0, 0, 0], [15, 76, 200]]], np.uint8)
    assert color.hsv_to_bgr(px)[0].reshape(-1, 3).tolist() == [[0, 0, 255], [0, 255, 0], [255, 0, 0], [255, 255, 255], [0, 0, 0], [140, 170, 200]]


FLAG_SETS = [
    dict(),                                                                     # the reference's default call
    dict(hsv_contrast_correct=False),
    dict(equalize_rgb=False),
    dict(rgb_extrema_clipping=False),
    dict(rgb_contrast_correct=True),
    dict(rgb_contrast_correct=True, hsv_contrast_correct=False, rgb_extrema_clipping=False),
    dict(equalize_rgb=False, hsv_contrast_correct=False, rgb_extrema_clipping=False),   # identity
]


@pytest.mark.parametrize("flags", FLAG_SETS)
def test_balance_vs_oracle(vp, oracle, flags):
    from vision.modules.color_balance import balance
    for i, (w, h) in enumerate([(320, 180), (257, 101), (64, 64), (5, 3)]):
        for f in (F.s1_buoy(i, w, h), F.s2_bins(i, w, h), F.s3_noise(i, w, h)):
            got = balance(f, **flags)
            exp = oracle.color_balance(f, mean_mode=1, **flags)
            assert got.shape == f.shape and got.dtype == np.uint8
            assert np.array_equal(got, exp), (flags, w, h, int((got != exp).sum()))
            lit = oracle.color_balance(f, mean_mode=0, **flags)     # the reference's running mean, literally
            assert np.abs(lit.astype(int) - exp.astype(int)).max() <= 1 and (lit != exp).mean() < 1e-3


def test_balance_flat_and_degenerate_frames(vp, oracle):
    """Frames where the reference's ranges collapse (all-equal channels): same output as the oracle's documented choices."""
    from vision.modules.color_balance import balance
    for val in (0, 7, 128, 255):
        f = np.full((40, 60, 3), val, np.uint8)
        assert np.array_equal(balance(f), oracle.color_balance(f, mean_mode=1))
    f = np.zeros((32, 32, 3), np.uint8)
    f[:, :, 2] = 200                                            # pure red: G and B means are 0 -> infinite gains
    assert np.array_equal(balance(f), oracle.color_balance(f, mean_mode=1))
    assert np.array_equal(balance(f, rgb_contrast_correct=True), oracle.color_balance(f, mean_mode=1, rgb_contrast_correct=True))


def test_balance_tiles(vp, oracle):
    from vision.modules.color_balance import balance
    f = F.s1_buoy(3, 320, 180)
    f[:90, :160] = (f[:90, :160] * 0.5).astype(np.uint8)        # one quadrant with a different cast
    for hb, vb in [(2, 2), (4, 3), (1, 5), (8, 1)]:
        for extra in (dict(), dict(hsv_contrast_correct=False)):
            got = balance(f, horizontal_blocks=hb, vertical_blocks=vb, **extra)
            exp = oracle.color_balance(f, horizontal_blocks=hb, vertical_blocks=vb, mean_mode=1, **extra)
            assert np.array_equal(got, exp), (hb, vb, extra)
    with pytest.raises(vp.VpError):
        balance(f, horizontal_blocks=3, vertical_blocks=1)      # 320 % 3 != 0: the reference wraps rows there


def test_balance_adaptive_cast_within_one(vp, oracle):
    """adaptive_cast_correction goes through pow(): device and host libm may round the last bit differently; tolerance 1."""
    from vision.modules.color_balance import balance
    f = F.s1_buoy(1, 320, 180)
    got = balance(f, adaptive_cast_correction=True, hsv_contrast_correct=False)
    exp = oracle.color_balance(f, adaptive_cast_correction=True, hsv_contrast_correct=False, mean_mode=1)
    d = np.abs(got.astype(int) - exp.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


def test_balance_full_size_and_batch(vp, oracle):
    """1080p against the oracle, and the batched device entry against the single-frame entry."""
    import ctypes as C
    from vision import _vp
    from vision.modules.color_balance import balance
    f = F.s1_buoy(0)
    assert np.array_equal(balance(f), oracle.color_balance(f, mean_mode=1))
    n, h, w = 5, 72, 128
    frames = np.stack([F.s1_buoy(i, w, h) if i % 2 else F.s2_bins(i, w, h) for i in range(n)])
    ctx = _vp.default_context()
    L = _vp.lib()
    d = C.c_void_p()
    _vp.check(L.vp_dev_alloc(ctx.handle, frames.nbytes, C.byref(d)), ctx.handle)
    _vp.check(L.vp_memcpy_h2d(ctx.handle, d, _vp.ptr(frames), frames.nbytes), ctx.handle)
    _vp.check(L.vp_color_balance_dev(ctx.handle, d, d, w, h, n, _vp.CB_DEFAULT, 1, 1), ctx.handle)   # in place
    ctx.synchronize()
    out = np.empty_like(frames)
    _vp.check(L.vp_memcpy_d2h(ctx.handle, _vp.ptr(out), d, frames.nbytes), ctx.handle)
    _vp.check(L.vp_dev_free(ctx.handle, d), ctx.handle)
    for i in range(n):
        assert np.array_equal(out[i], balance(frames[i])), i


@pytest.mark.parametrize("flags", [dict(hsi_contrast_correct=True, hsv_contrast_correct=False), dict(hsi_contrast_correct=True),
                                   dict(hsi_contrast_correct=True, hsv_contrast_correct=False, equalize_rgb=False, rgb_extrema_clipping=False)])
def test_balance_hsi_stage_within_one(vp, oracle, flags):
    """hsi_contrast_correct (cpp:678-775): acos / cos come from the device libm, the oracle's from glibc; a last-bit difference
    can move a value across an integer before the truncating cast, so the tolerance is 1 on a small fraction of pixels.
    The order statistics (device radix select vs a sort) are exact, so nothing larger can appear."""
    from vision.modules.color_balance import balance
    for i, (w, h) in enumerate([(320, 180), (257, 101), (33, 7)]):
        for f in (F.s1_buoy(i, w, h), F.s2_bins(i, w, h), F.s3_noise(i, w, h)):
            got = balance(f, **flags)
            exp = oracle.color_balance(f, mean_mode=1, **flags)
            d = np.abs(got.astype(int) - exp.astype(int))
            assert d.max() <= 1, (flags, w, h, int(d.max()))
            assert (d > 0).mean() < 5e-3, (flags, w, h, float((d > 0).mean()))
    g = np.repeat(np.arange(0, 250, 1, dtype=np.uint8)[None, :, None], 40, 0).repeat(3, 2)
    assert np.array_equal(balance(g, equalize_rgb=False, hsv_contrast_correct=False, rgb_extrema_clipping=False, hsi_contrast_correct=True),
                          oracle.color_balance(g, equalize_rgb=False, hsv_contrast_correct=False, rgb_extrema_clipping=False, hsi_contrast_correct=True))
    flat = np.full((16, 16, 3), 90, np.uint8)
    assert np.array_equal(balance(flat, hsi_contrast_correct=True, hsv_contrast_correct=False),
                          oracle.color_balance(flat, hsi_contrast_correct=True, hsv_contrast_correct=False))
