"""GPU suite: the per-operator kernels every module body launches (cvtColor + split, inRange, GRAY2BGR, addWeighted: reference call
sites utils/color.py:11-32, :105-121, modules/bins.py:13-20) in both of their forms - 16 pixels per lane for packed, aligned images
(the form a 1080p frame takes) and the generic one-pixel-per-thread kernels (strided views, unaligned planes, the last npx % 16
pixels) - bit-exact against the oracle for every subset of outputs a caller can ask for."""
import ctypes as C
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CODES = ("BGR2LAB", "BGR2HSV", "BGR2GRAY", "BGR2YCRCB", "BGR2HLS")


class Dev:
    """Raw device buffers through the C ABI (no DeviceMat: the test controls alignment and which outputs exist)."""

    def __init__(self, vp):
        self.vp, self.ctx, self.lib, self.bufs = vp, vp.default_context(), vp.lib(), []

    def alloc(self, nbytes, offset=0):
        p = C.c_void_p()
        self.vp.check(self.lib.vp_dev_alloc(self.ctx.handle, nbytes + 64, C.byref(p)), self.ctx.handle)
        self.bufs.append(p.value)
        return p.value + offset

    def up(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        d = self.alloc(arr.nbytes, offset)
        self.vp.check(self.lib.vp_memcpy_h2d(self.ctx.handle, d, arr.ctypes.data, arr.nbytes), self.ctx.handle)
        return d

    def down(self, d, shape):
        out = np.empty(shape, np.uint8)
        self.vp.check(self.lib.vp_memcpy_d2h(self.ctx.handle, out.ctypes.data, d, out.nbytes), self.ctx.handle)
        return out

    def close(self):
        for p in self.bufs:
            self.lib.vp_dev_free(self.ctx.handle, p)
        self.bufs = []


@pytest.fixture()
def dev(vp):
    d = Dev(vp)
    yield d
    d.lib.vp_set_option(d.ctx.handle, vp.OPT_FLAT_OPS, 1)
    d.close()


def _expect(oracle, code, img):
    if code == "BGR2GRAY":
        return oracle.bgr2gray(img)
    return {"BGR2LAB": oracle.bgr2lab, "BGR2HSV": oracle.bgr2hsv, "BGR2YCRCB": oracle.bgr2ycrcb, "BGR2HLS": oracle.bgr2hls}[code](img)


@pytest.mark.parametrize("flat", [1, 0])
@pytest.mark.parametrize("size", [(1920, 1080), (257, 101), (7, 3), (16, 1), (15, 1)])
def test_conversions_every_output_subset(vp, oracle, dev, flat, size):
    w, h = size
    rng = np.random.default_rng(w * 31 + h)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img[0, :min(w, 6)] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [1, 2, 3]][:min(w, 6)]
    assert dev.lib.vp_set_option(dev.ctx.handle, vp.OPT_FLAT_OPS, flat) == 0
    src = dev.up(img)
    for code in CODES:
        exp = _expect(oracle, code, img)
        dcn = 1 if code == "BGR2GRAY" else 3
        subsets = [(True, ())] if dcn == 1 else [(True, ()), (True, (0, 1, 2)), (False, (0,)), (False, (1,)), (False, (2,)), (False, (0, 2)), (False, (0, 1, 2))]
        for want_dst, planes in subsets:
            d_dst = dev.alloc(h * w * dcn) if want_dst else None
            d_pl = {c: dev.alloc(h * w) for c in planes}
            arr = (C.c_void_p * 3)(*[d_pl.get(c) for c in range(3)])
            vp.check(dev.lib.vp_cvt_color_dev(dev.ctx.handle, getattr(vp, code), src, w * 3, w, h, d_dst, arr if planes else None), dev.ctx.handle)
            if want_dst:
                assert np.array_equal(dev.down(d_dst, exp.shape), exp), (code, "interleaved", planes)
            for c in planes:
                assert np.array_equal(dev.down(d_pl[c], (h, w)), exp[:, :, c]), (code, "plane", c, planes)
        dev.close()
        src = dev.up(img)
    # GRAY2BGR, with and without planes
    g = rng.integers(0, 256, (h, w), dtype=np.uint8)
    d_g, d_dst = dev.up(g), dev.alloc(h * w * 3)
    d_pl = [dev.alloc(h * w) for _ in range(3)]
    vp.check(dev.lib.vp_cvt_color_dev(dev.ctx.handle, vp.GRAY2BGR, d_g, w, w, h, d_dst, (C.c_void_p * 3)(*d_pl)), dev.ctx.handle)
    assert np.array_equal(dev.down(d_dst, (h, w, 3)), np.repeat(g[:, :, None], 3, axis=2))
    assert all(np.array_equal(dev.down(p, (h, w)), g) for p in d_pl)


@pytest.mark.parametrize("flat", [1, 0])
def test_unaligned_and_strided_images_take_the_generic_kernels(vp, oracle, dev, flat):
    """Pointers that are not 16-B aligned (a plane at an odd offset inside a frame's allocation) and rows with padding: same results."""
    w, h = 64, 20
    rng = np.random.default_rng(3)
    wide = rng.integers(0, 256, (h, w + 5, 3), dtype=np.uint8)
    img = wide[:, :w]                                        # rows of 3 * (w + 5) bytes
    assert dev.lib.vp_set_option(dev.ctx.handle, vp.OPT_FLAT_OPS, flat) == 0
    d_wide = dev.up(wide)
    d_dst = dev.alloc(h * w * 3, offset=4)                   # misaligned output
    d_p1 = dev.alloc(h * w, offset=1)
    vp.check(dev.lib.vp_cvt_color_dev(dev.ctx.handle, vp.BGR2LAB, d_wide, (w + 5) * 3, w, h, d_dst, (C.c_void_p * 3)(None, d_p1, None)), dev.ctx.handle)
    exp = oracle.bgr2lab(np.ascontiguousarray(img))
    assert np.array_equal(dev.down(d_dst, (h, w, 3)), exp) and np.array_equal(dev.down(d_p1, (h, w)), exp[:, :, 1])
    d_src1 = dev.up(np.ascontiguousarray(img), offset=3)     # misaligned packed input
    d_out = dev.alloc(h * w * 3)
    vp.check(dev.lib.vp_cvt_color_dev(dev.ctx.handle, vp.BGR2HSV, d_src1, w * 3, w, h, d_out, None), dev.ctx.handle)
    assert np.array_equal(dev.down(d_out, (h, w, 3)), oracle.bgr2hsv(np.ascontiguousarray(img)))


@pytest.mark.parametrize("flat", [1, 0])
@pytest.mark.parametrize("size", [(1920, 1080), (257, 101), (5, 3)])
def test_inrange_both_forms(vp, oracle, dev, flat, size):
    w, h = size
    rng = np.random.default_rng(w + h)
    assert dev.lib.vp_set_option(dev.ctx.handle, vp.OPT_FLAT_OPS, flat) == 0
    c1 = rng.integers(0, 256, (h, w), dtype=np.uint8)
    c3 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    d1, d3, d_out = dev.up(c1), dev.up(c3), dev.alloc(h * w)
    i32 = lambda *v: (C.c_int32 * 3)(*v)                       # noqa: E731
    for lo, hi in ((100, 200), (0, 255), (0, 0), (255, 255), (200, 100), (-5, 300), (17, 17)):
        vp.check(dev.lib.vp_inrange_u8_dev(dev.ctx.handle, d1, w, w, h, 1, i32(lo, 0, 0), i32(hi, 0, 0), d_out), dev.ctx.handle)
        assert np.array_equal(dev.down(d_out, (h, w)), oracle.inrange(c1, lo, hi)), (lo, hi)
    for lo, hi in (((10, 20, 60), (30, 100, 255)), ((0, 0, 0), (255, 255, 255)), ((50, 200, 0), (60, 100, 255)), ((-1, -1, -1), (256, 256, 256)),
                   ((128, 0, 0), (128, 255, 255))):
        vp.check(dev.lib.vp_inrange_u8_dev(dev.ctx.handle, d3, w * 3, w, h, 3, i32(*lo), i32(*hi), d_out), dev.ctx.handle)
        assert np.array_equal(dev.down(d_out, (h, w)), oracle.inrange(c3, lo, hi)), (lo, hi)


@pytest.mark.parametrize("n", [1920 * 1080 * 3, 16, 15, 1000003])
def test_add_weighted_sixteen_bytes_per_lane(vp, dev, n):
    """modules/bins.py:20 (cv2.addWeighted): saturate(round-half-even(a * alpha + b * beta + gamma)) in correctly rounded doubles - the
    numpy statement - for aligned images, a ragged tail and unaligned pointers."""
    rng = np.random.default_rng(n % 1000)
    a, b = rng.integers(0, 256, n, dtype=np.uint8), rng.integers(0, 256, n, dtype=np.uint8)
    for (alpha, beta, gamma), off in (((0.7, 0.3, 0.0), 0), ((0.5, 0.5, 0.5), 0), ((1.5, -0.25, 3.0), 0), ((0.7, 0.3, 0.0), 1)):
        d_a, d_b, d_o = dev.up(a, off), dev.up(b, off), dev.alloc(n, off)
        vp.check(dev.lib.vp_add_weighted_u8_dev(dev.ctx.handle, d_a, C.c_double(alpha), d_b, C.c_double(beta), C.c_double(gamma), n, d_o), dev.ctx.handle)
        exp = np.clip(np.rint(a.astype(np.float64) * alpha + b.astype(np.float64) * beta + gamma), 0, 255).astype(np.uint8)
        assert np.array_equal(dev.down(d_o, (n,)), exp), (alpha, beta, gamma, off)
        dev.close()
