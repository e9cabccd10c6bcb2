"""Debug overlays (utils/draw.py:283-327 draw_contours / draw_polylines): the native rasteriser (libvp vp_draw_polylines_u8: stamps
collected in a coverage bit plane, image written once) must paint exactly the pixels of the statement-by-statement Python rasteriser
of vision/utils/draw.py (Bresenham steps, square brush).  Host code only: runs without a GPU."""
import numpy as np
import pytest

from vision.utils import draw as D


def _python(mat, polys, closed, color, thickness):
    for p in polys:
        pts = np.asarray(p, np.int64).reshape(-1, 2)
        if len(pts) == 1:
            D._line(mat, pts[0], pts[0], color, thickness)
        for i in range(len(pts) if closed else len(pts) - 1):
            D._line(mat, pts[i], pts[(i + 1) % len(pts)], color, thickness)


def _ring(cx, cy, r, n):
    t = np.linspace(0, 2 * np.pi, n, endpoint=False)
    return np.stack([cx + r * np.cos(t), cy + r * np.sin(t)], 1).round().astype(np.int32).reshape(-1, 1, 2)


@pytest.mark.parametrize("thickness", [1, 2, 3, 4, 10, 17, 31, 32, 33, 40])
@pytest.mark.parametrize("closed", [True, False])
def test_native_equals_python(thickness, closed):
    rng = np.random.default_rng(thickness)
    polys = [_ring(60, 50, 30, 200), _ring(150, 40, 55, 12),            # one-pixel steps; long straight edges, partly outside
             np.array([[5, 5], [190, 5], [190, 6], [191, 7], [20, 90]], np.int32),          # horizontal run, neighbours, a long diagonal
             np.array([[100, 100]], np.int32), np.array([[-30, 20], [230, 60]], np.int32),  # a single point; both ends outside
             rng.integers(-20, 220, (9, 2)).astype(np.int32), np.array([[63, 10], [64, 10], [65, 10], [127, 11], [128, 12]], np.int32)]
    for shape, color in (((110, 200, 3), (7, 200, 255)), ((110, 200), (180,)), ((110, 200, 4), (1, 2, 3, 4))):
        base = rng.integers(0, 255, shape).astype(np.uint8)
        a, b = base.copy(), base.copy()
        assert D._native_polylines(a, polys, closed, color, thickness)
        col = np.asarray(color, np.uint8) if len(shape) == 3 else np.uint8(color[0])
        _python(b, polys, closed, col, thickness)
        assert np.array_equal(a, b), (shape, thickness, closed)
    # a second call on the same thread starts from a clean plane
    c = np.zeros((110, 200, 3), np.uint8)
    assert D._native_polylines(c, [np.array([[10, 10], [12, 10]], np.int32)], False, (9, 9, 9), 1)
    assert int((c > 0).any(2).sum()) == 3


def test_native_on_a_view_and_wide_words():
    """Rows wider than one 64-bit word of the plane, an image that is a view with a row stride, full-width runs."""
    big = np.zeros((40, 400, 3), np.uint8)
    view = big[4:36, 20:330]                                       # 310 px wide: five words, the last one partial
    ref = np.zeros((32, 310, 3), np.uint8)
    polys = [np.array([[-5, 3], [400, 3]], np.int32), np.array([[0, 31], [309, 0]], np.int32), _ring(150, 16, 14, 90), np.array([[309, 0], [309, 31]], np.int32)]
    assert D._native_polylines(view, polys, False, (1, 2, 3), 5)
    _python(ref, polys, False, np.asarray((1, 2, 3), np.uint8), 5)
    assert np.array_equal(view, ref)
    assert not big[:4].any() and not big[36:].any() and not big[:, :20].any() and not big[:, 330:].any()


def test_draw_contours_entry():
    img = np.zeros((120, 160, 3), np.uint8)
    ref = img.copy()
    cs = [_ring(50, 60, 30, 150), _ring(110, 50, 20, 100)]
    D.draw_contours(img, cs, thickness=10)
    _python(ref, cs, True, np.asarray((0, 0, 255), np.uint8), 10)
    assert np.array_equal(img, ref) and img.any()


def test_bresenham_steps_have_a_closed_form():
    """The device overlay kernel (csrc/vp_morph.hip k_draw_segments) takes the steps of a segment in parallel: after i steps of the host
    rasteriser's loop (utils/draw.py _line / vp_draw_polylines_u8) the longer axis has advanced i and the shorter one stands at
    floor((2 i m + M) / (2 M)), dx >= |dy| counting as x-major.  Checked here against the loop itself."""
    def loop(x0, y0, x1, y1):
        dx, dy = abs(x1 - x0), -abs(y1 - y0)
        sx, sy = (1 if x0 < x1 else -1), (1 if y0 < y1 else -1)
        err, out = dx + dy, []
        while True:
            out.append((x0, y0))
            if x0 == x1 and y0 == y1:
                return out
            e2 = 2 * err
            if e2 >= dy:
                err += dy
                x0 += sx
            if e2 <= dx:
                err += dx
                y0 += sy

    def closed(x0, y0, x1, y1):
        dx, ady = abs(x1 - x0), abs(y1 - y0)
        sx, sy = (1 if x0 < x1 else -1), (1 if y0 < y1 else -1)
        n = max(dx, ady)
        if n == 0:
            return [(x0, y0)]
        if dx >= ady:
            return [(x0 + sx * i, y0 + sy * ((2 * i * ady + dx) // (2 * dx))) for i in range(n + 1)]
        return [(x0 + sx * ((2 * i * dx + ady) // (2 * ady)), y0 + sy * i) for i in range(n + 1)]

    for x1 in range(-40, 41):
        for y1 in range(-40, 41):
            assert loop(3, -2, x1, y1) == closed(3, -2, x1, y1), (x1, y1)
    rng = np.random.default_rng(5)
    for a in rng.integers(-4000, 4000, (400, 4)).tolist():
        assert loop(*a) == closed(*a), a
