"""Minimal stand-in for the reference utils/draw.py (debug overlays; not part of the accelerated path).
Only what the in-scope modules call: `draw_contours` (modules/red_buoy.py:39) and `draw_polylines`, as plain numpy
rasterisation (lines between consecutive contour points, square brush of the requested thickness; -1 fills by even-odd
scanline).  Pixel-exact agreement with cv2's anti-alias-free line drawing is not claimed."""
from enum import Enum
from typing import List, Tuple

import numpy as np


# The palette handlers name their overlay colours from (reference utils/draw.py:9-37: the names and BGR values are what
# `Color.LIME`, `get_color("teal")` ... must keep meaning); kept as data, the enum is made from it.
_PALETTE_BGR = """
    RED 75 25 230 | GREEN 75 180 60 | YELLOW 0 225 255 | BLUE 200 130 0 | ORANGE 48 130 245 | PURPLE 180 30 145 | CYAN 240 240 70
    MAGENTA 230 50 240 | LIME 60 245 210 | PINK 212 190 250 | TEAL 128 128 0 | LAVENDER 255 190 220 | BROWN 40 110 170
    BEIGE 200 250 255 | MAROON 0 0 128 | MINT 195 255 170 | OLIVE 0 128 128 | APRICOT 180 215 255 | NAVY 128 0 0 | GREY 128 128 128
    WHITE 255 255 255 | BLACK 0 0 0 | HOTPINK 180 105 255 | DEEPPINK 147 20 255 | FUCHSIA 255 0 255
"""


class _Callable:
    def __call__(self):                      # `Color.LIME()` gives the tuple, as `.value` does
        return self.value


Color = Enum("Color", [(f[0], tuple(int(v) for v in f[1:])) for f in (e.split() for e in _PALETTE_BGR.replace("\n", "|").split("|")) if f],
             type=_Callable, module=__name__)
Color.__doc__ = "Named BGR colours for overlays."


def get_color(color_name: str):
    """Palette entry by (case-insensitive) name; ValueError for a name the palette does not hold."""
    member = Color.__members__.get(str(color_name).upper())
    if member is None:
        raise ValueError(f"{color_name} is not a valid color name")
    return member.value


def _stamp(mat, x, y, color, r0, r1):
    h, w = mat.shape[:2]
    xa, xb = max(x - r0, 0), min(x + r1 + 1, w)
    ya, yb = max(y - r0, 0), min(y + r1 + 1, h)
    if xa < xb and ya < yb:
        mat[ya:yb, xa:xb] = color


def _line(mat, p0, p1, color, thickness):
    x0, y0 = int(p0[0]), int(p0[1])
    x1, y1 = int(p1[0]), int(p1[1])
    r0, r1 = (thickness - 1) // 2, thickness // 2
    dx, dy = abs(x1 - x0), -abs(y1 - y0)
    sx, sy = (1 if x0 < x1 else -1), (1 if y0 < y1 else -1)
    err = dx + dy
    while True:
        _stamp(mat, x0, y0, color, r0, r1)
        if x0 == x1 and y0 == y1:
            break
        e2 = 2 * err
        if e2 >= dy:
            err += dy
            x0 += sx
        if e2 <= dx:
            err += dx
            y0 += sy


def _fill(mat, pts, color):
    h, w = mat.shape[:2]
    n = len(pts)
    ys = pts[:, 1]
    for y in range(max(int(ys.min()), 0), min(int(ys.max()), h - 1) + 1):
        xs = []
        for i in range(n):
            (xa, ya), (xb, yb) = pts[i], pts[(i + 1) % n]
            if ya == yb:
                continue
            if min(ya, yb) <= y < max(ya, yb):
                xs.append(xa + (y - ya) * (xb - xa) / (yb - ya))
        xs.sort()
        for a, b in zip(xs[0::2], xs[1::2]):
            xa, xb = max(int(np.ceil(a)), 0), min(int(np.floor(b)), w - 1)
            if xa <= xb:
                mat[y, xa:xb + 1] = color


def _native_polylines(mat, polys, closed, color, thickness):
    """The same rasteriser in C (libvp vp_draw_polylines_u8, host code), all polylines in one call: returns False when the image
    cannot be handed over as is."""
    if not (isinstance(mat, np.ndarray) and mat.dtype == np.uint8 and mat.ndim in (2, 3) and mat.flags.writeable):
        return False
    cn = 1 if mat.ndim == 2 else mat.shape[2]
    if cn > 4 or mat.strides[-1] != 1 or (mat.ndim == 3 and mat.strides[1] != cn) or mat.strides[0] < mat.shape[1] * cn:
        return False
    try:
        from vision import _vp
        lib = _vp.lib()
    except Exception:
        return False
    flat = getattr(polys, "_flat", None)
    if flat is not None:                            # the tuple find_contours returned: its arrays are views of this one block
        if len(polys) == 0:
            return True
        counts, p32 = polys._counts, flat
    else:
        polys = [np.asarray(p).reshape(-1, 2) for p in polys]
        if not polys:
            return True
        counts = np.fromiter((len(p) for p in polys), np.int32, len(polys))
        p32 = np.ascontiguousarray(np.concatenate(polys) if len(polys) > 1 else polys[0], np.int32)
    col = np.zeros(4, np.uint8)
    col[:cn] = np.asarray(color, np.uint8).ravel()[:cn] if np.ndim(color) else np.uint8(color)
    return lib.vp_draw_polylines_u8(mat.ctypes.data, mat.strides[0], mat.shape[1], mat.shape[0], cn, p32.ctypes.data, counts.ctypes.data,
                                    len(polys), int(bool(closed)), col.ctypes.data, int(thickness)) == 0


def _device_polylines(mat, polys, closed, color, thickness):
    """The same polylines drawn by the device into an image that lives there and has no host copy (libvp vp_draw_polylines_dev):
    returns False when the image is not of that kind - then it is drawn on the host, as before."""
    from vision.devmat import DeviceMat
    if not isinstance(mat, DeviceMat) or mat.dtype != np.uint8 or mat.ndim not in (2, 3) or mat._host is not None or thickness > 255:
        return False
    from vision import _vp
    ctx = _vp.default_context()
    cn = 1 if mat.ndim == 2 else mat.shape[2]
    if cn > 4 or not mat.device_valid_for(ctx):
        return False
    flat = getattr(polys, "_flat", None)
    if flat is not None:
        if len(polys) == 0:
            return True
        counts, p32 = polys._counts, flat
    else:
        polys = [np.asarray(p).reshape(-1, 2) for p in polys]
        if not polys:
            return True
        counts = np.fromiter((len(p) for p in polys), np.int32, len(polys))
        p32 = np.ascontiguousarray(np.concatenate(polys) if len(polys) > 1 else polys[0], np.int32)
    col = np.zeros(4, np.uint8)
    col[:cn] = np.asarray(color, np.uint8).ravel()[:cn] if np.ndim(color) else np.uint8(color)
    mat._before_write()                                  # operators that were deferred on this image read it as it is now
    _vp.check(_vp.lib().vp_draw_polylines_dev(ctx.handle, mat.dev_ptr, mat.shape[1], mat.shape[0], cn, p32.ctypes.data, counts.ctypes.data, len(counts),
                                              int(bool(closed)), col.ctypes.data, int(thickness)), ctx.handle)
    mat.binary = False
    return True


def _native_polyline(mat, pts, closed, color, thickness):
    return _native_polylines(mat, [pts], closed, color, thickness)


def draw_polylines(mat: np.ndarray, points, isClosed: bool = False, color: Tuple[int, int, int] = (0, 0, 255), thickness: int = 1) -> None:
    """utils/draw.py:304-327; modifies `mat` in place."""
    from vision.devmat import to_host
    pts = np.asarray(points, np.int64).reshape(-1, 2)
    if len(pts) == 0:
        return
    if thickness >= 0 and _device_polylines(mat, [pts], isClosed, color, max(thickness, 1)):
        return
    mat = to_host(mat)
    if thickness >= 0 and _native_polyline(mat, pts, isClosed, color, max(thickness, 1)):
        return
    color = np.asarray(color, mat.dtype)[: (mat.shape[2] if mat.ndim == 3 else 1)]
    if mat.ndim == 2:
        color = color[0]
    if thickness < 0:
        _fill(mat, pts, color)
        thickness = 1
    last = len(pts) if isClosed else len(pts) - 1
    if len(pts) == 1:
        _line(mat, pts[0], pts[0], color, max(thickness, 1))
    for i in range(last):
        _line(mat, pts[i], pts[(i + 1) % len(pts)], color, max(thickness, 1))


def draw_contours(mat: np.ndarray, contours: List[np.ndarray], color: Tuple[int, int, int] = (0, 0, 255), thickness: int = 1) -> None:
    """utils/draw.py:283-301 (cv2.drawContours(mat, contours, -1, color, thickness)); modifies `mat` in place."""
    from vision.devmat import to_host
    if thickness >= 0 and _device_polylines(mat, contours, True, color, max(thickness, 1)):
        return
    mat = to_host(mat)
    if thickness >= 0 and _native_polylines(mat, contours, True, color, max(thickness, 1)):
        return
    for c in contours:
        draw_polylines(mat, c, True, color, thickness)


def draw_line(mat: np.ndarray, pt1: Tuple[int, int], pt2: Tuple[int, int], color: Tuple[int, int, int] = (0, 0, 255), thickness: int = 1) -> None:
    """utils/draw.py:101-121 (cv2.line): in place."""
    _line(mat, pt1, pt2, color, max(int(thickness), 1))


def draw_rect(mat: np.ndarray, pt1: Tuple[int, int], pt2: Tuple[int, int], color: Tuple[int, int, int] = (0, 0, 255), thickness: int = 1) -> None:
    """utils/draw.py:147-168 (cv2.rectangle): in place; negative thickness fills."""
    (x0, y0), (x1, y1) = (int(pt1[0]), int(pt1[1])), (int(pt2[0]), int(pt2[1]))
    if thickness < 0:
        h, w = mat.shape[:2]
        xa, xb = max(min(x0, x1), 0), min(max(x0, x1), w - 1)
        ya, yb = max(min(y0, y1), 0), min(max(y0, y1), h - 1)
        if xa <= xb and ya <= yb:
            mat[ya:yb + 1, xa:xb + 1] = color
        return
    for a, b in (((x0, y0), (x1, y0)), ((x1, y0), (x1, y1)), ((x1, y1), (x0, y1)), ((x0, y1), (x0, y0))):
        _line(mat, a, b, color, max(int(thickness), 1))


def draw_circle(mat: np.ndarray, center: Tuple[int, int], radius: int, color: Tuple[int, int, int] = (0, 0, 255), thickness: int = 1) -> None:
    """utils/draw.py:51-72 (cv2.circle): in place; negative thickness fills.  Midpoint rasterisation."""
    cx, cy, r = int(center[0]), int(center[1]), int(radius)
    h, w = mat.shape[:2]
    if r < 0:
        return
    if thickness < 0:
        for dy in range(-r, r + 1):
            y = cy + dy
            if 0 <= y < h:
                dx = int(np.floor(np.sqrt(r * r - dy * dy)))
                xa, xb = max(cx - dx, 0), min(cx + dx, w - 1)
                if xa <= xb:
                    mat[y, xa:xb + 1] = color
        return
    t = max(int(thickness), 1)
    r0, r1 = (t - 1) // 2, t // 2
    x, y, d = r, 0, 1 - r
    while x >= y:
        for px, py in ((x, y), (y, x), (-y, x), (-x, y), (-x, -y), (-y, -x), (y, -x), (x, -y)):
            _stamp(mat, cx + px, cy + py, color, r0, r1)
        y += 1
        if d < 0:
            d += 2 * y + 1
        else:
            x -= 1
            d += 2 * (y - x) + 1


def draw_text(mat: np.ndarray, s: str, origin: Tuple[int, int], scale: float, color: Tuple[int, int, int] = (0, 0, 255), thickness: int = 1) -> None:
    """utils/draw.py:218-242 (cv2.putText with the Hershey simplex font).  Glyph rendering is not reproduced: the stand-in marks the
    text's baseline with a line of the width the label would take (about 20 * scale pixels per character), so overlays stay
    legible as "something was labelled here" and handlers that annotate their debug images run unchanged."""
    x, y = int(origin[0]), int(origin[1])
    _line(mat, (x, y), (x + int(round(20 * float(scale) * len(str(s)))), y), color, max(int(thickness), 1))
