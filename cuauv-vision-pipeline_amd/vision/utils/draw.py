"""Minimal stand-in for the reference utils/draw.py (debug overlays; not part of the accelerated path).
Only what the in-scope modules call: `draw_contours` (modules/red_buoy.py:39) and `draw_polylines`, as plain numpy
rasterisation (lines between consecutive contour points, square brush of the requested thickness; -1 fills by even-odd
scanline).  Pixel-exact agreement with cv2's anti-alias-free line drawing is not claimed."""
from typing import List, Tuple

import numpy as np


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


def draw_polylines(mat: np.ndarray, points, isClosed: bool = False, color: Tuple[int, int, int] = (0, 0, 255), thickness: int = 1) -> None:
    """utils/draw.py:304-327; modifies `mat` in place."""
    pts = np.asarray(points, np.int64).reshape(-1, 2)
    if len(pts) == 0:
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
    for c in contours:
        draw_polylines(mat, c, True, color, thickness)
