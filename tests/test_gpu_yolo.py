"""Detector pre / post-processing kernels (BASELINE config 5) against plain PyTorch / numpy fp32 restatements of the published
algorithms (LetterBox + preprocess, greedy NMS, probabilistic-IoU rotated NMS).  The third-party package the reference calls
(modules/yolo.py:112) is absent, so parity with it is unpinned; these tests pin the kernels to the restatements."""
import numpy as np
import pytest
import torch

import frames as F

pytestmark = pytest.mark.gpu


def _resize_linear_u8(src, nw, nh):
    """cv2.resize(src, (nw, nh), interpolation=INTER_LINEAR) for uint8, OpenCV's 11-bit fixed-point generic path."""
    h, w = src.shape[:2]
    if w == 2 * nw and h == 2 * nh:          # OpenCV substitutes the 2x2 box average at an exact halving
        s = src.astype(np.int64)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def coef(n, ssize, scale):
        d = np.arange(n)
        f = ((d + 0.5) * np.float64(np.float32(scale)) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        lo = s < 0
        s[lo], f[lo] = 0, 0
        hi = s >= ssize - 1
        s[hi], f[hi] = ssize - 1, 0
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        return s, np.minimum(s + 1, ssize - 1), a0, a1
    sx, sx1, ax0, ax1 = coef(nw, w, w / nw)
    sy, sy1, ay0, ay1 = coef(nh, h, h / nh)
    s = src.astype(np.int64)
    rows0 = s[sy][:, sx] * ax0[None, :, None] + s[sy][:, sx1] * ax1[None, :, None]
    rows1 = s[sy1][:, sx] * ax0[None, :, None] + s[sy1][:, sx1] * ax1[None, :, None]
    out = (((ay0[:, None, None] * (rows0 >> 4)) >> 16) + ((ay1[:, None, None] * (rows1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def _letterbox_ref(img, H, W, pad=114):
    h, w = img.shape[:2]
    r = min(H / h, W / w)
    nw, nh = min(max(1, int(round(w * r))), W), min(max(1, int(round(h * r))), H)
    left, top = int(round((W - nw) / 2 - 0.1)), int(round((H - nh) / 2 - 0.1))
    res = img if (nw, nh) == (w, h) else _resize_linear_u8(img, nw, nh)
    canvas = np.full((H, W, 3), pad, np.uint8)
    canvas[top:top + nh, left:left + nw] = res
    chw = np.ascontiguousarray(canvas[:, :, ::-1].transpose(2, 0, 1)).astype(np.float32) / np.float32(255)
    return chw, (r, left, top), (nw, nh)


@pytest.mark.parametrize("shape,new", [((1080, 1920), (640, 640)), ((360, 640), (640, 640)), ((480, 480), (640, 640)), ((97, 333), (320, 256)),
                                       ((640, 640), (640, 640)), ((700, 300), (64, 96)), ((720, 1280), (640, 640))])
def test_letterbox(vp, shape, new):
    from vision.yolo import letterbox
    img = F.s1_buoy(1, shape[1], shape[0])
    got, geom = letterbox(img, new)
    exp, (r, left, top), (nw, nh) = _letterbox_ref(img, new[0], new[1])
    assert got.shape == (3, new[0], new[1]) and got.dtype == np.float32
    assert abs(geom[0] - r) < 1e-6 and geom[1] == left and geom[2] == top
    assert np.array_equal(got, exp)                              # same fixed-point arithmetic as the restatement
    # independent witness: float bilinear interpolation with half-pixel centres (torch), within the 8-bit rounding of the resize
    t = torch.from_numpy(np.ascontiguousarray(img[:, :, ::-1].transpose(2, 0, 1)).astype(np.float32))[None]
    ref = torch.nn.functional.interpolate(t, size=(nh, nw), mode="bilinear", align_corners=False)[0].numpy() / 255.0
    inner = got[:, top:top + nh, left:left + nw]
    assert np.abs(inner - ref).max() <= 1.01 / 255
    pad = np.ones((new[0], new[1]), bool)
    pad[top:top + nh, left:left + nw] = False
    assert (got[:, pad] == np.float32(114) / np.float32(255)).all()
    # device entry: torch tensor in, torch tensor out, same values
    g2, geom2 = letterbox(torch.from_numpy(img).cuda(), new)
    assert g2.is_cuda and np.array_equal(g2.cpu().numpy(), got) and geom2 == geom


def _iou(a, b):
    ix = np.maximum(np.float32(0), np.minimum(a[2], b[:, 2]) - np.maximum(a[0], b[:, 0]))
    iy = np.maximum(np.float32(0), np.minimum(a[3], b[:, 3]) - np.maximum(a[1], b[:, 1]))
    inter = ix * iy
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter)


def _nms_ref(boxes, scores, thr, max_det):
    order = np.lexsort((np.arange(len(scores)), -scores))          # score descending, ties by index
    alive = np.ones(len(order), bool)
    keep = []
    for pos, i in enumerate(order):
        if not alive[pos]:
            continue
        keep.append(i)
        if len(keep) == max_det:
            break
        rest = order[pos + 1:]
        alive[pos + 1:] &= ~(_iou(boxes[i], boxes[rest]) > np.float32(thr))
    return np.array(keep, np.int64)


def _boxes(rng, n, size=640):
    c = rng.uniform(0, size, (n, 2)).astype(np.float32)
    wh = rng.uniform(4, 200, (n, 2)).astype(np.float32)
    return np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 300, 2000, 8400])
def test_nms_greedy(vp, n):
    from vision.yolo import nms
    rng = np.random.default_rng(n)
    boxes = _boxes(rng, n)
    scores = rng.random(n).astype(np.float32)
    if n > 10:
        scores[5] = scores[3]                                      # a tie
        boxes[7] = boxes[2]                                        # exact duplicates
    for thr in (0.3, 0.45, 0.7):
        got = nms(boxes, scores, thr, max_det=300)
        exp = _nms_ref(boxes, scores, thr, 300)
        assert np.array_equal(got, exp), (n, thr)
    if n:
        assert np.array_equal(nms(boxes, scores, 0.45, max_det=3), _nms_ref(boxes, scores, 0.45, 3))
        g = nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.45, max_det=300)
        assert g.is_cuda and np.array_equal(g.cpu().numpy(), _nms_ref(boxes, scores, 0.45, 300))


def _probiou(b1, b2, eps=1e-7):
    def cov(b):
        a, bb, c = b[:, 2:3] ** 2 / 12, b[:, 3:4] ** 2 / 12, b[:, 4:5]
        cos, sin = c.cos(), c.sin()
        return a * cos ** 2 + bb * sin ** 2, a * sin ** 2 + bb * cos ** 2, (a - bb) * cos * sin
    x1, y1 = b1[:, 0:1], b1[:, 1:2]
    x2, y2 = b2[:, 0][None], b2[:, 1][None]
    a1, bb1, c1 = cov(b1)
    a2, bb2, c2 = (t[:, 0][None] for t in cov(b2))
    den = (a1 + a2) * (bb1 + bb2) - (c1 + c2) ** 2
    t1 = (((a1 + a2) * (y1 - y2) ** 2 + (bb1 + bb2) * (x1 - x2) ** 2) / (den + eps)) * 0.25
    t2 = (((c1 + c2) * (x2 - x1) * (y1 - y2)) / (den + eps)) * 0.5
    t3 = (den / (4 * ((a1 * bb1 - c1 ** 2).clamp(min=0) * (a2 * bb2 - c2 ** 2).clamp(min=0)).sqrt() + eps) + eps).log() * 0.5
    bd = (t1 + t2 + t3).clamp(eps, 100.0)
    return 1 - (1.0 - (-bd).exp() + eps).sqrt()


@pytest.mark.parametrize("n", [1, 50, 700, 3000])
def test_nms_rotated(vp, n):
    """Probabilistic IoU + the rule 'drop a box when some higher-scored box overlaps it by >= thr' (torch fp32 restatement).  The
    kernel's float32 ops are the same formula in a different evaluation order, so boxes whose overlap with the decisive
    neighbour lies within 1e-4 of the threshold are allowed to differ."""
    from vision.yolo import nms_rotated
    rng = np.random.default_rng(100 + n)
    xywh = np.concatenate([rng.uniform(0, 640, (n, 2)), rng.uniform(8, 160, (n, 2))], 1)
    boxes = np.concatenate([xywh, rng.uniform(-np.pi / 2, np.pi / 2, (n, 1))], 1).astype(np.float32)
    scores = rng.random(n).astype(np.float32)
    thr = 0.45
    got = set(nms_rotated(boxes, scores, thr, max_det=n).tolist())
    order = np.lexsort((np.arange(n), -scores))
    tb = torch.from_numpy(boxes[order])
    ious = _probiou(tb, tb).triu_(diagonal=1)
    mx = ious.max(dim=0)[0].numpy()
    exp = set(order[mx < thr].tolist())
    unsure = set(order[np.abs(mx - thr) < 1e-4].tolist())
    assert (got ^ exp) <= unsure, (len(got ^ exp), len(unsure))
    kept = nms_rotated(boxes, scores, thr, max_det=n)
    assert list(scores[kept]) == sorted(scores[kept], reverse=True)       # best score first
    if n > 1:
        g = nms_rotated(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), thr, max_det=n)
        assert set(g.cpu().numpy().tolist()) == got


def test_detection_records_and_corner_order(vp):
    from vision.yolo import OBBData, order_points, scale_boxes
    d = OBBData("torpedo_board", 0.9, 10, 5, 110, 8, 108, 60, 12, 58)
    tl, tr, bl, br = order_points([(d.x1, d.y1), (d.x2, d.y2), (d.x3, d.y3), (d.x4, d.y4)])
    assert tl == (10, 5) and tr == (110, 8) and br == (108, 60) and bl == (12, 58)
    b = scale_boxes(np.array([[100, 140, 300, 340]], np.float32), (0.5, 0, 140))
    assert np.allclose(b, [[200, 0, 600, 400]])


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_resize_linear_u8(vp, cn):
    """cv2.resize (facade) == the fixed-point restatement, and within one count of float bilinear interpolation."""
    from vision import cv2_facade as cv2
    rng = np.random.default_rng(cn)
    for (h, w), (dh, dw) in [((90, 160), (45, 80)), ((90, 160), (512, 512)), ((33, 65), (7, 200)), ((64, 64), (64, 64)), ((100, 50), (99, 51))]:
        img = rng.integers(0, 256, (h, w, cn), dtype=np.uint8)
        src = img[:, :, 0] if cn == 1 else img
        got = cv2.resize(src, (dw, dh))
        exp = _resize_linear_u8(img, dw, dh)
        assert got.shape == ((dh, dw) if cn == 1 else (dh, dw, cn))
        assert np.array_equal(got.reshape(dh, dw, cn), exp)
        t = torch.from_numpy(img.transpose(2, 0, 1).astype(np.float32))[None]
        ref = torch.nn.functional.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False)[0].numpy().transpose(1, 2, 0)
        assert np.abs(got.reshape(dh, dw, cn).astype(np.float32) - ref).max() <= 1.01
