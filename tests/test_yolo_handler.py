"""Config 5 pieces that need no GPU: the records and their mapping (`vision.yolo.data`), the torpedo-board handler's arithmetic
(reference handlers/torpedoes.py:21-209: corner order, normalisation, centre, area as a fraction of the image, thresholds), the
tracker stand-in, the letterbox geometry and the network's shapes."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims"))


def test_summary_entries_become_records():
    from vision.yolo.data import MAP_FN, OBBData, PoseData, YOLOData
    e = {"name": "torpedo_board", "class": 0, "confidence": 0.8, "box": {"x1": 1, "y1": 2, "x2": 3, "y2": 4, "x3": 5, "y3": 6, "x4": 7, "y4": 8}, "track_id": 9}
    d = MAP_FN["obb"](e)
    assert isinstance(d, OBBData) and (d.name, d.confidence, d.x1, d.y1, d.x4, d.y4, d.track_id) == ("torpedo_board", 0.8, 1, 2, 7, 8, 9)
    d = MAP_FN["detect"]({"name": "bin", "class": 10, "confidence": 0.5, "box": {"x1": 1, "y1": 2, "x2": 3, "y2": 4}})
    assert isinstance(d, YOLOData) and d.track_id == -1 and d.y2 == 4
    d = MAP_FN["pose"]({"name": "p", "class": 0, "confidence": 0.5, "box": {"x1": 1, "y1": 2, "x2": 3, "y2": 4}, "keypoints": {"x": [1.0], "y": [2.0], "visible": [0.9]}})
    assert isinstance(d, PoseData) and d.keypoints_visible == [0.9]


class _Parent:
    """What HandlerBase borrows from its module (core/handlers.py:47-58)."""

    def __init__(self, shape, threshold=0.1):
        self.shape = shape
        self.tuners = {"torpedo_threshold": threshold}
        self.posted = {}
        self._loop = None

    def normalize(self, c):                      # core/base.py:553-574: (y - h/2)/w, (x - w/2)/w
        h, w = self.shape[:2]
        return (c[0] - h / 2) / w, (c[1] - w / 2) / w

    def normalize_axis(self, c, axis):
        return self.normalize((c, c))[axis]

    def post(self, name, image, color_space="BGR"):
        self.posted[name] = np.array(image, copy=True)

    def get_latency(self):
        return 0


def test_torpedoes_handler_outputs():
    import shm
    from vision.handlers.torpedoes import TorpedoesOBB
    from vision.yolo.data import OBBData
    H, W = 360, 640
    parent = _Parent((H, W, 3))
    h = TorpedoesOBB("torpedoes")
    h.register(parent)
    img = np.zeros((H, W, 3), np.uint8)
    # a rotated board (corners given in no particular order), two candidates: the more confident one wins
    board = [OBBData("torpedo_board", 0.4, 10, 10, 20, 10, 20, 20, 10, 20), OBBData("torpedo_board", 0.9, 300, 100, 500, 120, 480, 300, 280, 280)]
    shark = [OBBData("shark_hole", 0.05, 1, 1, 2, 1, 2, 2, 1, 2)]                       # below the threshold: not visible
    saw = [OBBData("saw_hole", 0.6, 400, 150, 440, 150, 440, 200, 400, 200)]
    h.process("forward", img, board, shark, saw)
    g = shm.yolo_torpedoes_board.get()
    assert g.board_visible == 1 and g.board_confidence == 0.9 and g.shark_visible == 0 and g.saw_visible == 1 and g.saw_confidence == 0.6
    # corners by sum / difference of coordinates: tl = (300,100), br = (480,300), tr = (500,120), bl = (280,280)
    exp = {"top_left": (300, 100), "top_right": (500, 120), "bottom_left": (280, 280), "bottom_right": (480, 300)}
    for name, (x, y) in exp.items():
        ny, nx = parent.normalize((y, x))
        assert getattr(g, f"board_{name}_x") == nx and getattr(g, f"board_{name}_y") == ny, name
    n = {k: parent.normalize((y, x)) for k, (x, y) in exp.items()}
    assert g.board_center_y == (n["bottom_right"][0] + n["bottom_left"][0] + n["top_right"][0] + n["top_left"][0]) / 4
    assert g.board_center_x == (n["bottom_right"][1] + n["bottom_left"][1] + n["top_right"][1] + n["top_left"][1]) / 4
    # area: shoelace over (br, tr, tl, bl) in normalised coordinates, times width / height; cross-check with pixel area / (W * H)
    px = np.array([exp["bottom_right"], exp["top_right"], exp["top_left"], exp["bottom_left"]], np.float64)
    pix_area = abs(sum(px[i][0] * px[(i + 1) % 4][1] - px[(i + 1) % 4][0] * px[i][1] for i in range(4))) / 2
    assert abs(g.board_area - pix_area / (W * H)) < 1e-12
    assert abs(g.saw_area - (40 * 50) / (W * H)) < 1e-12 and abs(g.saw_center_x - parent.normalize((175, 420))[1]) < 1e-15
    assert shm.relay.point_x.get() == ((n["top_left"][1] + n["bottom_left"][1]) / 2 + (n["top_right"][1] + n["bottom_right"][1]) / 2) / 2
    # outlines were drawn into the image that is posted: lime for the board, red for the saw hole, nothing for the shark hole
    from vision.utils.draw import Color
    out = parent.posted["torpedoes handler"]
    assert tuple(out[100, 300]) == Color.LIME() and tuple(out[150, 420]) == Color.RED() and not (out == np.array(Color.BLUE(), np.uint8)).all(axis=2).any()
    # nothing detected: everything invisible, image posted untouched
    h.process("forward", np.zeros((H, W, 3), np.uint8), [], [], [])
    g = shm.yolo_torpedoes_board.get()
    assert (g.board_visible, g.shark_visible, g.saw_visible) == (0, 0, 0) and parent.posted["torpedoes handler"].max() == 0


def test_tracker_and_letterbox_geometry():
    from vision.yolo.engine import Tracker, letterbox_shape
    assert letterbox_shape(1080, 1920) == (384, 640) and letterbox_shape(640, 640) == (640, 640) and letterbox_shape(1920, 1080) == (640, 384)
    t = Tracker()
    assert t.update([0, 1], [(0, 0, 10, 10), (20, 20, 30, 30)]) == [1, 2]
    assert t.update([1, 0, 0], [(21, 21, 31, 31), (1, 1, 11, 11), (100, 100, 110, 110)]) == [2, 1, 3]      # same objects keep their ids
    assert t.update([0], [(200, 200, 210, 210)]) == [4]


def test_network_shapes_and_box_helpers():
    import torch
    from vision.yolo.engine import regularize, xywhr_to_corners
    from vision.yolo.model import YOLOv8nOBB
    m = YOLOv8nOBB(15).eval()
    assert 3.0e6 < sum(p.numel() for p in m.parameters()) < 3.2e6            # YOLOv8n-obb: 3.1 M parameters
    with torch.no_grad():
        y = m(torch.zeros(1, 3, 96, 160))
    assert y.shape == (1, 4 + 15 + 1, 12 * 20 + 6 * 10 + 3 * 5)
    assert float(y[0, 4:19].min()) >= 0 and float(y[0, 4:19].max()) <= 1 and float(y[0, 19].min()) >= -np.pi / 4 - 1e-6
    b = torch.tensor([[10.0, 20.0, 4.0, 8.0, 0.3], [0.0, 0.0, 6.0, 2.0, -0.2]])
    r = regularize(b)
    assert torch.allclose(r[0], torch.tensor([10.0, 20.0, 8.0, 4.0, 0.3 + np.pi / 2])) and torch.allclose(r[1], torch.tensor([0.0, 0.0, 6.0, 2.0, np.pi - 0.2]))
    ca, cb = xywhr_to_corners(b), xywhr_to_corners(r)                         # the same rectangles: the same corner sets
    for k in range(2):
        sa = sorted((round(float(x), 4), round(float(y), 4)) for x, y in ca[k])
        sb = sorted((round(float(x), 4), round(float(y), 4)) for x, y in cb[k])
        assert sa == sb
    sq = xywhr_to_corners(torch.tensor([[5.0, 5.0, 4.0, 2.0, 0.0]]))[0].tolist()
    assert sq == [[7.0, 6.0], [7.0, 4.0], [3.0, 4.0], [3.0, 6.0]]
