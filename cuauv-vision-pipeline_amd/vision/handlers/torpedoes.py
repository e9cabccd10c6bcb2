"""Torpedo-board handler (BASELINE config 5), same interface and outputs as the reference's handlers/torpedoes.py:21-209.

For each of the three targets (the board, the shark hole, the saw hole) the reference takes the most confident oriented box, drops
it below the `torpedo_threshold` tuner, orders its corners (top-left, top-right, bottom-left, bottom-right), draws the outline,
normalises the corners with the module's `normalize` ((y - h/2)/w, (x - w/2)/w) and writes visibility, confidence, the four corners,
their centre and the quadrilateral's area (shoelace on the normalised corners, scaled by width / height so that the whole image is
1.0) into the `yolo_torpedoes_board` group; the board additionally steers `relay.point_x / point_y`.  The three blocks differ only in
a field prefix, the outline colour and that relay update, so this implementation is one loop over a table instead of three copies.
"""
from typing import List, Sequence

import numpy as np

import shm
from vision.core.handlers import HandlerBase
from vision.utils.color import bgr_to_gray
from vision.utils.draw import Color, draw_polylines
from vision.yolo.data import OBBData
from vision.yolo.utils import order_points

# (field prefix in the shm group, outline colour, corner-name style of the group's fields)
_TARGETS = (("board", Color.LIME, True), ("shark", Color.BLUE, False), ("saw", Color.RED, False))


class TorpedoesOBB(HandlerBase):

    def compute_area_normalized(self, corners: Sequence, img_shape) -> float:
        """Area of the quadrilateral given as (y, x) corners in normalised coordinates, as a fraction of the image
        (handlers/torpedoes.py:23-50): |shoelace| / 2, times width / height because both axes were divided by the width."""
        acc = 0.0
        n = len(corners)
        for i in range(n):
            (y0, x0), (y1, x1) = corners[i], corners[(i + 1) % n]
            acc += x0 * y1 - x1 * y0
        height, width = img_shape[0], img_shape[1]
        return abs(acc) / 2.0 * (width / height)

    def _best(self, results: List[OBBData]):
        if not results:
            return None
        best = max(results, key=lambda d: d.confidence)
        return None if best.confidence < self.tuners["torpedo_threshold"] else best

    def process(self, direction: str, img: np.ndarray, board_results: List[OBBData], shark_hole_results: List[OBBData],
                saw_hole_results: List[OBBData]):
        group = shm.yolo_torpedoes_board
        out = group.get()
        for (prefix, colour, is_board), results in zip(_TARGETS, (board_results, shark_hole_results, saw_hole_results)):
            det = self._best(results)
            if det is None:
                setattr(out, f"{prefix}_visible", 0)
                continue
            tl, tr, bl, br = order_points([(det.x1, det.y1), (det.x2, det.y2), (det.x3, det.y3), (det.x4, det.y4)])
            draw_polylines(img, [br, tr, tl, bl], color=colour(), isClosed=True, thickness=3)
            # (x, y) pixels -> normalised (y, x)
            br, tr, tl, bl = (self.normalize((p[1], p[0])) for p in (br, tr, tl, bl))
            setattr(out, f"{prefix}_visible", 1)
            setattr(out, f"{prefix}_confidence", det.confidence)
            for name, corner in (("bottom_right", br), ("top_right", tr), ("top_left", tl), ("bottom_left", bl)):
                setattr(out, f"{prefix}_{name}_y", corner[0])
                setattr(out, f"{prefix}_{name}_x", corner[1])
            if is_board:
                setattr(out, "board_center_y", (br[0] + bl[0] + tr[0] + tl[0]) / 4)
                setattr(out, "board_center_x", (br[1] + bl[1] + tr[1] + tl[1]) / 4)
                shm.relay.point_x.set(((tl[1] + bl[1]) / 2 + (tr[1] + br[1]) / 2) / 2)
                shm.relay.point_y.set(((tl[0] + tr[0]) / 2 + (bl[0] + br[0]) / 2) / 2)
            else:
                setattr(out, f"{prefix}_center_x", (tl[1] + tr[1] + bl[1] + br[1]) / 4)
                setattr(out, f"{prefix}_center_y", (tl[0] + tr[0] + bl[0] + br[0]) / 4)
            setattr(out, f"{prefix}_area", self.compute_area_normalized([br, tr, tl, bl], img.shape))
        group.set(out)
        self.post("torpedoes handler", img)

    def post_grayscale(self, img: np.ndarray):
        gray_img, _ = bgr_to_gray(img)
        self.post("torpedoes handler", gray_img)
