"""Detector pre / post-processing on the GPU (BASELINE config 5): `letterbox`, `nms`, `nms_rotated`, and the detection records
the handlers read (`OBBData`: handlers/torpedoes.py:76-82 uses name, confidence, x1..x4, y1..y4).

The reference imports `vision.yolo.data` (modules/yolo.py:15) but that package is not in its tree, and the steps themselves live
inside ultralytics; see include/vp.h for what these functions follow."""
from vision.yolo.ops import letterbox, nms, nms_rotated, order_points, scale_boxes  # noqa: F401
from vision.yolo.data import MAP_FN, OBBData, PoseData, YOLOData  # noqa: F401
