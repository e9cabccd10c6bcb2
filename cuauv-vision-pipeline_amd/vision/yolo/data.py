"""Detection records handed to the handlers and the mapping from a result summary to them (modules/yolo.py:15 imports `MAP_FN`,
`OBBData`, `PoseData`, `YOLOData` from `vision.yolo.data`, a package that is not in the reference tree; field names as read by
handlers/torpedoes.py:60-130).  A summary entry is what `Results.summary()` yields per detection: {"name", "class", "confidence",
"box": {...}, optional "track_id"} with box keys x1, y1, x2, y2 (detect) or x1 .. y4 (obb), or "keypoints": {"x", "y", "visible"} (pose)."""
from dataclasses import dataclass, field
from typing import Callable, Dict, List


@dataclass
class YOLOData:
    name: str
    confidence: float
    x1: float
    y1: float
    x2: float
    y2: float
    track_id: int = -1


@dataclass
class OBBData:
    name: str
    confidence: float
    x1: float
    y1: float
    x2: float
    y2: float
    x3: float
    y3: float
    x4: float
    y4: float
    track_id: int = -1


@dataclass
class PoseData:
    name: str
    confidence: float
    x1: float
    y1: float
    x2: float
    y2: float
    keypoints_x: List[float] = field(default_factory=list)
    keypoints_y: List[float] = field(default_factory=list)
    keypoints_visible: List[float] = field(default_factory=list)
    track_id: int = -1


def _detect(entry: dict) -> YOLOData:
    b = entry["box"]
    return YOLOData(entry["name"], float(entry["confidence"]), b["x1"], b["y1"], b["x2"], b["y2"], int(entry.get("track_id", -1)))


def _obb(entry: dict) -> OBBData:
    b = entry["box"]
    return OBBData(entry["name"], float(entry["confidence"]), b["x1"], b["y1"], b["x2"], b["y2"], b["x3"], b["y3"], b["x4"], b["y4"],
                   int(entry.get("track_id", -1)))


def _pose(entry: dict) -> PoseData:
    b, k = entry["box"], entry.get("keypoints", {})
    return PoseData(entry["name"], float(entry["confidence"]), b["x1"], b["y1"], b["x2"], b["y2"], list(k.get("x", [])), list(k.get("y", [])),
                    list(k.get("visible", [])), int(entry.get("track_id", -1)))


MAP_FN: Dict[str, Callable[[dict], object]] = {"detect": _detect, "obb": _obb, "pose": _pose}
