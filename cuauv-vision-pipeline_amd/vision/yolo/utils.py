"""`vision.yolo.utils` (handlers/torpedoes.py:9 imports `order_points` from it; the package is not in the reference tree)."""
from vision.yolo.ops import order_points  # noqa: F401
