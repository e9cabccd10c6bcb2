"""Detection records handed to the handlers (field names as read by handlers/torpedoes.py:60-130)."""
from dataclasses import dataclass


@dataclass
class YOLOData:
    name: str
    confidence: float
    x1: float
    y1: float
    x2: float
    y2: float


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
