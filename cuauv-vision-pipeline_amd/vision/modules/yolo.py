"""Detector module (BASELINE config 5), same structure as the reference's modules/yolo.py:37-165: a `ModuleBase` + `HandlerMixin`
that runs the network on every `zed[forward]` frame, turns each summary entry into a record (`MAP_FN[task]`), sorts the records by
class name into per-object lists and hands them to the object's handler when the object is active in `shm.active_objects`.
The network is vision.yolo.engine.YOLO (PyTorch-ROCm YOLOv8n-OBB between the HIP letterbox and rotated-NMS kernels) in place of
`ultralytics.YOLO`; the weight path is the reference's, and when the file is absent (it is outside the reference tree) the network
keeps its seeded random initialisation."""
import os
from pathlib import Path
from typing import Callable, Union

import numpy as np

import shm
from vision.core import tuners
from vision.core.base import ModuleBase, sources
from vision.core.handlers import HandlerMixin
from vision.handlers.torpedoes import TorpedoesOBB
from vision.yolo.data import MAP_FN, OBBData, PoseData, YOLOData
from vision.yolo.engine import YOLO

YOLO_WEIGHT = "obb_v14.pt"

HANDLERS = [TorpedoesOBB("torpedoes")]

TUNERS = [
    tuners.DoubleTuner("torpedo_threshold", 0.1, 0, 1),
    tuners.DoubleTuner("slalom_threshold", 0.0, 0, 1),
    tuners.DoubleTuner("gate_threshold", 0.1, 0, 1),
    tuners.DoubleTuner("gate_behind_threshold", 0.7, 0, 1),
    tuners.DoubleTuner("bins_threshold", 0.4, 0, 1),
    tuners.DoubleTuner("manipulator_threshold", 0.4, 0, 1),
]

DetectionData = Union[YOLOData, OBBData, PoseData]

# class name -> the object whose handler receives it (modules/yolo.py:130-151; only the torpedo board is wired up there)
_TORPEDO_CLASSES = ("torpedo_board", "shark_hole", "saw_hole")


class Yolo(ModuleBase, HandlerMixin):

    def __init__(self, video_sources, tuners, handlers, model=None, **kwargs):
        ModuleBase.__init__(self, video_sources, tuners, **kwargs)
        HandlerMixin.__init__(self, handlers)
        self.weight_path = Path("/home/software/cuauv/workspaces/yolo_weights") / YOLO_WEIGHT
        self.model = model if model is not None else YOLO(str(self.weight_path))
        self.device = "cpu" if os.environ.get("CUAUV_LOCALE") == "simulator" else "cuda"
        self.model.to(self.device)
        self.yolo_model_type = self.model.task
        self.map_fn: Callable[[dict], DetectionData] = MAP_FN[self.yolo_model_type]

    def torpedoes_active(self) -> bool:
        return shm.active_objects.yolo_torpedoes_board.get()

    def torpedoes_direction(self, direction: str) -> bool:
        return shm.active_objects.yolo_torpedoes_board_direction.get() == direction

    @sources("zed[forward]")
    def fwd_process(self, image: np.ndarray):
        direction = "forward"
        self.post("original image", image)
        results = self.model.track(image, verbose=False)[0].summary()
        torpedoes_info = {name: [] for name in _TORPEDO_CLASSES}
        for result in results:
            data: DetectionData = self.map_fn(result)
            if data.name in torpedoes_info and self.torpedoes_active() and self.torpedoes_direction(direction):
                torpedoes_info[data.name].append(data)
        if self.torpedoes_direction(direction):
            if self.torpedoes_active():
                self.handlers["torpedoes"].process(direction, image.copy(), torpedoes_info["torpedo_board"], torpedoes_info["shark_hole"],
                                                   torpedoes_info["saw_hole"])
            else:
                self.handlers["torpedoes"].post_grayscale(image)


if __name__ == "__main__":
    Yolo(video_sources=["zed"], tuners=TUNERS, handlers=HANDLERS)()
