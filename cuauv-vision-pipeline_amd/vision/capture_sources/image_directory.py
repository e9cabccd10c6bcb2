#!/usr/bin/env python3
"""Replays the images of a directory into a direction, forever, at a fixed rate.

Mirror of the reference capture_sources/image_directory.py:13-54 (every file is decoded once up front, then
cycled).  cv2.imread is not available in this image; PNG/JPEG/BMP files are decoded with Pillow and converted to
BGR uint8, `.npy` files are loaded as they are."""
import argparse
import itertools
import os

import numpy as np

from vision.core.capture_source import CaptureSource, FpsLimiter


def load_image(path: str) -> np.ndarray:
    if path.endswith(".npy"):
        return np.load(path)
    from PIL import Image
    rgb = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(rgb[:, :, ::-1])   # BGR, like cv2.imread


def image_direction_capture(fps_limiter: FpsLimiter, args):
    direction, directory, fps = args
    files = sorted(f for f in os.listdir(directory) if f.lower().endswith((".png", ".jpg", ".jpeg", ".bmp", ".npy")))
    if not files:
        raise RuntimeError(f"no images found in {directory}")
    images = [load_image(os.path.join(directory, f)) for f in files]
    source = itertools.cycle(images)
    for acq_time in fps_limiter.rate(fps):
        yield direction, acq_time, next(source)


class ImageDirectory(CaptureSource):
    def __init__(self, direction: str, directory: str, fps: int = 10):
        super().__init__()
        self.register_capture_udl(direction, image_direction_capture, (direction, directory, fps))


def main():
    ap = argparse.ArgumentParser(description="publish the images of a directory as a camera direction")
    ap.add_argument("direction")
    ap.add_argument("directory")
    ap.add_argument("-f", "--fps", type=int, default=10)
    a = ap.parse_args()
    ImageDirectory(a.direction, a.directory, a.fps).run_event_loop()


if __name__ == "__main__":
    main()
