#!/usr/bin/env python3
"""Plays a frame stack into one or more directions at the stack's frame rate.

Mirror of the reference capture_sources/video.py:9-39 (one decoded frame per tick, fanned out to every listed
direction).  DECODING IS THE CALLER'S: the reference opens the file with cv2.VideoCapture (capture_sources/video.py:14-20, a CPU
decoder outside the measured path); no video decoder exists in this image, so this source plays frames that are already decoded - an
`.npy` array (n, h, w, 3), memory-mapped so that a 4K clip does not have to fit in RAM.  Everything after the decoder - pacing, fan-out
to the listed directions, the block writes - is the reference's behaviour."""
import argparse

import numpy as np

from vision.core.capture_source import CaptureSource, FpsLimiter


def video_to_directions(fps_limiter: FpsLimiter, args):
    path, directions, fps, loop = args
    frames = np.load(path, mmap_mode="r")
    if frames.ndim != 4:
        raise RuntimeError("expected an (n, h, w, c) frame stack")
    idx = 0
    for acq_time in fps_limiter.rate(fps):
        if idx >= len(frames):
            if not loop:
                return
            idx = 0
        frame = np.ascontiguousarray(frames[idx])
        idx += 1
        for direction in directions:
            yield direction, acq_time, frame


class Video(CaptureSource):
    def __init__(self, path: str, directions, fps: int = 30, loop: bool = True):
        super().__init__()
        self.register_capture_udl("video", video_to_directions, (path, tuple(directions), fps, loop))


def main():
    ap = argparse.ArgumentParser(description="publish a frame stack (.npy) as camera directions")
    ap.add_argument("path")
    ap.add_argument("directions", nargs="+")
    ap.add_argument("-f", "--fps", type=int, default=30)
    ap.add_argument("--once", action="store_true")
    a = ap.parse_args()
    Video(a.path, a.directions, a.fps, not a.once).run_event_loop()


if __name__ == "__main__":
    main()
