"""Frames per second of a module on the runtime, end to end: a capture thread publishes 1080p frames into a shared-memory block as
fast as it can (a free-running camera: the module always finds a new frame), a `ModuleBase` subclass with the red_buoy body (modules/red_buoy.py:19-52) runs on its loop thread in
performance mode (posts off).  What a frame costs here beyond the body (tools/exp_process.py): the library's seqlock copy out of the
block and - unless it went straight into page-locked memory that becomes the module's frame (VP_PRIVATE_READS=0 to switch that off) -
the runtime's own copy of it.

usage: python tools/exp_runtime.py [seconds]
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "shims")):
    sys.path.insert(0, p)
sys.argv = [sys.argv[0]] + ["--enable-performance"] + sys.argv[1:]
import numpy as np
import frames as F
import shm
from vision.core.base import ModuleBase, sources
from vision.core.bindings.camera_message_framework import BlockAccessor
from vision.core.tuners import IntTuner
from vision.utils.color import bgr_to_lab, range_threshold
from vision.utils.draw import draw_contours
from vision.utils.feature import contour_area, contour_centroid, outer_contours
from vision.utils.transform import morph_close_holes, morph_remove_noise, rect_kernel

SECONDS = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
done = []


class BuoyLAB(ModuleBase):
    @sources("zed[forward]", "zed[normal]")
    def process_img(self, image, normal):
        lab, (lab_l, lab_a, lab_b) = bgr_to_lab(image)
        threshed = range_threshold(lab_a, self.tuners["thresh_min"], self.tuners["thresh_max"])
        self.post("threshed", threshed, "GRAY")
        kernel = rect_kernel(5)
        cleaned = morph_remove_noise(threshed, kernel)
        cleaned = morph_close_holes(cleaned, kernel)
        self.post("threshed_cleaned", cleaned, "GRAY")
        contours = outer_contours(threshed)
        draw_contours(image, contours, thickness=10)
        contour = max(contours, key=contour_area)
        x, y = contour_centroid(contour)
        area = contour_area(contour)
        ny, nx = self.normalize((y, x))
        shm.red_buoy_results.center_x.set(nx)
        shm.red_buoy_results.center_x.set(ny)
        shm.red_buoy_results.area.set(area)
        self.post("contours", image)
        done.append(time.perf_counter())


d = f"expzed{os.getpid()}"
base = [F.s1_buoy(i) for i in range(4)]
normal = np.zeros((8, 8, 3), np.float32)                  # the second plane red_buoy's signature names (unused by the body)
with BlockAccessor(d, max_entry_size_bytes=base[0].nbytes + normal.nbytes) as w:
    mod = BuoyLAB([d], [IntTuner("thresh_min", 150, 0, 255), IntTuner("thresh_max", 255, 0, 255)])
    mod._fps = 100000
    runner = threading.Thread(target=mod)
    runner.start()
    stop = False

    def capture():                                        # free-running camera, faster than the module: it always finds a new frame
        i = 0
        while not stop:
            w.write_frame(int(time.monotonic() * 1000), [("forward", base[i % 4]), ("normal", normal)])
            i += 1
            time.sleep(0.0002)
    cap = threading.Thread(target=capture)
    cap.start()
    time.sleep(1.5)                                        # contexts, first frames
    n0, t0 = len(done), time.perf_counter()
    time.sleep(SECONDS)
    n1, t1 = len(done), time.perf_counter()
    stop = True
    cap.join()
    mod.stop()
    runner.join(10)
print(f"red_buoy module on the runtime, 1080p, posts off: {(n1 - n0) / (t1 - t0):.1f} frames/s ({1e3 * (t1 - t0) / max(n1 - n0, 1):.3f} ms per frame, "
      f"{n1 - n0} frames; private reads {'off' if os.environ.get('VP_PRIVATE_READS') == '0' else 'on'})")
