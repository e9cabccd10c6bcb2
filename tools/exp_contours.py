import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import frames as F
from vision.utils import color, transform as T, feature
f = F.s1_buoy(0)
th = color.range_threshold(color.bgr_to_lab(f)[1][1], 150, 255)
k = T.rect_kernel(5)
cl = T.morph_close_holes(T.morph_remove_noise(th, k), k)
for name, m in (("threshed", th), ("cleaned", cl)):
    for mode in (0, 1):
        cs = feature.find_contours(m, mode, 2)
        t0 = time.perf_counter(); K = 200
        for _ in range(K): feature.find_contours(m, mode, 2)
        dt = (time.perf_counter() - t0) / K
        print(f"{name} mode={mode}: {len(cs)} contours, {sum(len(c) for c in cs)} points, {dt*1e3:.3f} ms/call")
