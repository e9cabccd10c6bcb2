"""Where the per-operator API spends its time (1080p, host numpy arrays)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import frames as F
from vision.utils import color, transform as T, feature
f = F.s1_buoy(0)
k = T.rect_kernel(5)
def t(name, fn, K=30):
    fn(); t0 = time.perf_counter()
    for _ in range(K): r = fn()
    print(f"{name:28s} {(time.perf_counter() - t0) / K * 1e3:7.3f} ms"); return r
lab, (l, a, b) = t("bgr_to_lab", lambda: color.bgr_to_lab(f))
th = t("range_threshold", lambda: color.range_threshold(a, 150, 255))
op = t("morph_remove_noise", lambda: T.morph_remove_noise(th, k))
cl = t("morph_close_holes", lambda: T.morph_close_holes(op, k))
t("canny(50, 150)", lambda: feature.canny(f, 50, 150))
t("adaptive_threshold_mean(15)", lambda: color.adaptive_threshold_mean(a, 15, 2))
t("rotate(12.5 deg)", lambda: T.rotate(f, 12.5))
t("bgr_to_hls", lambda: color.bgr_to_hls(f))
t("connected_components+labels", lambda: feature.connected_components(cl, max_labels=256, want_labels=True))
t("connected_components stats", lambda: feature.connected_components(cl, max_labels=256, want_labels=False))
t("outer_contours", lambda: feature.outer_contours(cl))
t("np.empty 12MB + touch", lambda: np.empty((1080, 1920, 3), np.uint8).fill(0))
