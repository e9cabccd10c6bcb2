"""Host-side cost of one operator call of the vision.utils mirror (tiny image: the kernels are negligible), us per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from vision import _vp
from vision.utils.color import bgr_to_lab, range_threshold
from vision.utils.feature import outer_contours
from vision.utils.transform import morph_remove_noise, rect_kernel
img = np.random.default_rng(0).integers(0, 256, (32, 64, 3), dtype=np.uint8)
img[8:20, 10:40] = (40, 45, 210)
ctx = _vp.default_context()
lab, (l, a, b) = bgr_to_lab(img)
th = range_threshold(a, 150, 255)
k = rect_kernel(5)
def t(name, fn, n=2000):
    for _ in range(50): fn()
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    ctx.synchronize(); print(f"{name:40s} {(time.perf_counter() - t0) / n * 1e6:7.1f} us")
t("bgr_to_lab (ndarray in: upload + sync)", lambda: bgr_to_lab(img))
t("range_threshold (device in)", lambda: range_threshold(a, 150, 255))
t("morph_remove_noise (device in)", lambda: morph_remove_noise(th, k))
t("rect_kernel(5)", lambda: rect_kernel(5))
t("outer_contours (device in, sync)", lambda: outer_contours(th), 500)
t("default_context()", lambda: _vp.default_context())
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): morph_remove_noise(th, k)
pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12); print(s.getvalue()[:2600])
