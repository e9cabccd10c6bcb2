"""Development measurement: PCIe-inclusive rates (host numpy buffers) — never the benchmark value."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import frames as F
from vision import _vp
from vision.utils.chain import run_chain
from vision.utils import color, transform as T, feature
W, H = 1920, 1080
frames = np.stack([F.s1_buoy(i, W, H) for i in range(16)])
morph = [(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)]
for want in (("threshed", "cleaned", "labels", "stats"), ("stats",)):
    run_chain(frames, 0, (0, 150, 0), (255, 255, 255), morph, max_labels=256, want=want)
    t0 = time.perf_counter(); K = 10
    for _ in range(K): run_chain(frames, 0, (0, 150, 0), (255, 255, 255), morph, max_labels=256, want=want)
    dt = time.perf_counter() - t0
    print(f"vp_chain_run_host, outputs {want}: {16 * K / dt:8.1f} fps")
f = frames[0]
def per_op():
    lab, (l, a, b) = color.bgr_to_lab(f)
    th = color.range_threshold(a, 150, 255)
    k = T.rect_kernel(5)
    cl = T.morph_close_holes(T.morph_remove_noise(th, k), k)
    return feature.connected_components(cl, max_labels=256, want_labels=True)
per_op(); t0 = time.perf_counter(); K = 30
for _ in range(K): per_op()
print(f"per-operator API (bgr_to_lab, range_threshold, open, close, connected_components): {K / (time.perf_counter() - t0):8.1f} fps")
t0 = time.perf_counter()
for _ in range(K): feature.outer_contours(per_op()[1].astype(np.uint8) * 255 if False else color.range_threshold(color.bgr_to_lab(f)[1][1], 150, 255))
print(f"bgr_to_lab + range_threshold + outer_contours: {K / (time.perf_counter() - t0):8.1f} fps")

from vision.utils.chain import ChainRunner
for want in (("threshed", "cleaned", "labels", "stats"), ("stats",)):
    r = ChainRunner(16, H, W, 0, (0, 150, 0), (255, 255, 255), morph, max_labels=256, want=want)
    r.input[:] = frames
    r.run(); t0 = time.perf_counter(); K = 20
    for _ in range(K): r.run()
    dt = time.perf_counter() - t0
    print(f"ChainRunner (pinned), outputs {want}: {16 * K / dt:8.1f} fps")
