"""Generic (grey / 3-channel / ellipse) morphology timings at 1080p, host arrays."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import frames as F
from vision.utils import transform as T
f = F.s1_buoy(0)
g = np.ascontiguousarray(f[:, :, 1])
for k in (1, 2, 5, 10, 25, 50):
    ke = T.elliptic_kernel(2 * k + 1)
    for name, img in (("bgr", f), ("grey", g)):
        T.erode(img, ke); t0 = time.perf_counter(); K = 5
        for _ in range(K): T.erode(img, ke)
        print(f"erode {name} ellipse {2*k+1:3d}: {(time.perf_counter() - t0) / K * 1e3:8.2f} ms", flush=True)
kr = T.rect_kernel(31)
T.dilate(f, kr); t0 = time.perf_counter()
for _ in range(5): T.dilate(f, kr)
print(f"dilate bgr rect 31: {(time.perf_counter() - t0) / 5 * 1e3:8.2f} ms")
