import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, frames as F
from oracle import oracle as O
from vision.utils import feature
rng = np.random.default_rng(123)
tot = dif = 0
for t in range(400):
    h, w = int(rng.integers(5, 120)), int(rng.integers(5, 160))
    m = F.random_mask(rng, h, w, rng.uniform(0.25, 0.75))
    if t % 4 == 0:   # thin walls: rings of 1-px lines with content inside
        m = np.zeros((h, w), np.uint8)
        for _ in range(6):
            y0, x0 = int(rng.integers(0, h - 3)), int(rng.integers(0, w - 3)); y1, x1 = int(rng.integers(y0 + 2, h)), int(rng.integers(x0 + 2, w))
            m[y0, x0:x1 + 1] = 255; m[y1, x0:x1 + 1] = 255; m[y0:y1 + 1, x0] = 255; m[y0:y1 + 1, x1] = 255
        m |= (rng.random((h, w)) < 0.08).astype(np.uint8) * 255
    got = feature.find_contours(m, 0, 2)
    exp = O.find_contours(m, 0, 2)
    tot += 1
    same = len(got) == len(exp) and all(np.array_equal(a, b) for a, b in zip(got, exp))
    if not same:
        dif += 1
        if dif <= 2:
            np.save(os.path.join(ROOT, "gpurun_out", f"div_{dif}.npy"), m)
print("differ", dif, "of", tot)
