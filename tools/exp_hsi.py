import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, frames as F
from oracle import oracle as O
from vision.modules.color_balance import balance
f = F.s1_buoy(0)
got = balance(f, hsi_contrast_correct=True, hsv_contrast_correct=False)
exp = O.color_balance(f, hsi_contrast_correct=True, hsv_contrast_correct=False, mean_mode=1)
d = np.abs(got.astype(int) - exp.astype(int))
print("1080p hsi: max diff", d.max(), "differing values", int((d > 0).sum()), "of", d.size)
t0 = time.perf_counter()
for _ in range(10): balance(f, hsi_contrast_correct=True, hsv_contrast_correct=False)
print("host call with hsi:", (time.perf_counter() - t0) / 10 * 1e3, "ms")
