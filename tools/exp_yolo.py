"""Letterbox / NMS timings with device-resident tensors."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import frames as F
from vision.yolo import letterbox, nms, nms_rotated
img = torch.from_numpy(F.s1_buoy(0)).cuda()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    def t(name, fn, K=50):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K): fn()
        torch.cuda.synchronize()
        print(f"{name:40s} {(time.perf_counter() - t0) / K * 1e6:8.1f} us")
    t("letterbox 1080p -> 640x640 (device)", lambda: letterbox(img, (640, 640)))
    rng = np.random.default_rng(0)
    for n in (300, 2000, 8400):
        c = rng.uniform(0, 640, (n, 2)); wh = rng.uniform(4, 200, (n, 2))
        b = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)).cuda()
        sc = torch.from_numpy(rng.random(n).astype(np.float32)).cuda()
        t(f"nms greedy n={n} (device, incl. .item())", lambda: nms(b, sc, 0.45))
        br = torch.from_numpy(np.concatenate([c, wh, rng.uniform(-1.5, 1.5, (n, 1))], 1).astype(np.float32)).cuda()
        t(f"nms rotated n={n}", lambda: nms_rotated(br, sc, 0.45))
