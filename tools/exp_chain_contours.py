"""The chain + outer contours of every cleaned mask for 128 S1 frames (the red_buoy body for a batch), a few steps: for per-kernel tables
under rocprofv3 (tools/prof_kernels.sh <tag> tools/exp_chain_contours.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import frames as F
from vision import _vp
W, H, B = 1920, 1080, 128
ctx = _vp.Context(0)
d = torch.from_numpy(np.stack([F.s1_buoy(i % 8, W, H) for i in range(B)])).cuda()
t = {"thr": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"), "cln": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"),
     "lab": torch.empty((B, H, W), dtype=torch.int32, device="cuda"), "st": torch.zeros((B, 256, 5), dtype=torch.int32, device="cuda"),
     "ce": torch.zeros((B, 256, 2), dtype=torch.float64, device="cuda"), "nl": torch.zeros((B,), dtype=torch.int32, device="cuda")}
b = _vp.ChainBuffers(); b.bgr = d.data_ptr()
b.threshed, b.cleaned, b.labels, b.stats, b.centroids, b.nlabels = (t[k].data_ptr() for k in ("thr", "cln", "lab", "st", "ce", "nl"))
cdesc = _vp.make_contour_desc("cleaned", _vp.RETR_EXTERNAL, _vp.CHAIN_APPROX_SIMPLE, 64, 8192)
carr = {"info": torch.zeros((B, 2), dtype=torch.int32, device="cuda"), "counts": torch.zeros((B, 64), dtype=torch.int32, device="cuda"),
        "offsets": torch.zeros((B, 64), dtype=torch.int32, device="cuda"), "is_hole": torch.zeros((B, 64), dtype=torch.uint8, device="cuda"),
        "points": torch.zeros((B, 8192, 2), dtype=torch.int32, device="cuda"), "features": torch.zeros((B, 64, 8), dtype=torch.float64, device="cuda")}
cb = _vp.ContourBuffers()
for k, v in carr.items(): setattr(cb, k, v.data_ptr())
desc = _vp.make_chain_desc(W, H, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), [(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)], ccl=1, max_labels=256)
for _ in range(3): ctx.chain_run_contours(desc, b, cdesc, cb, B)
ctx.synchronize(); t0 = time.perf_counter(); K = 10
for _ in range(K): ctx.chain_run_contours(desc, b, cdesc, cb, B)
ctx.synchronize()
print(f"chain + outer contours, {B} frames: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per step")
