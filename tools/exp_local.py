"""k_ccl_local cost vs mask content (empty / S1 / dense) at batch 128."""
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
morph = [(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)]
for name, lo, hi in (("empty mask", (0, 254, 0), (255, 255, 255)), ("S1 a in [150,255]", (0, 150, 0), (255, 255, 255)), ("everything", (0, 0, 0), (255, 255, 255)),
                     ("S1 a in [120,255] (background speckle)", (0, 118, 0), (255, 255, 255))):
    desc = _vp.make_chain_desc(W, H, _vp.BGR2LAB, lo, hi, morph, ccl=1, max_labels=256)
    for _ in range(3): ctx.chain_run(desc, b, B)
    ctx.synchronize(); ctx.profile_begin(400)
    for _ in range(10): ctx.chain_run(desc, b, B)
    pr = ctx.profile_end()
    print(f"{name:40s}", {k.replace("k_ccl_", ""): round(1e3 * v[0] / v[1], 1) for k, v in pr.items() if "ccl" in k}, "labels", t["nl"][:3].tolist())
