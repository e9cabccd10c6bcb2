"""Repeatability of the crowded-frame labelling: raw noise at three densities, a batch of distinct frames, many runs - labels,
statistics, centroids and counts of every run must equal the first run's (the union-finds are racy by design; their RESULT is not).
usage: stress_noise.py [frames=64] [runs=40]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import frames as F
from vision import _vp
W, H = 1920, 1080
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ML = 1 << 18
ctx = _vp.Context(0)
d = torch.from_numpy(np.stack([F.s3_noise(i % 16, W, H) for i in range(B)])).cuda()
t = {"thr": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"), "lab": torch.empty((B, H, W), dtype=torch.int32, device="cuda"),
     "st": torch.zeros((B, ML, 5), dtype=torch.int32, device="cuda"), "ce": torch.zeros((B, ML, 2), dtype=torch.float64, device="cuda"),
     "nl": torch.zeros((B,), dtype=torch.int32, device="cuda")}
b = _vp.ChainBuffers(); b.bgr = d.data_ptr(); b.threshed = t["thr"].data_ptr(); b.labels = t["lab"].data_ptr()
b.stats, b.centroids, b.nlabels = t["st"].data_ptr(), t["ce"].data_ptr(), t["nl"].data_ptr()
bad = 0
for lo in (230, 190, 128):
    for numbering in (_vp.CCL_BLOCK2X2, _vp.CCL_PIXEL):
        desc = _vp.make_chain_desc(W, H, _vp.BGR2GRAY, (lo, 0, 0), (255, 255, 255), [], ccl=1, max_labels=ML, numbering=numbering)
        ctx.chain_run(desc, b, B); ctx.synchronize()
        ref = {k: t[k].clone() for k in ("lab", "st", "nl")}
        refce = t["ce"].clone().view(torch.int64)
        # copies of one frame inside the batch must agree with each other, too
        for f in range(16, B):
            if not torch.equal(ref["lab"][f], ref["lab"][f % 16]) or int(ref["nl"][f]) != int(ref["nl"][f % 16]):
                bad += 1; print("copy differs", lo, numbering, f)
        for r in range(R):
            ctx.chain_run(desc, b, B); ctx.synchronize()
            ok = all(torch.equal(t[k], ref[k]) for k in ("lab", "st", "nl")) and torch.equal(t["ce"].view(torch.int64), refce)
            if not ok:
                bad += 1; print("run differs", lo, numbering, r)
        print(f"lo {lo} numbering {numbering}: {R} runs of {B} frames, labels per frame {int(ref['nl'].min())}..{int(ref['nl'].max())}", flush=True)
print("mismatches:", bad)
