import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import frames as F
from vision import _vp
W4, H4, B = 3840, 2160, 32
ctx = _vp.Context(0)
morph = [(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)]
desc4 = _vp.make_chain_desc(W4, H4, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), morph, ccl=1, max_labels=256)
base = [F.s1_buoy(i, W4, H4) for i in range(4)]
d = torch.from_numpy(np.stack([base[i % 4] for i in range(B)])).cuda()
t = {"thr": torch.empty((B, H4, W4), dtype=torch.uint8, device="cuda"), "cln": torch.empty((B, H4, W4), dtype=torch.uint8, device="cuda"),
     "lab": torch.empty((B, H4, W4), dtype=torch.int32, device="cuda"), "st": torch.zeros((B, 256, 5), dtype=torch.int32, device="cuda"),
     "ce": torch.zeros((B, 256, 2), dtype=torch.float64, device="cuda"), "nl": torch.zeros((B,), dtype=torch.int32, device="cuda")}
b = _vp.ChainBuffers()
b.bgr = d.data_ptr()
b.threshed, b.cleaned, b.labels, b.stats, b.centroids, b.nlabels = (t[k].data_ptr() for k in ("thr", "cln", "lab", "st", "ce", "nl"))
for _ in range(3): ctx.chain_run(desc4, b, B)
ctx.synchronize()
ctx.profile_begin(200)
for _ in range(10): ctx.chain_run(desc4, b, B)
pr = ctx.profile_end()
print({k: round(1e3 * v[0] / v[1], 1) for k, v in pr.items()}, t["nl"][:4].tolist())
