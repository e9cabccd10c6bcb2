"""Crowded-frame labelling alone: noise at one density through grey >= lo -> CCL (+ stats, labels), per-kernel times.
usage: exp_noise.py [lo=190] ; env B (frames, default 128), K (steps), ML (max labels)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import frames as F
from vision import _vp
W, H = 1920, 1080
B = int(os.environ.get("B", "128")); K = int(os.environ.get("K", "5")); ML = int(os.environ.get("ML", "65536"))
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 190
ctx = _vp.Context(0)
d = torch.from_numpy(np.stack([F.s3_noise(i % 8, W, H) for i in range(B)])).cuda()
t = {"thr": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"), "lab": torch.empty((B, H, W), dtype=torch.int32, device="cuda"),
     "st": torch.zeros((B, ML, 5), dtype=torch.int32, device="cuda"), "ce": torch.zeros((B, ML, 2), dtype=torch.float64, device="cuda"),
     "nl": torch.zeros((B,), dtype=torch.int32, device="cuda")}
b = _vp.ChainBuffers(); b.bgr = d.data_ptr(); b.threshed = t["thr"].data_ptr(); b.labels = t["lab"].data_ptr()
b.stats, b.centroids, b.nlabels = t["st"].data_ptr(), t["ce"].data_ptr(), t["nl"].data_ptr()
desc = _vp.make_chain_desc(W, H, _vp.BGR2GRAY, (lo, 0, 0), (255, 255, 255), [], ccl=1, max_labels=ML)
for _ in range(2): ctx.chain_run(desc, b, B)
ctx.synchronize(); t0 = time.perf_counter()
for _ in range(K): ctx.chain_run(desc, b, B)
ctx.synchronize(); dt = (time.perf_counter() - t0) / K
ctx.profile_begin(K * 24)
for _ in range(K): ctx.chain_run(desc, b, B)
pr = ctx.profile_end()
print(json.dumps({"lo": lo, "dbg": os.environ.get("VP_CCL3_DBG", "0"), "ms_per_step": round(1e3 * dt, 3), "labels": int(t["nl"][0]),
                  "kernels_us": {k.replace("k_", ""): round(1e3 * v[0] / v[1], 1) for k, v in pr.items()}}), flush=True)
