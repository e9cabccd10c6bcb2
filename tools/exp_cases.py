"""Per-kernel cost of the chain for the frame families of SURVEY 8d and for crowded masks (no morphology, raw noise):
the numbers the labelling work of a round is compared against.  Batch B (default 128) of 1080p frames resident in HBM."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import frames as F
from vision import _vp
W, H = 1920, 1080
B = int(os.environ.get("B", "128")); K = int(os.environ.get("K", "10"))
ML = int(os.environ.get("ML", "256"))
ctx = _vp.Context(0)


def bufs_for(d, ml, labels=True):
    t = {"thr": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"), "cln": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"),
         "lab": torch.empty((B, H, W), dtype=torch.int32, device="cuda"), "st": torch.zeros((B, ml, 5), dtype=torch.int32, device="cuda"),
         "ce": torch.zeros((B, ml, 2), dtype=torch.float64, device="cuda"), "nl": torch.zeros((B,), dtype=torch.int32, device="cuda")}
    b = _vp.ChainBuffers(); b.bgr = d.data_ptr()
    b.threshed, b.cleaned, b.stats, b.centroids, b.nlabels = (t[k].data_ptr() for k in ("thr", "cln", "st", "ce", "nl"))
    if labels: b.labels = t["lab"].data_ptr()
    return b, t


def run(name, d, mode, lo, hi, morph, ml=ML):
    b, t = bufs_for(d, ml)
    desc = _vp.make_chain_desc(W, H, mode, lo, hi, morph, ccl=1, max_labels=ml)
    for _ in range(2): ctx.chain_run(desc, b, B)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(K): ctx.chain_run(desc, b, B)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / K
    ctx.profile_begin(K * 24)
    for _ in range(K): ctx.chain_run(desc, b, B)
    pr = ctx.profile_end()
    nl = t["nl"].cpu().numpy()
    row = {"case": name, "ms_per_step": round(1e3 * dt, 4), "fps": round(B / dt, 0), "labels_per_frame": [int(nl.min()), int(nl.max())],
           "kernels_us": {k.replace("k_", ""): round(1e3 * v[0] / v[1], 1) for k, v in pr.items()}}
    print(json.dumps(row), flush=True)
    del t
    return row


OC = [(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)]
rows = []
s1 = torch.from_numpy(np.stack([F.s1_buoy(i % 8, W, H) for i in range(B)])).cuda()
rows.append(run("S1 LAB-a[150,255] OPEN CLOSE", s1, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), OC))
rows.append(run("S1 speckle LAB-a[118,255] OPEN CLOSE", s1, _vp.BGR2LAB, (0, 118, 0), (255, 255, 255), OC, ml=4096))
rows.append(run("S1 everything foreground", s1, _vp.BGR2LAB, (0, 0, 0), (255, 255, 255), OC))
rows.append(run("S1 LAB-a[150,255] no morphology (salt + edges)", s1, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), [], ml=4096))
del s1
s2 = torch.from_numpy(np.stack([F.s2_bins(i % 8, W, H) for i in range(B)])).cuda()
rows.append(run("S2 HSV[10,20,60]-[30,100,255] OPEN", s2, _vp.BGR2HSV, (10, 20, 60), (30, 100, 255), [(_vp.MORPH_OPEN, 5, 5)]))
del s2
s3 = torch.from_numpy(np.stack([F.s3_noise(i % 8, W, H) for i in range(B)])).cuda()
rows.append(run("S3 noise LAB-a[150,255] OPEN CLOSE", s3, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), OC))
for lo_g, tag in ((230, "2 %"), (190, "10 %"), (128, "50 %")):
    rows.append(run("S3 noise grey>=%d (%s density) no morphology" % (lo_g, tag), s3, _vp.BGR2GRAY, (lo_g, 0, 0), (255, 255, 255), [], ml=1 << 20 if B <= 32 else 65536))
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
