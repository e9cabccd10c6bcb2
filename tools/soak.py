"""Determinism soak: the chain (+ contours) repeated on the same batches must return identical results every time."""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import frames as F
from vision import _vp
W, H, B = 1920, 1080, 32
ctx = _vp.Context(0)
gens = [F.s1_buoy, F.s2_bins, F.s3_noise]
d = torch.from_numpy(np.stack([gens[i % 3](i, W, H) for i in range(B)])).cuda()
t = {"thr": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"), "cln": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"),
     "lab": torch.empty((B, H, W), dtype=torch.int32, device="cuda"), "st": torch.zeros((B, 4096, 5), dtype=torch.int32, device="cuda"),
     "ce": torch.zeros((B, 4096, 2), dtype=torch.float64, device="cuda"), "nl": torch.zeros((B,), dtype=torch.int32, device="cuda")}
b = _vp.ChainBuffers(); b.bgr = d.data_ptr()
b.threshed, b.cleaned, b.labels, b.stats, b.centroids, b.nlabels = (t[k].data_ptr() for k in ("thr", "cln", "lab", "st", "ce", "nl"))
cdesc = _vp.make_contour_desc("cleaned", 1, 2, 512, 1 << 16)
carr = {"info": torch.zeros((B, 2), dtype=torch.int32, device="cuda"), "counts": torch.zeros((B, 512), dtype=torch.int32, device="cuda"),
        "offsets": torch.zeros((B, 512), dtype=torch.int32, device="cuda"), "is_hole": torch.zeros((B, 512), dtype=torch.uint8, device="cuda"),
        "points": torch.zeros((B, 1 << 16, 2), dtype=torch.int32, device="cuda"), "features": torch.zeros((B, 512, 8), dtype=torch.float64, device="cuda")}
cb = _vp.ContourBuffers()
for k, v in carr.items(): setattr(cb, k, v.data_ptr())
desc = _vp.make_chain_desc(W, H, _vp.BGR2LAB, (0, 140, 0), (255, 255, 255), [(_vp.MORPH_OPEN, 3, 3), (_vp.MORPH_CLOSE, 5, 5)], ccl=1, max_labels=4096)
def digest():
    ctx.chain_run_contours(desc, b, cdesc, cb, B); ctx.synchronize()
    h = hashlib.sha1()
    for k in ("lab", "st", "nl"): h.update(t[k].cpu().numpy().tobytes())
    for k in ("info", "counts", "offsets", "points"): h.update(carr[k].cpu().numpy().tobytes())
    return h.hexdigest()
ref = digest()
t0 = time.time(); n = 0
while time.time() - t0 < float(os.environ.get("SOAK_SECONDS", "40")):
    for _ in range(20): ctx.chain_run_contours(desc, b, cdesc, cb, B)
    assert digest() == ref, f"result changed after {n} runs"
    n += 21
    if n % 420 == 0: print(n, "runs identical", flush=True)
print("soak ok:", n, "runs, digest", ref[:12], "labels", t["nl"][:6].tolist(), "contours", carr["info"][:3, 0].tolist())
