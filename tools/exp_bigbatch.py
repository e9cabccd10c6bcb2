"""One very large batch (> 2^32 pixels in a single call): every frame's outputs must equal those of the same frame in a small batch.
Checks the 64-bit indexing of the chain kernels and the workspace carving at sizes only a 288 GB part allows."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import frames as F
from vision import _vp
W, H = 1920, 1080
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2100
D = 8
ctx = _vp.Context(0)
morph = [(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)]
desc = _vp.make_chain_desc(W, H, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), morph, ccl=1, max_labels=64)
small = torch.from_numpy(np.stack([F.s1_buoy(i, W, H) for i in range(D)])).cuda()

def run(d):
    n = d.shape[0]
    t = {"thr": torch.empty((n, H, W), dtype=torch.uint8, device="cuda"), "cln": torch.empty((n, H, W), dtype=torch.uint8, device="cuda"),
         "lab": torch.empty((n, H, W), dtype=torch.int32, device="cuda"), "st": torch.zeros((n, 64, 5), dtype=torch.int32, device="cuda"),
         "ce": torch.zeros((n, 64, 2), dtype=torch.float64, device="cuda"), "nl": torch.zeros((n,), dtype=torch.int32, device="cuda")}
    b = _vp.ChainBuffers(); b.bgr = d.data_ptr()
    b.threshed, b.cleaned, b.labels, b.stats, b.centroids, b.nlabels = (t[k].data_ptr() for k in ("thr", "cln", "lab", "st", "ce", "nl"))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.chain_run(desc, b, n); ctx.synchronize()
    return t, time.perf_counter() - t0

ref, _ = run(small)
big = small.repeat((B + D - 1) // D, 1, 1, 1)[:B].contiguous()
print(f"batch {B}: {B * W * H / 2**32:.2f} x 2^32 pixels, input {big.numel() / 2**30:.1f} GiB", flush=True)
out, dt = run(big)
out, dt = run(big)
bad = 0
for k in ref:
    for i in range(B):
        if not torch.equal(out[k][i], ref[k][i % D]):
            bad += 1
            print("MISMATCH", k, i, flush=True)
            break
print(f"{'OK' if bad == 0 else 'FAILED'}: {B} frames in {dt * 1e3:.1f} ms = {B / dt:,.0f} frames/s; device memory in use {torch.cuda.memory_allocated() / 2**30:.1f} GiB (+ libvp workspace)")
sys.exit(1 if bad else 0)
