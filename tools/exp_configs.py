"""BASELINE.json configs 3 and 4 timed on one GPU (device-resident frames): not bench lines, numbers for DESIGN.md."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import frames as F
from vision import _vp

def bufs_for(B, H, W, ml=256):
    t = {"thr": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"), "cln": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"),
         "lab": torch.empty((B, H, W), dtype=torch.int32, device="cuda"), "st": torch.zeros((B, ml, 5), dtype=torch.int32, device="cuda"),
         "ce": torch.zeros((B, ml, 2), dtype=torch.float64, device="cuda"), "nl": torch.zeros((B,), dtype=torch.int32, device="cuda")}
    b = _vp.ChainBuffers()
    b.threshed, b.cleaned, b.labels, b.stats, b.centroids, b.nlabels = (t[k].data_ptr() for k in ("thr", "cln", "lab", "st", "ce", "nl"))
    return t, b

# config 4: 4K frames, 32-deep batch; one rank's share on 8 GPUs is 4 frames, the whole batch on 1 GPU is 32
W4, H4 = 3840, 2160
ctx = _vp.Context(0)
morph = [(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)]
desc4 = _vp.make_chain_desc(W4, H4, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), morph, ccl=1, max_labels=256)
base = [F.s1_buoy(i, W4, H4) for i in range(4)]
for B in (4, 32):
    d = torch.from_numpy(np.stack([base[i % 4] for i in range(B)])).cuda()
    keep, b = bufs_for(B, H4, W4)
    b.bgr = d.data_ptr()
    for _ in range(3): ctx.chain_run(desc4, b, B)
    ctx.synchronize(); t0 = time.perf_counter(); K = 10
    for _ in range(K): ctx.chain_run(desc4, b, B)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"config 4: 4K x {B:2d} frames per step: {dt * 1e3:7.3f} ms/step = {B / dt:8.0f} fps ({B * W4 * H4 * 9 / dt / 1e12:.2f} TB/s algorithmic)")
    del d, keep

# config 3: two directions concurrently, one context (= one stream) each, one host thread each
W, H, B = 1920, 1080, 16
descA = _vp.make_chain_desc(W, H, _vp.BGR2HSV, (10, 20, 60), (30, 100, 255), [(_vp.MORPH_OPEN, 5, 5)], ccl=1, max_labels=256)   # bins (modules/bins.py:13-27)
dA = torch.from_numpy(np.stack([F.s2_bins(i, W, H) for i in range(B)])).cuda()
dB = torch.from_numpy(np.stack([F.s1_buoy(i, W, H) for i in range(B)])).cuda()
oB = torch.empty_like(dB)
ctxA, ctxB = _vp.Context(0), _vp.Context(0)
keepA, bA = bufs_for(B, H, W); bA.bgr = dA.data_ptr()
L = _vp.lib()
def run_a(K, out):
    for _ in range(K): ctxA.chain_run(descA, bA, B)
    ctxA.synchronize(); out.append(time.perf_counter())
def run_b(K, out):   # gate-style echo: the frame is handed on unchanged (a device copy on its own stream)
    for _ in range(K): torch_copy()
    sB.synchronize(); out.append(time.perf_counter())
sB = torch.cuda.Stream()
def torch_copy():
    with torch.cuda.stream(sB): oB.copy_(dB, non_blocking=True)
for fn, name in ((lambda K, o: run_a(K, o), "bins chain alone"), (lambda K, o: run_b(K, o), "echo alone")):
    o = []; fn(3, o); t0 = time.perf_counter(); o = []; fn(20, o)
    print(f"config 3, {name}: {B * 20 / (o[0] - t0):8.0f} fps")
oa, ob = [], []
run_a(3, []); run_b(3, [])
t0 = time.perf_counter()
ta = threading.Thread(target=run_a, args=(20, oa)); tb = threading.Thread(target=run_b, args=(20, ob))
ta.start(); tb.start(); ta.join(); tb.join()
print(f"config 3, both directions concurrently: bins {B * 20 / (oa[0] - t0):8.0f} fps, echo {B * 20 / (ob[0] - t0):8.0f} fps, "
      f"aggregate {2 * B * 20 / (max(oa[0], ob[0]) - t0):8.0f} fps")
