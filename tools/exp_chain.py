"""Development experiment: marginal cost of each stage/output of the chain (wall time over K steps)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import frames as F
from vision import _vp
W, H = 1920, 1080
B = int(os.environ.get("B", "64")); K = int(os.environ.get("K", "30"))
ctx = _vp.Context(0)
if os.environ.get("S"): ctx.set_option(_vp.OPT_CHAIN_STREAMS, int(os.environ["S"]))
distinct = [F.s1_buoy(i, W, H) for i in range(8)]
host = np.stack([distinct[i % 8] for i in range(B)])
d_bgr = torch.from_numpy(host).cuda()
d_thr = torch.empty((B, H, W), dtype=torch.uint8, device="cuda"); d_cln = torch.empty_like(d_thr)
d_lab = torch.empty((B, H, W), dtype=torch.int32, device="cuda")
d_stats = torch.zeros((B, 256, 5), dtype=torch.int32, device="cuda"); d_cent = torch.zeros((B, 256, 2), dtype=torch.float64, device="cuda")
d_nl = torch.zeros((B,), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
morph = [(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)]

def run(name, thr=True, cln=True, lab=True, stats=True, ccl=1, morph_ops=morph, prof=False):
    desc = _vp.make_chain_desc(W, H, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), morph_ops, ccl=ccl, max_labels=256)
    b = _vp.ChainBuffers(); b.bgr = d_bgr.data_ptr()
    if thr: b.threshed = d_thr.data_ptr()
    if cln: b.cleaned = d_cln.data_ptr()
    if lab: b.labels = d_lab.data_ptr()
    if stats: b.stats, b.centroids = d_stats.data_ptr(), d_cent.data_ptr()
    b.nlabels = d_nl.data_ptr()
    for _ in range(3): ctx.chain_run(desc, b, B)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(K): ctx.chain_run(desc, b, B)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / K * 1e6
    print(f"{name:34s} {dt:8.1f} us/step  {B / dt * 1e6:10.0f} fps", flush=True)
    if prof:
        ctx.profile_begin(K * 16)
        for _ in range(K): ctx.chain_run(desc, b, B)
        pr = ctx.profile_end()
        print("   ", {k: round(1e3 * v[0] / v[1], 1) for k, v in pr.items()})

run("full", prof=True)
run("no labels", lab=False)
run("no labels, no stats", lab=False, stats=False)
run("no ccl", ccl=0, lab=False, stats=False)
run("no ccl, no cleaned mask", ccl=0, lab=False, stats=False, cln=False)
run("colour only (no morph, no ccl)", ccl=0, lab=False, stats=False, cln=False, morph_ops=[])
run("colour only, no thr mask", ccl=0, lab=False, stats=False, cln=False, thr=False, morph_ops=[])
run("full again", prof=False)
