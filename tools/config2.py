"""BASELINE config 2 as SURVEY 8d words it: S1 1080p, BGR->LAB(a) -> inRange -> OPEN 5x5 -> CLOSE 5x5 -> CCL (+stats); frames
pre-staged in HBM (kernel-only) and host-fed through pinned staging (end to end); >= 1000 timed frames after 100 warm-up
frames, median of 5 runs."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import frames as F
from vision import _vp
from vision.utils.chain import ChainRunner
W, H, B = 1920, 1080, 128
morph = [(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)]
frames = np.stack([F.s1_buoy(i % 8, W, H) for i in range(B)])
ctx = _vp.Context(0)
d = torch.from_numpy(frames).cuda()
t = {"thr": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"), "cln": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"),
     "lab": torch.empty((B, H, W), dtype=torch.int32, device="cuda"), "st": torch.zeros((B, 256, 5), dtype=torch.int32, device="cuda"),
     "ce": torch.zeros((B, 256, 2), dtype=torch.float64, device="cuda"), "nl": torch.zeros((B,), dtype=torch.int32, device="cuda")}
b = _vp.ChainBuffers(); b.bgr = d.data_ptr()
b.threshed, b.cleaned, b.labels, b.stats, b.centroids, b.nlabels = (t[k].data_ptr() for k in ("thr", "cln", "lab", "st", "ce", "nl"))
desc = _vp.make_chain_desc(W, H, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), morph, ccl=1, max_labels=256)
def resident():
    ctx.chain_run(desc, b, B); ctx.synchronize()                     # 128 warm-up frames
    t0 = time.perf_counter()
    for _ in range(8): ctx.chain_run(desc, b, B)                      # 1024 timed frames
    ctx.synchronize()
    return 8 * B / (time.perf_counter() - t0)
print(f"frames resident in HBM, all outputs: median of 5 runs {statistics.median(resident() for _ in range(5)):10.0f} fps")
for want, name in ((("stats",), "stats + centroids only"), (("threshed", "cleaned", "labels", "stats"), "both masks + labels + stats")):
    r = ChainRunner(16, H, W, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), morph, max_labels=256, want=want)
    def fed():
        for k in range(7):                                            # 112 warm-up frames
            r.input[:] = frames[:16]; r.run()
        t0 = time.perf_counter()
        for k in range(63):                                           # 1008 timed frames, the producer's copy into pinned memory included
            r.input[:] = frames[(k % 8) * 16:(k % 8) * 16 + 16]; r.run()
        return 63 * 16 / (time.perf_counter() - t0)
    print(f"host-fed through pinned staging, {name}: median of 5 runs {statistics.median(fed() for _ in range(5)):10.0f} fps")
