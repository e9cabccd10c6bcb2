"""Host-fed rate of vision.dispatch.BatchDispatcher against the PCIe upload rate measured in the same run, for several numbers of
staging-copy threads per feeder (1 = round 2's single np.copyto).  usage: exp_hostfed.py [1080p|4k]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import module_harness as MH
import numpy as np
import frames as F
from vision import _vp
from vision.dispatch import BatchDispatcher
which = sys.argv[1] if len(sys.argv) > 1 else "1080p"
w, h, nb = (3840, 2160, 4) if which == "4k" else (1920, 1080, 10)
ctx = _vp.default_context()
up = MH.pcie_upload_rate(ctx)
print(f"page-locked upload on this box: {up:.1f} GB/s")
base = [F.s1_buoy(i, w, h) for i in range(2)]
frames = np.stack([base[i % 2] for i in range(32)])
chain = dict(color_mode=_vp.BGR2LAB, lo=(0, 150, 0), hi=(255, 255, 255), morph=[(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)], ccl=1, max_labels=256, want=("stats",))
for ring, ct in ((3, 1), (3, 2), (3, 4), (3, 8), (2, 4), (4, 4), (2, 8)):
    with BatchDispatcher([0], 32, h, w, chain=chain, ring=ring, copy_threads=ct) as d:
        d.submit(frames); d.collect()
        t0 = time.perf_counter(); infl = 0
        for _ in range(nb):
            d.submit(frames); infl += 1
            if infl > ring:
                d.collect(); infl -= 1
        while infl:
            d.collect(); infl -= 1
        dt = time.perf_counter() - t0
    fps = nb * 32 / dt
    print(f"{w}x{h} ring {ring} copy threads {ct}: {fps:8.1f} frames/s = {fps * w * h * 3 / 1e9:5.1f} GB/s = {fps * w * h * 3 / 1e9 / up:.2f} of the link", flush=True)
