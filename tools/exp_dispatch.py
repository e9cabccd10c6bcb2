"""Host-fed rate of vision.dispatch.BatchDispatcher on this box's device(s): config 4 (4K frames, 32-deep batches, the red_buoy
chain, statistics coming back) and the same at 1080p; one process, `ring` feeder threads per device.  A one-GPU box gives the
per-device figure (one rank's share: --share 8 feeds only frames [0, 4) of every batch, what each of 8 ranks would do)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import frames as F
from vision import _vp
from vision.dispatch import BatchDispatcher

ndev = _vp.lib().vp_device_count()
share = int(sys.argv[sys.argv.index("--share") + 1]) if "--share" in sys.argv else 1
for (W, H, B, nb) in ((3840, 2160, 32, 6), (1920, 1080, 32, 12)):
    base = [F.s1_buoy(i, W, H) for i in range(4)]
    batch = np.stack([base[i % 4] for i in range(B)])
    chain = dict(color_mode=_vp.BGR2LAB, lo=(0, 150, 0), hi=(255, 255, 255), morph=[(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)], ccl=1,
                 max_labels=256, want=("stats",))
    for ring in (1, 2, 3):
        with BatchDispatcher(list(range(ndev)), B, H, W, chain=chain, rank=0, world=share, ring=ring) as d:
            d.submit(batch); d.collect()                       # warm-up: contexts, workspace, pinned buffers
            t0 = time.perf_counter()
            inflight = 0
            for _ in range(nb):
                d.submit(batch); inflight += 1
                if inflight > ring:
                    d.collect(); inflight -= 1
            while inflight:
                d.collect(); inflight -= 1
            dt = time.perf_counter() - t0
            frames = nb * sum(hi - lo for lo, hi in d.slices)
            print(f"{W}x{H} batch {B}, {ndev} device(s), share 1/{share}, ring {ring}: {frames / dt:8.1f} frames/s host-fed "
                  f"({frames * W * H * 3 / dt / 1e9:.1f} GB/s over PCIe), feeders bound to CPUs {sorted(d.bound_cpus.get(0, []))[:8]}...", flush=True)
