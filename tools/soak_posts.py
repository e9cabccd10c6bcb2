"""Soak of the runtime with posts on: a capture process publishes 1080p frames as fast as it can, the red_buoy harness module runs on
the runtime with its three posts going out by DMA, and a GUI-side reader (the reference's read_frame on every post block) checks every
post it accepts against what the same body produces for each of the four source frames (computed once beforehand): every accepted
post must be bit-equal (CRC) to one of the four.  Reports frames, posts accepted, process RSS at start / end (leaks show as growth).
usage: soak_posts.py [seconds=30]"""
import glob
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "shims")):
    sys.path.insert(0, p)
import zlib
import numpy as np
import psutil
import frames as F
import module_harness as MH
from vision.core.bindings.camera_message_framework import BlockAccessor, ReadStatus

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
W, H = 1920, 1080
base = [F.s1_buoy(i, W, H) for i in range(4)]
# what the body posts for each source frame: run it once per frame outside the runtime and keep the CRCs
from vision import _vp
from vision.devmat import DeviceMat
expect = {"threshed": set(), "threshed_cleaned": set(), "contours": set()}
me = MH.PlainSelf((H, W), False)
for b in base:
    img = DeviceMat.from_host(_vp.default_context(), b)
    th, cl, _, _, _ = MH.buoy_body(me, img, np.zeros((8, 8, 3), np.float32))
    expect["threshed"].add(zlib.crc32(np.ascontiguousarray(np.asarray(th))))
    expect["threshed_cleaned"].add(zlib.crc32(np.ascontiguousarray(np.asarray(cl))))
    expect["contours"].add(zlib.crc32(np.ascontiguousarray(np.asarray(img))))
me.close()
MH.module_argv()
d = f"soak{os.getpid()}"
done, bad, accepted = [], [], {"threshed": 0, "threshed_cleaned": 0, "contours": 0}
stop = threading.Event()


def gui(name, kind):
    last = 0
    with BlockAccessor(name) as r:
        while not stop.is_set():
            st, data, t = r.read_frame()
            if st != ReadStatus.SUCCESS:
                time.sleep(0.0005)
                continue
            if t < last:
                bad.append((kind, "time went backwards"))
            last = t
            if data.shape != ((H, W, 3) if kind == "contours" else (H, W, 1)):
                bad.append((kind, "shape"))
            elif zlib.crc32(data) not in expect[kind]:
                bad.append((kind, "matches none of the four expected images"))
            accepted[kind] += 1


with MH.FeederProcess(d, W, H, "s1_buoy", planes=2, period=0.0):
    mod = MH.buoy_module(lambda *a: done.append(1))([d], MH.buoy_tuners())
    mod._fps = 1000000
    runner = threading.Thread(target=mod)
    runner.start()
    t0 = time.time()
    while len(done) < 5 and time.time() - t0 < 60:
        time.sleep(0.05)
    readers = []
    for kind in accepted:
        hits = glob.glob(f"/dev/shm/auv_visiond_module_{mod._name}_post%*%{kind}#*")
        assert hits, kind
        th = threading.Thread(target=gui, args=(hits[0][len("/dev/shm/auv_visiond_"):], kind))
        th.start()
        readers.append(th)
    proc = psutil.Process()
    rss0, n0, t0 = proc.memory_info().rss, len(done), time.time()
    while time.time() - t0 < seconds:
        time.sleep(5)
        print(f"{time.time() - t0:5.0f} s: {len(done) - n0} frames, posts accepted {accepted}, bad {len(bad)}, rss {proc.memory_info().rss >> 20} MiB", flush=True)
    n1, t1, rss1 = len(done), time.time(), proc.memory_info().rss
    stop.set()
    [th.join(5) for th in readers]
    mod.stop()
    runner.join(15)
MH.unlink_block(d)
print(f"soak: {n1 - n0} frames in {t1 - t0:.1f} s = {(n1 - n0) / (t1 - t0):.0f} frames/s with posts on; accepted by the readers {accepted}; "
      f"bad {bad[:3]}; rss {rss0 >> 20} -> {rss1 >> 20} MiB; dma posts {mod._posts.dma_posts}, host posts {mod._posts.host_posts}")
sys.exit(1 if bad or mod._posts.host_posts else 0)
