"""Frames per second of a module on the runtime, end to end: a capture process publishes 1080p frames into a shared-memory block as
fast as it can (a free-running camera: the module always finds a new frame), the red_buoy harness module (tests/module_harness.py,
modules/red_buoy.py:19-52) runs on its loop thread in performance mode (posts off).  VP_DEVICE_FRAMES=0: frames are copied out of
the block into page-locked host memory and uploaded by the first operator (round 2); default: one DMA from the ring slot to HBM.

usage: python tools/exp_runtime.py [seconds] [module: buoy | bins | gate] [seconds between the capture process's writes, default 0]
(with VP_DEVICE_FRAMES=0 give the writer a pause, e.g. 0.0002: a seqlock reader that copies 6 MB per frame never finishes a copy beside a
writer that never pauses - the reference's read loop, lib/camera_message_framework.cpp:421-452, has the same property)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import module_harness as MH

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
WHICH = sys.argv[2] if len(sys.argv) > 2 else "buoy"
PERIOD = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
MH.module_argv("--enable-performance")
done = []
d = f"exprt{os.getpid()}"
if WHICH == "bins":
    from vision import cv2_facade
    cv2_facade.install()
with MH.FeederProcess(d, 1920, 1080, "s2_bins" if WHICH == "bins" else "s1_buoy", planes=2 if WHICH == "buoy" else 1, period=PERIOD) as feeder:
    if WHICH == "buoy":
        mod = MH.buoy_module(lambda *a: done.append(1))([d], MH.buoy_tuners())
    elif WHICH == "bins":
        mod = MH.bins_module(lambda *a: done.append(1))([d], [])
    else:
        mod = MH.gate_module(lambda *a: done.append(1))([d], MH.gate_tuners())
    mod._fps = 100000
    n, dt = MH.run_module_for(mod, SECONDS, lambda: len(done))
    acc = mod._module_manager.video_accessor(d)
    torn = getattr(acc, "torn_reads", 0)
MH.unlink_block(d)
print(f"{WHICH} module on the runtime, 1080p, posts off: {n / dt:.1f} frames/s ({1e3 * dt / max(n, 1):.3f} ms per frame, {n} frames; "
      f"device frames {'off' if os.environ.get('VP_DEVICE_FRAMES') == '0' else 'on'}, copies dropped as lapped: {torn}, writer pause {PERIOD} s)")
