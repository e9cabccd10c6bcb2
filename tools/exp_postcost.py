"""Where the time of a post goes: host time inside post() / flush() per red_buoy body call, and the copy engine's time per image size.
usage: python tools/exp_postcost.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "shims")):
    sys.path.insert(0, p)
import numpy as np
import frames as F
import module_harness as MH
from vision import _vp
from vision.devmat import DeviceMat

ctx = _vp.default_context()
w, h = 1920, 1080
base = [F.s1_buoy(i, w, h) for i in range(4)]
normal = np.zeros((8, 8, 3), np.float32)
me = MH.PlainSelf((h, w), True, tag="Cost")
acc = {"post": 0.0, "flush": 0.0, "n": 0}
real_post, real_flush = me.post, me.flush


def post(*a, **k):
    t = time.perf_counter(); real_post(*a, **k); acc["post"] += time.perf_counter() - t


def flush(*a, **k):
    t = time.perf_counter(); real_flush(*a, **k); acc["flush"] += time.perf_counter() - t


me.post, me.flush = post, flush
# finer: the steps inside PostQueue.post
import vision.core.posts as P
steps = {}


def timed(obj, name, label):
    real = getattr(obj, name)

    def f(*a, **k):
        t = time.perf_counter()
        try:
            return real(*a, **k)
        finally:
            steps[label] = steps.get(label, 0.0) + time.perf_counter() - t
    setattr(obj, name, f)


timed(me.queue, "_settle", "settle (wait for the block's previous copy + commit)")
timed(me.queue, "_open_block", "open_block")
timed(P._DmaPost, "copy_from", "copy_from (forces deferred operators, queues the copy)")
timed(P._DmaPost, "wait", "  of which event waits")
timed(P._DmaPost, "commit", "commit")
timed(P, "as_mat", "as_mat")
real_begin = P.BlockAccessor.begin_device_write if hasattr(P, "BlockAccessor") else None
from vision.core.bindings.camera_message_framework import BlockAccessor
timed(BlockAccessor, "begin_device_write", "begin_device_write")
for i in range(5):
    MH.buoy_body(me, DeviceMat.from_host(ctx, base[i % 4]), normal); me.flush()
acc.update(post=0.0, flush=0.0)
steps.clear()
N = 300
t_all = 0.0
for i in range(N):
    img = DeviceMat.from_host(ctx, base[i % 4])
    t = time.perf_counter()
    MH.buoy_body(me, img, normal); me.flush()
    t_all += time.perf_counter() - t
for k, v in steps.items():
    print(f"   {k}: {1e3 * v / N:.4f} ms per call")
print(f"body+flush {1e3 * t_all / N:.4f} ms per call; inside post() {1e3 * acc['post'] / N:.4f} ms (3 posts), inside flush() {1e3 * acc['flush'] / N:.4f} ms")
real_flush(wait=True)

# the copy engine alone: one image of each size into an open slot, timed from queueing to the event
lib = _vp.lib()
for nbytes, tag in ((h * w, "mask 2.07 MB"), (h * w * 3, "frame 6.22 MB")):
    img = DeviceMat.from_host(ctx, np.zeros(nbytes, np.uint8))
    blk = me._open_block(f"raw{nbytes}#BGR", 9, nbytes)
    ts, tq = [], []
    for _ in range(50):
        slot, ticket = blk.begin_device_write(ctx, nbytes)
        lib.vp_synchronize(ctx.handle)
        done = C.c_void_p()
        t0 = time.perf_counter()
        lib.vp_post_d2h(ctx.handle, 0, slot, img.dev_ptr, nbytes, C.byref(done))
        t1 = time.perf_counter()
        lib.vp_post_wait(ctx.handle, done)
        t2 = time.perf_counter()
        lib.vp_post_free(ctx.handle, done)
        blk.commit_device_write(ticket, 0, (nbytes,))
        tq.append(t1 - t0); ts.append(t2 - t0)
    print(f"{tag}: queueing {1e6 * np.median(tq):.1f} us, queue -> arrived {1e6 * np.median(ts):.1f} us = {nbytes / np.median(ts) / 1e9:.1f} GB/s")
me.close()
