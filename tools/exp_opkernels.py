"""Kernel time of the per-operator entries on one 1080p device image (HIP events around 200 back-to-back launches), flat and generic forms.
usage: python tools/exp_opkernels.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import frames as F
from vision import _vp

ctx = _vp.default_context()
lib = _vp.lib()
w, h = 1920, 1080
img = F.s1_buoy(0, w, h)


def alloc(n):
    p = C.c_void_p()
    _vp.check(lib.vp_dev_alloc(ctx.handle, n, C.byref(p)), ctx.handle)
    return p.value


src = alloc(img.nbytes)
_vp.check(lib.vp_memcpy_h2d(ctx.handle, src, img.ctypes.data, img.nbytes), ctx.handle)
dst, p0, p1, p2, m = alloc(img.nbytes), alloc(w * h), alloc(w * h), alloc(w * h), alloc(w * h)
stream = torch.cuda.ExternalStream(lib.vp_get_stream(ctx.handle))


def timeit(tag, fn, nbytes, reps=200):
    for _ in range(5):
        fn()
    lib.vp_synchronize(ctx.handle)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    lib.vp_synchronize(ctx.handle)
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{tag:44s} {us:7.2f} us   {nbytes / us / 1e6:6.2f} TB/s")


planes = (C.c_void_p * 3)(p0, p1, p2)
i32 = lambda *v: (C.c_int32 * 3)(*v)   # noqa: E731
for flat in (1, 0):
    lib.vp_set_option(ctx.handle, _vp.OPT_FLAT_OPS, flat)
    tag = "flat" if flat else "generic"
    npx = w * h
    timeit(f"[{tag}] BGR2LAB interleaved + 3 planes", lambda: lib.vp_cvt_color_dev(ctx.handle, _vp.BGR2LAB, src, w * 3, w, h, dst, planes), npx * 9)
    timeit(f"[{tag}] BGR2LAB interleaved only", lambda: lib.vp_cvt_color_dev(ctx.handle, _vp.BGR2LAB, src, w * 3, w, h, dst, None), npx * 6)
    timeit(f"[{tag}] BGR2LAB plane a only", lambda: lib.vp_cvt_color_dev(ctx.handle, _vp.BGR2LAB, src, w * 3, w, h, None, (C.c_void_p * 3)(None, p1, None)), npx * 4)
    timeit(f"[{tag}] BGR2HSV interleaved", lambda: lib.vp_cvt_color_dev(ctx.handle, _vp.BGR2HSV, src, w * 3, w, h, dst, None), npx * 6)
    timeit(f"[{tag}] BGR2GRAY", lambda: lib.vp_cvt_color_dev(ctx.handle, _vp.BGR2GRAY, src, w * 3, w, h, p0, None), npx * 4)
    timeit(f"[{tag}] GRAY2BGR", lambda: lib.vp_cvt_color_dev(ctx.handle, _vp.GRAY2BGR, p0, w, w, h, dst, None), npx * 4)
    timeit(f"[{tag}] inRange C1", lambda: lib.vp_inrange_u8_dev(ctx.handle, p1, w, w, h, 1, i32(150, 0, 0), i32(255, 0, 0), m), npx * 2)
    timeit(f"[{tag}] inRange C3", lambda: lib.vp_inrange_u8_dev(ctx.handle, dst, w * 3, w, h, 3, i32(10, 20, 60), i32(30, 100, 255), m), npx * 4)
timeit("addWeighted (16 B per lane)", lambda: lib.vp_add_weighted_u8_dev(ctx.handle, src, C.c_double(0.7), dst, C.c_double(0.3), C.c_double(0.0), npx * 3, dst), w * h * 9)
