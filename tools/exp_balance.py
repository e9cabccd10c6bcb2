"""Colour balance timings: host call (1 frame, pageable numpy) and device-resident batch."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import frames as F
from vision import _vp
from vision.modules.color_balance import balance
f = F.s1_buoy(0)
balance(f)
t0 = time.perf_counter(); K = 20
for _ in range(K): balance(f)
print(f"host call, 1080p: {(time.perf_counter() - t0) / K * 1e3:.2f} ms/frame")
B = int(os.environ.get("B", "64"))
ctx = _vp.default_context(); L = _vp.lib()
d = torch.from_numpy(np.stack([F.s1_buoy(i) for i in range(8)] * (B // 8))).cuda()
o = torch.empty_like(d)
for flags, name in ((_vp.CB_DEFAULT, "default (equalize + hsv stretch + clipping)"), (_vp.CB_EQUALIZE_RGB | _vp.CB_EXTREMA_CLIPPING, "without hsv stretch")):
    for _ in range(3):
        _vp.check(L.vp_color_balance_dev(ctx.handle, d.data_ptr(), o.data_ptr(), 1920, 1080, B, flags, 1, 1), ctx.handle)
    ctx.synchronize()
    t0 = time.perf_counter(); K = 20
    for _ in range(K):
        _vp.check(L.vp_color_balance_dev(ctx.handle, d.data_ptr(), o.data_ptr(), 1920, 1080, B, flags, 1, 1), ctx.handle)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / K
    px = B * 1920 * 1080
    bpp = 12 if flags & _vp.CB_HSV_CONTRAST else 9
    print(f"device batch {B} x 1080p, {name}: {dt * 1e3:.3f} ms/step = {B / dt:.0f} fps, {px * bpp / dt / 1e9:.0f} GB/s algorithmic ({bpp} B/px)")
