"""cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) of ONE 1080p device image - the call modules/red_buoy.py:38 makes on the
un-cleaned threshold mask (utils/feature.py:5-21) - on the masks a module can meet: S1 threshold mask (blobs + salt), S1 cleaned, S3 raw
noise at 2 % and 10 % density.  ms per call (device image in, contour list out) and contours per call.
usage: python tools/exp_contours_single.py [calls]        (json on the last line)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import frames as F
from vision import _vp
from vision.devmat import DeviceMat
from vision.utils import color, feature, transform as T


def measure(calls=200):
    ctx = _vp.default_context()
    f = F.s1_buoy(0)
    th = color.range_threshold(color.bgr_to_lab(f)[1][1], 150, 255)
    k = T.rect_kernel(5)
    cl = T.morph_close_holes(T.morph_remove_noise(th, k), k)
    g = color.bgr_to_gray(F.s3_noise(0))[0]
    masks = [("s1_threshold_mask", th), ("s1_cleaned_mask", cl), ("s3_noise_2pct", color.range_threshold(g, 230, 255)),
             ("s3_noise_10pct", color.range_threshold(g, 190, 255))]
    out = {}
    for name, m in masks:
        m = DeviceMat.from_host(ctx, np.ascontiguousarray(np.asarray(m)), binary=True)
        cs = feature.outer_contours(m)
        n = max(10, calls // (1 + len(cs) // 2000))
        t0 = time.perf_counter()
        for _ in range(n):
            feature.outer_contours(m)
        dt = (time.perf_counter() - t0) / n
        # the C-ABI call alone, buffers large enough the first time
        npts = int(sum(len(c) for c in cs))
        max_c, max_p = max(256, 2 * len(cs)), max(1 << 14, 2 * npts)
        pts, counts, holes = np.empty((max_p, 2), np.int32), np.empty(max_c, np.int32), np.empty(max_c, np.uint8)
        nc, np_ = _vp.C.c_int32(0), _vp.C.c_int64(0)
        h, w = m.shape
        call = lambda: _vp.check(_vp.lib().vp_find_contours_dev(ctx.handle, m.dev_ptr, w, w, h, 0, 2, _vp.ptr(pts), max_p, _vp.ptr(counts), _vp.ptr(holes),  # noqa: E731
                                                                max_c, _vp.C.byref(nc), _vp.C.byref(np_)), ctx.handle)
        call()
        t0 = time.perf_counter()
        for _ in range(n):
            call()
        dn = (time.perf_counter() - t0) / n
        out[name] = {"ms_per_call": round(1e3 * dt, 4), "ms_native_call": round(1e3 * dn, 4), "contours": len(cs), "points": npts, "calls": n}
    # the first call on a speckled mask after calls on clean ones (the pass is chosen by the last call's head count: this one is
    # answered "too many heads" by the one block and repeated as launches inside the call)
    clean, speck = DeviceMat.from_host(ctx, np.ascontiguousarray(np.asarray(masks[1][1])), binary=True), DeviceMat.from_host(
        ctx, np.ascontiguousarray(np.asarray(masks[3][1])), binary=True)
    feature.outer_contours(speck)
    first = []
    for _ in range(5):
        for _ in range(3):
            feature.outer_contours(clean)
        t0 = time.perf_counter()
        feature.outer_contours(speck)
        first.append(time.perf_counter() - t0)
    out["s3_noise_10pct_first_after_clean"] = {"ms_per_call": round(1e3 * float(np.median(first)), 4), "calls": len(first)}
    return out


if __name__ == "__main__":
    r = measure(int(sys.argv[1]) if len(sys.argv) > 1 else 200)
    for k, v in r.items():
        print(k, v)
    print(json.dumps(r))
