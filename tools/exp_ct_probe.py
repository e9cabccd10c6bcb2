"""Phase times inside the one-block contour bookkeeping (k_ct_jump, LDS form): needs a libvp built with -DVP_CT_PROBE, loaded through
VP_LIB; the kernel prints its stamps (10 ns units) for frame 0.   usage: VP_LIB=.../libvp_ctprobe.so python tools/exp_ct_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import frames as F
from vision import _vp
from vision.devmat import DeviceMat
from vision.utils import color, feature

ctx = _vp.default_context()
th = color.range_threshold(color.bgr_to_lab(F.s1_buoy(0))[1][1], 150, 255)
m = DeviceMat.from_host(ctx, np.ascontiguousarray(np.asarray(th)), binary=True)
for _ in range(6):
    feature.outer_contours(m)
for _ in range(3):
    feature.all_contours(m)
