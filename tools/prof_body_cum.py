"""cProfile of the red_buoy / bins body with device frames, posts off, sorted by cumulative time (the wrappers around each launch).
usage: prof_body_cum.py [buoy|bins] [calls=600]"""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import frames as F
import module_harness as MH
from vision import _vp
from vision.devmat import DeviceMat
which = sys.argv[1] if len(sys.argv) > 1 else "buoy"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 600
if which == "bins":
    from vision import cv2_facade
    cv2_facade.install()
ctx = _vp.default_context()
gen = F.s2_bins if which == "bins" else F.s1_buoy
base = [gen(i) for i in range(4)]
normal = np.zeros((8, 8, 3), np.float32)
me = MH.PlainSelf((1080, 1920), False, tag="Prof")
imgs = [DeviceMat.from_host(ctx, base[i % 4]) for i in range(calls + 3)]
call = (lambda img: MH.buoy_body(me, img, normal)) if which == "buoy" else (lambda img: MH.bins_body(me, "forward", img))
for i in range(3):
    call(imgs[i])
pr = cProfile.Profile()
pr.enable()
for i in range(calls):
    call(imgs[3 + i])
pr.disable()
me.close()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(32)
print(s.getvalue())
