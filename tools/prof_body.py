"""cProfile of the red_buoy body with device frames (what the runtime hands over), posts off: where the host time of a call goes.
usage: prof_body.py [buoy|bins] [calls=400] [posts]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import frames as F
import module_harness as MH
from vision import _vp
from vision.devmat import DeviceMat
which = sys.argv[1] if len(sys.argv) > 1 else "buoy"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 400
posts = len(sys.argv) > 3 and sys.argv[3] == "posts"
if which == "bins":
    from vision import cv2_facade
    cv2_facade.install()
ctx = _vp.default_context()
gen = F.s2_bins if which == "bins" else F.s1_buoy
base = [gen(i) for i in range(4)]
normal = np.zeros((8, 8, 3), np.float32)
me = MH.PlainSelf((1080, 1920), posts, tag="Prof")


def call(img):
    out = MH.buoy_body(me, img, normal) if which == "buoy" else MH.bins_body(me, "forward", img)
    me.flush()
    return out
imgs = [DeviceMat.from_host(ctx, base[i % 4]) for i in range(calls + 3)]
for i in range(3):
    call(imgs[i])
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(calls):
    call(imgs[3 + i])
pr.disable()
dt = time.perf_counter() - t0
me.close()
print(f"{which} body, device frames, posts {'on' if posts else 'off'}, under cProfile: {1e3 * dt / calls:.3f} ms per call")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(24)
print(s.getvalue())
