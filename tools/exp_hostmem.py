"""Which host memory should the runtime's frame copies live in?  For a 1080p frame: time of the copy into it, of drawing a frame's
contours into it (host code, vp_draw_polylines_u8), and of the upload from it, for (a) pageable numpy memory, (b) hipHostMalloc'ed
memory (default flags), (c) hipHostMalloc with hipHostMallocNumaUser, (d) numpy memory page-locked in place with hipHostRegister.

usage: python tools/exp_hostmem.py
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import frames as F
from vision import _vp
from vision.utils.color import range_threshold
from vision.utils.draw import draw_contours
from vision.utils.feature import outer_contours

ctx = _vp.default_context()
hip_path = next(line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line)    # the runtime libvp is linked against
hip = C.CDLL(hip_path)
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]

base = F.s1_buoy(0)
nb = base.nbytes
from vision.utils.color import bgr_to_lab
_, (_, a_, _) = bgr_to_lab(base)
cs = outer_contours(range_threshold(a_, 150, 255))
dev = C.c_void_p()
assert hip.hipMalloc(C.byref(dev), nb) == 0
stream = _vp.lib().vp_get_stream(ctx.handle)


def host_malloc(flags):
    p = C.c_void_p()
    rc = hip.hipHostMalloc(C.byref(p), nb, flags)
    if rc != 0:
        return None
    return np.frombuffer((C.c_ubyte * nb).from_address(p.value), np.uint8).reshape(base.shape)


def registered():
    a = np.empty(base.shape, np.uint8)
    a[:] = 0                                       # touch the pages first
    rc = hip.hipHostRegister(a.ctypes.data, nb, 0)
    return a if rc == 0 else None


kinds = {"pageable numpy": lambda: np.empty(base.shape, np.uint8), "hipHostMalloc default": lambda: host_malloc(0),
         "hipHostMalloc NumaUser (0x20000000)": lambda: host_malloc(0x20000000), "hipHostMalloc NonCoherent (0x80000000)": lambda: host_malloc(0x80000000),
         "numpy + hipHostRegister": registered}
print("cpus this process may run on:", len(os.sched_getaffinity(0)))
for name, make in kinds.items():
    bufs = [make() for _ in range(8)]
    if any(b is None for b in bufs):
        print(f"{name:42s} not available")
        continue
    tc = td = tu = 0.0
    n = 40
    for i in range(n):
        b = bufs[i % 8]
        t0 = time.perf_counter(); np.copyto(b, base); t1 = time.perf_counter()
        hip.hipMemcpyAsync(dev, b.ctypes.data, nb, 1, stream); hip.hipStreamSynchronize(stream); t2 = time.perf_counter()
        draw_contours(b, cs, thickness=10); t3 = time.perf_counter()
        if i >= 8:
            tc += t1 - t0; tu += t2 - t1; td += t3 - t2
    k = n - 8
    print(f"{name:42s} copy {1e3 * tc / k:.3f} ms   upload {1e3 * tu / k:.3f} ms ({nb / (tu / k) / 1e9:.1f} GB/s)   draw {1e3 * td / k:.3f} ms")

# where the time of a draw goes: the same image again (warm), after a copy only, after copy + upload
from vision.utils import draw as D
b = np.empty(base.shape, np.uint8)
np.copyto(b, base)
for label, prep in (("same image again", lambda: None), ("after a copy into it", lambda: np.copyto(b, base)),
                    ("after copy + upload", lambda: (np.copyto(b, base), hip.hipMemcpyAsync(dev, b.ctypes.data, nb, 1, stream), hip.hipStreamSynchronize(stream)))):
    t = 0.0
    for i in range(40):
        prep()
        t0 = time.perf_counter(); draw_contours(b, cs, thickness=10); t += time.perf_counter() - t0
    print(f"draw, {label:24s} {1e3 * t / 40:.3f} ms")
polys = [np.asarray(p).reshape(-1, 2) for p in cs]
counts = np.fromiter((len(p) for p in polys), np.int32, len(polys))
p32 = np.ascontiguousarray(np.concatenate(polys), np.int32)
col = np.array([0, 0, 255, 0], np.uint8)
lib = _vp.lib()
t0 = time.perf_counter()
for i in range(40):
    lib.vp_draw_polylines_u8(b.ctypes.data, b.strides[0], 1920, 1080, 3, p32.ctypes.data, counts.ctypes.data, len(polys), 1, col.ctypes.data, 10)
print(f"the C call alone, same image: {1e3 * (time.perf_counter() - t0) / 40:.3f} ms; {len(polys)} contours, {len(p32)} points, "
      f"{sum(int(max(abs(p[i] - p[(i + 1) % len(p)]).max(), 0)) for p in polys for i in range(len(p)))} steps")
