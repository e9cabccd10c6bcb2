"""Harness stand-ins for the four modules BASELINE.json's north_star names (red_buoy, bins, gate, preprocessor), written against the
behaviour SURVEY.md section 3 records - NOT the reference's files: the reference's own module files cannot run even upstream
(`red_buoy.py:40` calls a method that exists nowhere, `gate.py:21` instantiates at import, `bins.py:74` uses `np.int0`,
`preprocessor.py:35,45` expects attributes `ModuleBase` no longer has).  Each class makes the same calls into
`vision.utils` / the cv2 stand-in, in the same order and with the same arguments as the lines it cites, and fills the gaps the way
SURVEY says a harness should (a subclass hook for the missing method, a main guard, `np.intp`).

Used by the GPU tests (tests/test_gpu_harness.py), by bench.py's `extras` (process-body and runtime rates) and by tools/exp_*.py.
Test infrastructure: nothing under cuauv-vision-pipeline_amd/ imports this file.
"""
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "shims")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def module_argv(*flags):
    """ModuleBase parses sys.argv itself (core/base.py:599-635): the harness sets the flags a launch line would carry."""
    sys.argv = [sys.argv[0] if sys.argv else "module"] + list(flags)


# ---- red_buoy (modules/red_buoy.py:11-52) ----------------------------------------------------------------------------------------
def buoy_body(self, image, normal):
    """LAB-a threshold -> OPEN -> CLOSE (posted only) -> outer contours of the *threshold* mask -> overlay -> largest contour ->
    centroid / area -> normalised centre into the shm group.  `self` offers tuners / post / normalize and the contour choice the
    reference leaves out."""
    import shm
    from vision.utils.color import bgr_to_lab, range_threshold
    from vision.utils.draw import draw_contours
    from vision.utils.feature import contour_area, contour_centroid, outer_contours
    from vision.utils.transform import morph_close_holes, morph_remove_noise, rect_kernel
    lab, (lab_l, lab_a, lab_b) = bgr_to_lab(image)
    threshed = range_threshold(lab_a, self.tuners["thresh_min"], self.tuners["thresh_max"])
    self.post("threshed", threshed, "GRAY")
    kernel = rect_kernel(5)
    cleaned = morph_close_holes(morph_remove_noise(threshed, kernel), kernel)
    self.post("threshed_cleaned", cleaned, "GRAY")
    contours = self._contours = outer_contours(threshed)
    draw_contours(image, contours, thickness=10)
    contour = self.extract_most_likely_contour()
    x, y = contour_centroid(contour)
    area = contour_area(contour)
    ny, nx = self.normalize((y, x))
    shm.red_buoy_results.center_x.set(nx)
    shm.red_buoy_results.center_x.set(ny)          # (the reference writes center_x twice and never center_y: red_buoy.py:48-49)
    shm.red_buoy_results.area.set(area)
    self.post("contours", image)
    return threshed, cleaned, contours, (x, y), area


def largest_contour(self):
    from vision.utils.feature import contour_area
    return max(self._contours, key=contour_area)


def buoy_module(on_frame=None):
    """-> a ModuleBase subclass named BuoyLAB listening on zed[forward] + zed[normal] (red_buoy.py:18)."""
    from vision.core.base import ModuleBase, sources

    class BuoyLAB(ModuleBase):
        @sources("zed[forward]", "zed[normal]")
        def process_img(self, image, normal):
            out = buoy_body(self, image, normal)
            if on_frame is not None:
                on_frame(self, image, out)

        extract_most_likely_contour = largest_contour
    return BuoyLAB


def buoy_tuners(lo=150, hi=255):
    from vision.core.tuners import IntTuner
    return [IntTuner("thresh_min", lo, 0, 255), IntTuner("thresh_max", hi, 0, 255)]


class PlainSelf:
    """What a module body touches of ModuleBase when it is called outside the runtime (body rates: bench.py extras).  With posts on,
    post() and flush() are the runtime's own (`vision.core.posts.PostQueue` over real post blocks under /dev/shm, named as
    ModuleManager names them): the body rate with posts includes everything a module pays for them."""

    def __init__(self, shape_hw, posts, tuners=None, tag="Body"):
        import contextlib
        from vision.core.posts import PostQueue
        self.tuners = dict(tuners or {"thresh_min": 150, "thresh_max": 255})
        self.posts, self.shape = posts, shape_hw
        self._stem = f"module_{tag}{os.getpid()}-on-x"
        self._stack, self._blocks = contextlib.ExitStack(), {}
        self.queue = PostQueue(self._open_block, self._write_host, enabled=bool(posts))

    def _open_block(self, key, idx, nbytes):
        from vision.core.bindings.camera_message_framework import BlockAccessor
        block = self._blocks.get(key)
        if block is None:
            block = self._blocks[key] = self._stack.enter_context(BlockAccessor(f"{self._stem}_post%{idx}%{key}", nbytes))
        return block

    def _write_host(self, key, idx, stamp, data):
        self._open_block(key, idx, data.nbytes).write_frame(stamp, data)

    def post(self, name, image, color_space="BGR"):
        self.queue.post(name, image, color_space)           # a no-op with posts off (--enable-performance, core/base.py:857)

    def flush(self, wait=False):
        self.queue.flush(wait)

    def block_names(self):
        return {key: b.direction for key, b in self._blocks.items()}

    def close(self):
        self.queue.drain()
        self._stack.close()
        self._blocks.clear()

    def normalize(self, c):
        return (c[0] - self.shape[0] / 2) / self.shape[1], (c[1] - self.shape[1] / 2) / self.shape[1]

    extract_most_likely_contour = largest_contour


# ---- bins (modules/bins.py:11-81) -------------------------------------------------------------------------------------------------
def bins_body(self, direction, img):
    """HSV beige inRange -> translucent mask overlay -> OPEN 5x5 -> outer contours -> minAreaRect filter (area >= 500, aspect 1..3) ->
    boxes drawn into the overlay -> post."""
    import cv2
    from vision.utils.feature import outer_contours
    from vision.utils.transform import morph_remove_noise, rect_kernel
    hsv = cv2.cvtColor(img, cv2.COLOR_BGR2HSV)
    mask = cv2.inRange(hsv, np.array([10, 20, 60]), np.array([30, 100, 255]))
    overlayed = cv2.addWeighted(img, 0.7, cv2.cvtColor(mask, cv2.COLOR_GRAY2BGR), 0.3, 0)
    cleaned = morph_remove_noise(mask, rect_kernel(5))
    contours = outer_contours(cleaned)
    valid = []
    for contour in contours:
        rect = cv2.minAreaRect(contour)
        (w, h) = rect[1]
        if w * h >= 500 and 1.0 <= max(w, h) / min(w, h) <= 3.0:
            valid.append(rect)
    for rect in valid:
        cv2.drawContours(overlayed, [np.intp(cv2.boxPoints(rect))], 0, (0, 255, 0), 4)      # np.int0 upstream (gone in numpy 2)
    self.post("bins", overlayed)
    return cleaned, contours, valid, overlayed


def bins_module(on_frame=None):
    from vision.core.base import ModuleBase

    class BinDetector(ModuleBase):
        def process(self, direction, img):
            out = bins_body(self, direction, img)
            if on_frame is not None:
                on_frame(self, direction, img, out)
    return BinDetector


# ---- gate (modules/gate.py:8-21) ---------------------------------------------------------------------------------------------------
def gate_tuners():
    from vision.core.tuners import DoubleTuner, IntTuner
    return [IntTuner("rgb_a", 0, 0, 255), DoubleTuner("area_tuner", -23.23, -50, 2130)]


def gate_module(on_frame=None):
    """The echo module: posts every frame as post_<direction>, reports normalize((600, 800)) and the latency of the direction
    (gate.py:14-17; the reference prints them, the harness hands them to `on_frame` as well)."""
    from vision.core.base import ModuleBase

    class GateVision(ModuleBase):
        def process(self, direction, image):
            self.post(f"post_{direction}", image)
            norm = self.normalize((600, 800))
            lat = self.get_latency()
            print(f"normalized (y, x) for {direction}", norm)
            print(f"latency {direction}", lat)
            if on_frame is not None:
                on_frame(self, direction, image, norm, lat)
    return GateVision


# ---- preprocessor (modules/preprocessor.py:7-151) ----------------------------------------------------------------------------------
PPX_DEFAULTS = dict(
    PPX_grayscale=False, PPX_lab=False, PPX_rgb_split=False, PPX_lab_split=False, PPX_hsv_split=False, PPX_hls_split=False,
    PPX_ycrcb_split=False, PPX_luv_split=False, PPX_color_correction=False, PPX_r_bias=0, PPX_g_bias=0, PPX_b_bias=0,
    PPX_contrast=1, PPX_brightness=0, PPX_gaussian_blur=False, PPX_gaussian_blur_kernel=1, PPX_gaussian_noise=0, PPX_erode=False,
    PPX_erode_kernel=1, PPX_dilate=False, PPX_dilate_kernel=1, PPX_rotate=0, PPX_resize=False, PPX_resize_width=512,
    PPX_resize_height=512, PPX_resize_ratio=1, PPX_translate_x=0, PPX_translate_y=0)
_PPX_RANGES = dict(PPX_r_bias=(-255, 255), PPX_g_bias=(-255, 255), PPX_b_bias=(-255, 255), PPX_contrast=(0, 5), PPX_brightness=(-255, 255),
                   PPX_gaussian_blur_kernel=(1, 100), PPX_gaussian_noise=(0, 255), PPX_erode_kernel=(1, 50), PPX_dilate_kernel=(1, 50),
                   PPX_rotate=(0, 359), PPX_resize_width=(1, 2048), PPX_resize_height=(1, 2048), PPX_resize_ratio=(0.01, 1),
                   PPX_translate_x=(-2048, 2048), PPX_translate_y=(-2048, 2048))
_SPLITS = (("PPX_lab_split", "COLOR_BGR2LAB", "lab", "lab"), ("PPX_hsv_split", "COLOR_BGR2HSV", "hsv", "hsv"),
           ("PPX_hls_split", "COLOR_BGR2HLS", "hls", "hls"), ("PPX_ycrcb_split", "COLOR_BGR2YCrCb", "ycrcb", ("y", "cr", "cb")),
           ("PPX_luv_split", "COLOR_BGR2LUV", "luv", "luv"))


class PreprocessorHarness:
    """The 28 PPX_* options of preprocessor.py:10-41 as tuners (same names, kinds, defaults and ranges) registered in the owning
    module's `options_dict`, and `process(*images)` applying the stages in the order of preprocessor.py:50-150: debug posts of channel
    splits, grey / LAB posts, colour balance, per-channel bias, contrast, brightness, blur, noise, ellipse erode / dilate (side
    2k + 1, on all three channels), rotate about the centre (BORDER_REPLICATE), resize, resize by ratio, translate."""

    def __init__(self, module):
        from vision.core import tuners
        self.module = module
        self.options = []
        for name, default in PPX_DEFAULTS.items():
            if isinstance(default, bool):
                self.options.append(tuners.BoolTuner(name, default))
            elif name in ("PPX_contrast", "PPX_resize_ratio"):
                self.options.append(tuners.DoubleTuner(name, default, *_PPX_RANGES[name]))
            else:
                self.options.append(tuners.IntTuner(name, default, *_PPX_RANGES[name]))
        self.options_dict = {o.name: o for o in self.options}
        for o in self.options:
            self.module.options_dict[o.name] = o

    def set(self, **values):
        for k, v in values.items():
            self.options_dict[k]._current_value = v

    def process(self, *images):
        import cv2
        from vision.modules.color_balance import balance       # imported at every call, as preprocessor.py:48 does
        opt = {k: o.value for k, o in self.options_dict.items()}
        post = self.module.post
        done = []
        for mat in images:
            if opt["PPX_rgb_split"]:
                b, g, r = cv2.split(mat)
                post("PPX_rgb_r_channel", r); post("PPX_rgb_g_channel", g); post("PPX_rgb_b_channel", b)
            for flag, code, tag, names in _SPLITS:
                if opt[flag]:
                    for ch, plane in zip(names, cv2.split(cv2.cvtColor(mat, getattr(cv2, code)))):
                        post(f"PPX_{tag}_{ch}_channel", plane)
            if opt["PPX_grayscale"]:
                post("PPX_grayscale", cv2.cvtColor(mat, cv2.COLOR_BGR2GRAY))
            if opt["PPX_lab"]:
                post("PPX_lab", cv2.cvtColor(mat, cv2.COLOR_BGR2LAB))
            if opt["PPX_color_correction"]:
                mat = balance(mat)
            for idx, key in ((2, "PPX_r_bias"), (1, "PPX_g_bias"), (0, "PPX_b_bias")):
                if opt[key] != 0:
                    planes = list(cv2.split(mat))
                    planes[idx] = cv2.add(opt[key], planes[idx])
                    mat = cv2.merge(planes)
            if opt["PPX_contrast"] != 1:
                mat = np.clip(mat * opt["PPX_contrast"], 0., 255.).astype(np.uint8)
            if opt["PPX_brightness"] != 0:
                mat = np.clip(mat + float(opt["PPX_brightness"]), 0., 255.).astype(np.uint8)
            if opt["PPX_gaussian_blur"]:
                side = opt["PPX_gaussian_blur_kernel"] * 2 + 1
                mat = cv2.GaussianBlur(mat, (side, side), 0)
            if opt["PPX_gaussian_noise"] != 0:
                mat = np.clip(mat + np.random.randn(*mat.shape) * opt["PPX_gaussian_noise"], 0., 255.).astype(np.uint8)
            if opt["PPX_erode"]:
                side = opt["PPX_erode_kernel"] * 2 + 1
                mat = cv2.erode(mat, cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (side, side)))
            if opt["PPX_dilate"]:
                side = opt["PPX_dilate_kernel"] * 2 + 1
                mat = cv2.dilate(mat, cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (side, side)))
            if opt["PPX_rotate"] != 0:
                rot = cv2.getRotationMatrix2D((mat.shape[1] / 2, mat.shape[0] / 2), opt["PPX_rotate"], 1)
                mat = cv2.warpAffine(mat, rot, (mat.shape[1], mat.shape[0]), borderMode=cv2.BORDER_REPLICATE)
            if opt["PPX_resize"]:
                mat = cv2.resize(mat, (opt["PPX_resize_width"], opt["PPX_resize_height"]))
            if opt["PPX_resize_ratio"] != 1:
                mat = cv2.resize(mat, (int(mat.shape[1] * opt["PPX_resize_ratio"]), int(mat.shape[0] * opt["PPX_resize_ratio"])))
            if opt["PPX_translate_x"] != 0 or opt["PPX_translate_y"] != 0:
                shift = np.float32([[1, 0, opt["PPX_translate_x"]], [0, 1, opt["PPX_translate_y"]]])
                mat = cv2.warpAffine(mat, shift, (mat.shape[1], mat.shape[0]))
            done.append(mat)
        return done


class LegacyModule:
    """The two attributes the legacy helper expects of its owner (preprocessor.py:35,45) plus a post() that records."""

    def __init__(self, max_buffer_size=1 << 26):
        self.options_dict, self.max_buffer_size, self.posted = {}, max_buffer_size, {}

    def post(self, name, image, color_space="BGR"):
        self.posted[name] = np.array(image, np.uint8, copy=True)


# ---- driving a module on the runtime -------------------------------------------------------------------------------------------------
_FEEDER = r"""
import signal, sys, time
signal.signal(signal.SIGTERM, lambda *a: sys.exit(0))      # leave through the block's __exit__: it marks the block deleted and unlinks it
sys.path[:0] = %(path)r
import numpy as np
import frames as F
from vision.core.bindings.camera_message_framework import BlockAccessor
direction, w, h, gen, planes, period = %(direction)r, %(w)d, %(h)d, %(gen)r, %(planes)d, %(period)f
base = [getattr(F, gen)(i, w, h) for i in range(4)]
normal = np.zeros((8, 8, 3), np.float32)
entry = base[0].nbytes + (normal.nbytes if planes == 2 else 0)
with BlockAccessor(direction, max_entry_size_bytes=entry) as blk:
    print("ready", flush=True)
    i = 0
    while True:
        frame = base[i %% 4]
        blk.write_frame(int(time.monotonic() * 1000), [("forward", frame), ("normal", normal)] if planes == 2 else frame)
        i += 1
        if period > 0:
            time.sleep(period)
"""


class FeederProcess:
    """A capture source in a process of its own (as on the vehicle: capture_sources/*.py are separate programs): creates the block of
    `direction` and publishes four distinct synthetic frames round robin, free-running (period 0: one write after the other) or paced.
    planes = 2 gives the ("forward", "normal") pair red_buoy listens to on `zed`.  It never touches the GPU."""

    def __init__(self, direction, w=1920, h=1080, gen="s1_buoy", planes=1, period=0.0):
        code = _FEEDER % dict(path=[os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests"), ROOT], direction=direction,
                              w=w, h=h, gen=gen, planes=planes, period=period)
        env = dict(os.environ, VP_DEVICE_FRAMES="0", HIP_VISIBLE_DEVICES="")
        self.proc = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True, env=env)
        line = self.proc.stdout.readline()
        if "ready" not in line:
            self.close()
            raise RuntimeError("the feeder process did not come up")

    def close(self):
        if self.proc.poll() is None:
            self.proc.terminate()
            try:
                self.proc.wait(5)
            except subprocess.TimeoutExpired:
                self.proc.kill()
                self.proc.wait(5)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def unlink_block(direction):
    try:
        os.unlink("/dev/shm/auv_visiond_" + direction)
    except OSError:
        pass


def run_module_for(mod, seconds, counter, settle=1.5):
    """Runs `mod` on its loop thread; -> (frames processed in the timed window, its length in s).  `counter()` = frames so far."""
    runner = threading.Thread(target=mod)
    runner.start()
    try:
        t_end = time.perf_counter() + 60
        while counter() < 3 and time.perf_counter() < t_end:      # contexts, first frames
            time.sleep(0.05)
        time.sleep(settle)
        n0, t0 = counter(), time.perf_counter()
        time.sleep(seconds)
        n1, t1 = counter(), time.perf_counter()
    finally:
        mod.stop()
        runner.join(15)
    return n1 - n0, t1 - t0


# ---- rates (bench.py extras, tools/exp_*.py) ------------------------------------------------------------------------------------------
def body_rates(which="buoy", w=1920, h=1080, calls=200):
    """Calls per second of a module body outside the runtime, a fresh 1080p frame per call made before the call and outside its timing
    (the runtime hands every call a frame of its own: core/base.py:765-768).  Two kinds of frame: `host` = a writable numpy array in
    page-locked memory (what round 2 handed over: the first operator uploads it), `device` = a device image (what the runtime hands
    over now: the copy engine has already put the frame into HBM).  Posts off (--enable-performance) and on."""
    import frames as F
    from vision import _vp
    from vision.core.frames import copy_frame
    from vision.devmat import DeviceMat
    if which == "bins":
        from vision import cv2_facade
        cv2_facade.install()
    ctx = _vp.default_context()
    gen = F.s2_bins if which == "bins" else F.s1_buoy
    base = [gen(i, w, h) for i in range(4)]
    normal = np.zeros((8, 8, 3), np.float32)
    out = {}
    for kind in ("host", "device"):
        for posts in (False, True):
            me = PlainSelf((h, w), posts, tag="Rate" + which.capitalize())

            def call(img):
                # the body, then what the loop does after the handlers of an iteration (core/base.py:832-839): publish the posts
                out = buoy_body(me, img, normal) if which == "buoy" else bins_body(me, "forward", img)
                me.flush()
                return out

            def fresh(i):
                return copy_frame(base[i % 4]) if kind == "host" else DeviceMat.from_host(ctx, base[i % 4])
            try:
                for i in range(3):
                    call(fresh(i))
                t_body = 0.0
                for i in range(calls):
                    img = fresh(i)
                    t1 = time.perf_counter()
                    call(img)
                    t_body += time.perf_counter() - t1
                t1 = time.perf_counter()
                me.flush(wait=True)                         # the last call's posts: the tail belongs to the measurement
                t_body += time.perf_counter() - t1
                counts = (me.queue.dma_posts, me.queue.host_posts)
            finally:
                me.close()
            out[f"{kind}_frame_posts_{'on' if posts else 'off'}"] = {"calls_per_s": round(calls / t_body, 1), "ms_per_call": round(1e3 * t_body / calls, 4)}
            if posts:
                out[f"{kind}_frame_posts_on"]["posts_by_dma"], out[f"{kind}_frame_posts_on"]["posts_by_host_copy"] = counts
    return out


def runtime_rate(which="buoy", seconds=3.0, w=1920, h=1080, period=0.0, flags=("--enable-performance",)):
    """Frames per second of a harness module ON THE RUNTIME, end to end: a capture process publishes frames into a shared-memory block
    faster than the module takes them (`period` s between writes; 0 = one write after the other, about 4-5 k frames a second),
    `ModuleBase.__call__` runs the loop thread: read_messages ->
    frame into HBM -> @sources handler / process()."""
    saved = sys.argv[:]
    module_argv(*flags)
    done = []
    d = f"rt{which}{os.getpid()}"
    if which == "bins":
        from vision import cv2_facade
        cv2_facade.install()
    try:
        with FeederProcess(d, w, h, "s2_bins" if which == "bins" else "s1_buoy", planes=2 if which == "buoy" else 1, period=period):
            count = lambda *a: done.append(1)
            if which == "buoy":
                mod = buoy_module(count)([d], buoy_tuners())
            elif which == "bins":
                mod = bins_module(count)([d], [])
            else:
                mod = gate_module(count)([d], gate_tuners())
            mod._fps = 1000000
            n, dt = run_module_for(mod, seconds, lambda: len(done))
            acc = mod._module_manager.video_accessor(d)
            torn = int(getattr(acc, "torn_reads", 0))
            on_device = bool(getattr(acc, "_dev_state", False)) or os.environ.get("VP_DEVICE_FRAMES", "1") != "0"
    finally:
        sys.argv = saved
        unlink_block(d)
    return {"frames_per_s": round(n / dt, 1), "ms_per_frame": round(1e3 * dt / max(n, 1), 4), "frames": n, "seconds": round(dt, 2),
            "posts": "off" if "--enable-performance" in flags else "on", "frames_reach_the_module_as": "device images (one DMA out of the ring slot)"
            if on_device else "page-locked host copies", "copies_dropped_as_lapped": torn, "feeder_period_s": period}


def pcie_upload_rate(ctx, nbytes=256 << 20, reps=4):
    """GB/s of a plain page-locked host -> device copy on this box, measured now: the bound the host-fed rates are quoted against."""
    from vision import _vp
    import ctypes as C
    src = _vp.pinned_empty(ctx, (nbytes,), np.uint8)
    src[::4096] = 1
    p = C.c_void_p()
    _vp.check(_vp.lib().vp_dev_alloc(ctx.handle, nbytes, C.byref(p)), ctx.handle)
    try:
        _vp.check(_vp.lib().vp_memcpy_h2d(ctx.handle, p, src.ctypes.data, nbytes), ctx.handle)
        t0 = time.perf_counter()
        for _ in range(reps):
            _vp.check(_vp.lib().vp_memcpy_h2d(ctx.handle, p, src.ctypes.data, nbytes), ctx.handle)
        dt = time.perf_counter() - t0
    finally:
        _vp.lib().vp_dev_free(ctx.handle, p)
    return reps * nbytes / dt / 1e9


def host_fed_rate(w, h, batch=32, batches=8, ring=4, share=1):
    """Frames per second of vision.dispatch.BatchDispatcher on this box's device: `batch`-deep batches of host frames through pinned
    staging, the red_buoy chain, statistics back (BASELINE config 4 at 4K).  Beside it the PCIe rate it amounts to."""
    import frames as F
    from vision import _vp
    from vision.dispatch import BatchDispatcher
    base = [F.s1_buoy(i, w, h) for i in range(2)]
    frames = np.stack([base[i % 2] for i in range(batch)])
    chain = dict(color_mode=_vp.BGR2LAB, lo=(0, 150, 0), hi=(255, 255, 255), morph=[(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)], ccl=1,
                 max_labels=256, want=("stats",))
    with BatchDispatcher([0], batch, h, w, chain=chain, rank=0, world=share, ring=ring) as d:
        d.submit(frames); d.collect()
        t0 = time.perf_counter()
        inflight = 0
        for _ in range(batches):
            d.submit(frames); inflight += 1
            if inflight > ring:
                d.collect(); inflight -= 1
        while inflight:
            d.collect(); inflight -= 1
        dt = time.perf_counter() - t0
        n = batches * sum(hi - lo for lo, hi in d.slices)
    return {"frames_per_s": round(n / dt, 1), "pcie_GBps": round(n * w * h * 3 / dt / 1e9, 2), "batch": batch, "ring": ring, "batches": batches,
            "frames_per_batch_fed": n // batches}


# ---- detector (modules/yolo.py:37-165; BASELINE config 5) -----------------------------------------------------------------------------
def detector_tuners():
    from vision.core.tuners import DoubleTuner
    return [DoubleTuner(f"{name}_threshold", default, 0, 1) for name, default in
            (("torpedo", 0.1), ("slalom", 0.0), ("gate", 0.1), ("gate_behind", 0.7), ("bins", 0.4), ("manipulator", 0.4))]


# object -> (classes its handler is given, in the handler's argument order, shm switch, shm direction variable): the one object the
# reference wires up (modules/yolo.py:130-151)
DETECTOR_ROUTES = {"torpedoes": (("torpedo_board", "shark_hole", "saw_hole"), "yolo_torpedoes_board", "yolo_torpedoes_board_direction")}


def detector_module(model_factory):
    """Harness stand-in for the detector module, written from SURVEY's row for component 22 / section 8f rank 4: a ModuleBase +
    HandlerMixin listening on zed[forward]; per frame: post the original, `model.track(image)[0].summary()` -> one record per entry
    (MAP_FN[model.task]) -> per active object whose direction matches, the records of its classes go to its handler's process(); an
    object that is switched off gets the handler's grey post only.  `model_factory()` supplies the network: the reference's import
    `from ultralytics import YOLO` becomes `from vision.yolo.engine import YOLO` - nothing else changes."""
    import shm
    from vision.core.base import ModuleBase, sources
    from vision.core.handlers import HandlerMixin
    from vision.yolo.data import MAP_FN

    class Yolo(ModuleBase, HandlerMixin):
        def __init__(self, video_sources, tuners, handlers, **kw):
            ModuleBase.__init__(self, video_sources, tuners, **kw)
            HandlerMixin.__init__(self, handlers)
            self.model = model_factory()
            self.model.to("cpu" if os.environ.get("CUAUV_LOCALE") == "simulator" else "cuda")
            self.to_record = MAP_FN[self.model.task]

        @sources("zed[forward]")
        def fwd_process(self, image):
            direction = "forward"
            self.post("original image", image)
            records = [self.to_record(entry) for entry in self.model.track(image, verbose=False)[0].summary()]
            for obj, (classes, switch, where) in DETECTOR_ROUTES.items():
                if getattr(shm.active_objects, where).get() != direction:
                    continue
                handler = self.handlers[obj]
                if getattr(shm.active_objects, switch).get():
                    handler.process(direction, image.copy(), *[[r for r in records if r.name == c] for c in classes])
                else:
                    handler.post_grayscale(image)
    return Yolo
