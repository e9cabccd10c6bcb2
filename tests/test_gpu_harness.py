"""GPU suite: the four modules north_star names, as harness classes (tests/module_harness.py), on the runtime and through the cv2
stand-in; frames that reach a module as device images (one DMA out of the ring slot, sequence check after the copy)."""
import os
import sys
import threading
import time

import numpy as np
import pytest

import frames as F
import module_harness as MH
from vision.core.bindings.camera_message_framework import BlockAccessor, ReadStatus

pytestmark = pytest.mark.gpu
PID = os.getpid()


@pytest.fixture()
def cv2mod():
    from vision import cv2_facade
    had = sys.modules.get("cv2")
    mod = cv2_facade.install()
    yield mod
    if had is None:
        sys.modules.pop("cv2", None)


def _next_frame(r, timeout=2.0):
    """read_frame_device until it reports something other than NO_NEW_FRAME: with the feeder thread a frame becomes visible a
    fraction of a millisecond after it was written (the runtime's loop polls anyway)."""
    t0 = time.time()
    while True:
        out = r.read_frame_device()
        if out[0] != ReadStatus.NO_NEW_FRAME or time.time() - t0 > timeout:
            return out
        time.sleep(0.0005)


def test_device_reads_deliver_the_frames_that_were_written(vp):
    """read_frame_device: every plane arrives as a device image that equals what the writer published (types, shapes, names, time),
    is private to the reader (later frames do not touch it) and is writable like the runtime's copies."""
    from vision.devmat import DeviceMat
    vp.default_context()
    d = f"pytdev{PID}"
    a, b = F.s1_buoy(0, 320, 200), F.s1_buoy(1, 320, 200)
    depth = np.arange(200 * 320, dtype=np.float32).reshape(200, 320)
    with BlockAccessor(d, max_entry_size_bytes=a.nbytes + depth.nbytes) as w, BlockAccessor(d) as r:
        st, data, t, private = r.read_frame_device()
        assert st == ReadStatus.NO_NEW_FRAME and data is None
        w.write_frame(11, [("forward", a), ("depth", depth)])
        st, data, t, private = _next_frame(r)
        assert st == ReadStatus.SUCCESS and t == 11 and private and r.last_plane_names() == ("forward", "depth")
        fwd, dep = data
        assert isinstance(fwd, DeviceMat) and isinstance(dep, DeviceMat) and fwd._host is None and dep._host is None
        assert fwd.shape == (200, 320, 3) and fwd.dtype == np.uint8 and dep.shape == (200, 320, 1) and dep.dtype == np.float32
        w.write_frame(12, [("forward", b), ("depth", depth * 2)])
        st2, data2, t2, _ = _next_frame(r)
        assert st2 == ReadStatus.SUCCESS and t2 == 12
        assert np.array_equal(fwd, a) and np.array_equal(dep[:, :, 0], depth)            # the first frame is untouched by the second
        assert np.array_equal(data2[0], b) and np.array_equal(data2[1][:, :, 0], depth * 2)
        st3, data3, _, _ = r.read_frame_device()
        assert st3 == ReadStatus.NO_NEW_FRAME and data3 is data2
        fwd[0, 0, 0] = 7                                                                  # writable, and the writes stay the reader's
        assert fwd[0, 0, 0] == 7 and np.asarray(data2[0])[0, 0, 0] == b[0, 0, 0]
        # a single-plane frame comes back as one image, and operators take it without an upload
        w.write_frame(13, a)
        st4, img, t4, _ = _next_frame(r)
        assert st4 == ReadStatus.SUCCESS and isinstance(img, DeviceMat) and img.shape == a.shape
        from vision.utils.color import bgr_to_gray
        assert np.array_equal(bgr_to_gray(img)[0], bgr_to_gray(a)[0]) and img._host is None


def test_device_reads_under_a_lapping_writer(vp):
    """A writer that never pauses laps the three-slot ring while the copy engine is at work: copies the writer overtook are dropped
    (torn_reads counts them), and every frame that is handed out is bit-equal to one frame that was written, with its own time."""
    import zlib
    vp.default_context()
    d = f"pytlap{PID}"
    w_, h_ = 1920, 1080
    rng = np.random.default_rng(5)
    pool = [rng.integers(0, 256, (h_, w_, 3), dtype=np.uint8) for _ in range(4)]
    for k, p in enumerate(pool):
        p[0, 0, 0] = k                                   # which of the four it is
    sums = [zlib.crc32(p) for p in pool]
    stop = threading.Event()
    with BlockAccessor(d, max_entry_size_bytes=pool[0].nbytes) as w, BlockAccessor(d) as r:
        def writer():
            t = 1
            while not stop.is_set():
                w.write_frame(t * 4 + (t % 4), pool[t % 4])     # the time names the frame
                t += 1
        th = threading.Thread(target=writer)
        th.start()
        try:
            got, last_t, t_end = 0, 0, time.time() + 4.0
            while time.time() < t_end and got < 300:
                st, img, t, _ = r.read_frame_device()
                if st != ReadStatus.SUCCESS:
                    continue
                host = np.asarray(img)
                k = int(host[0, 0, 0])
                assert k == t % 4 and zlib.crc32(host) == sums[k], f"frame at t={t} is not the frame that was written"
                assert t > last_t
                last_t = t
                got += 1
        finally:
            stop.set()
            th.join()
        assert got >= 20
        print(f"{got} frames accepted, {r.torn_reads} copies dropped as lapped")


def test_gate_module_echoes_two_directions(vp, oracle, capsys):
    """modules/gate.py:13-21 on two camera directions: every frame comes back as post_<direction> (bit-equal), normalize((600, 800))
    uses the direction's own frame shape ((y - h/2)/w, (x - w/2)/w), get_latency() is the mean age of the direction's last frames."""
    MH.module_argv()
    seen = {}

    def on_frame(mod, direction, image, norm, lat):
        seen.setdefault(direction, []).append((np.array(image, copy=True), norm, lat))
    fwd, dwn = f"pytgfwd{PID}", f"pytgdwn{PID}"
    a, b = F.s1_buoy(3, 640, 360), F.s2_bins(3, 320, 240)
    posts = {}
    with BlockAccessor(fwd, max_entry_size_bytes=a.nbytes) as wf, BlockAccessor(dwn, max_entry_size_bytes=b.nbytes) as wd:
        mod = MH.gate_module(on_frame)(video_sources=[fwd, dwn], tuners=MH.gate_tuners())
        mod._fps = 200
        runner = threading.Thread(target=mod)
        runner.start()
        try:
            t0 = time.time()
            while (len(seen.get(fwd, [])) < 2 or len(seen.get(dwn, [])) < 2) and time.time() - t0 < 30:
                now = int(time.monotonic() * 1000)
                wf.write_frame(now - 40, a)              # 40 ms old when written
                wd.write_frame(now - 15, b)
                time.sleep(0.03)
            # the GUI's end: read the two post blocks back
            name = mod._name
            import glob
            for direction in (fwd, dwn):
                hits = glob.glob(f"/dev/shm/auv_visiond_module_{name}_post%*%post_{direction}#BGR")     # (% idx % orders the GUI's panes)
                if hits:
                    with BlockAccessor(hits[0][len("/dev/shm/auv_visiond_"):]) as rd:
                        st, data, _ = rd.read_frame()
                        if data is not None:
                            posts[direction] = np.array(data, copy=True)
        finally:
            mod.stop()
            runner.join(10)
    assert len(seen[fwd]) >= 2 and len(seen[dwn]) >= 2
    img_f, norm_f, lat_f = seen[fwd][-1]
    img_d, norm_d, lat_d = seen[dwn][-1]
    assert np.array_equal(img_f, a) and np.array_equal(img_d, b)
    assert norm_f == ((600 - 360 / 2) / 640, (800 - 640 / 2) / 640) and norm_d == ((600 - 240 / 2) / 320, (800 - 320 / 2) / 320)
    assert isinstance(lat_f, int) and 35 <= lat_f <= 400 and isinstance(lat_d, int) and 10 <= lat_d <= 400 and lat_f > lat_d
    assert set(posts) == {fwd, dwn}, "the post blocks were not published under module_<name>_post%<idx>%post_<direction>#BGR"
    assert np.array_equal(posts[fwd], a) and np.array_equal(posts[dwn], b)
    out = capsys.readouterr().out
    assert f"normalized (y, x) for {fwd}" in out and f"latency {dwn}" in out


def _resize_ref(img, nw, nh):
    from test_gpu_yolo import _resize_linear_u8
    return _resize_linear_u8(img, nw, nh)


@pytest.mark.parametrize("setting", ["defaults", "morph", "geometry", "balance_and_posts"])
def test_preprocessor_harness(vp, oracle, cv2mod, setting):
    """modules/preprocessor.py:47-151 through the cv2 stand-in at 720p against the oracle: the defaults (identity), ellipse erode +
    dilate on the three channels (:120-129), rotate + resize + translate (:130-149), colour balance + biases + channel-split posts."""
    owner = MH.LegacyModule()
    ppx = MH.PreprocessorHarness(owner)
    assert len(ppx.options) == 28 and set(owner.options_dict) == set(MH.PPX_DEFAULTS)
    img = F.s1_buoy(7, 1280, 720)
    if setting == "defaults":
        (out,) = ppx.process(img)
        assert np.array_equal(out, img) and not owner.posted
        a, b = ppx.process(img, img[::-1].copy())
        assert np.array_equal(b, img[::-1])
        return
    if setting == "morph":
        ppx.set(PPX_erode=True, PPX_erode_kernel=3, PPX_dilate=True, PPX_dilate_kernel=6)
        (out,) = ppx.process(img)
        k7, k13 = oracle.structuring_element(2, 7, 7), oracle.structuring_element(2, 13, 13)
        exp = oracle.morph(oracle.DILATE, oracle.morph(oracle.ERODE, img, k7), k13)
        assert np.array_equal(out, exp)
        return
    if setting == "geometry":
        ppx.set(PPX_rotate=30, PPX_resize=True, PPX_resize_width=640, PPX_resize_height=400, PPX_resize_ratio=0.5, PPX_translate_x=17, PPX_translate_y=-9)
        (out,) = ppx.process(img)
        rot = oracle.rotation_matrix_2d((1280 / 2, 720 / 2), 30, 1)
        e = oracle.warp_affine(img, rot, (1280, 720), border="replicate")
        e = _resize_ref(e, 640, 400)
        e = _resize_ref(e, 320, 200)
        e = oracle.warp_affine(e, np.float32([[1, 0, 17], [0, 1, -9]]), (320, 200))
        assert out.shape == (200, 320, 3) and np.array_equal(out, e)
        return
    ppx.set(PPX_color_correction=True, PPX_r_bias=12, PPX_b_bias=-20, PPX_lab_split=True, PPX_hsv_split=True, PPX_ycrcb_split=True,
            PPX_hls_split=True, PPX_rgb_split=True, PPX_grayscale=True, PPX_lab=True, PPX_gaussian_blur=True, PPX_gaussian_blur_kernel=2,
            PPX_brightness=5)
    (out,) = ppx.process(img)
    lab, hsv, ycc, hls = oracle.bgr2lab(img), oracle.bgr2hsv(img), oracle.bgr2ycrcb(img), oracle.bgr2hls(img)
    for tag, ref, names in (("lab", lab, "lab"), ("hsv", hsv, "hsv"), ("hls", hls, "hls"), ("ycrcb", ycc, ("y", "cr", "cb"))):
        for c, ch in enumerate(names):
            assert np.array_equal(owner.posted[f"PPX_{tag}_{ch}_channel"], ref[:, :, c]), (tag, ch)
    assert np.array_equal(owner.posted["PPX_rgb_r_channel"], img[:, :, 2]) and np.array_equal(owner.posted["PPX_grayscale"], oracle.bgr2gray(img))
    assert np.array_equal(owner.posted["PPX_lab"], lab)
    e = oracle.color_balance(img, mean_mode=0).astype(np.int32)
    e[:, :, 2] = np.clip(e[:, :, 2] + 12, 0, 255)
    e[:, :, 0] = np.clip(e[:, :, 0] - 20, 0, 255)
    e = np.clip(e.astype(np.uint8) + 5.0, 0., 255.).astype(np.uint8)
    e = oracle.gaussian_blur(e, (5, 5))
    assert np.array_equal(out, e)
    # the one conversion of the debug posts that is not restated: it must fail loudly, not fall back
    ppx2 = MH.PreprocessorHarness(MH.LegacyModule())
    ppx2.set(PPX_luv_split=True)
    if not hasattr(oracle, "bgr2luv"):
        with pytest.raises(cv2mod.error):
            ppx2.process(img)


def test_buoy_module_gets_device_frames(vp, oracle):
    """The red_buoy harness on the runtime: its `image` arrives as a device image (no host copy was made for it), the overlay is drawn
    into it on the device, and what it posts equals the oracle's chain + the host rasteriser's overlay."""
    from vision.devmat import DeviceMat
    from vision.utils.draw import draw_contours
    MH.module_argv()
    log = []

    def on_frame(mod, image, out):
        log.append((isinstance(image, DeviceMat), image._host is None if isinstance(image, DeviceMat) else None, out, np.array(image, copy=True)))
    d = f"pytbd{PID}"
    frame = F.s1_buoy(2, 640, 360)
    normal = np.zeros((360, 640, 3), np.float32)
    with BlockAccessor(d, max_entry_size_bytes=frame.nbytes + normal.nbytes) as w:
        mod = MH.buoy_module(on_frame)([d], MH.buoy_tuners())
        mod._fps = 200
        runner = threading.Thread(target=mod)
        runner.start()
        try:
            t0 = time.time()
            while not log and time.time() - t0 < 30:
                w.write_frame(int(time.monotonic() * 1000), [("forward", frame), ("normal", normal)])
                time.sleep(0.02)
        finally:
            mod.stop()
            runner.join(10)
    assert log
    was_dev, no_host, (threshed, cleaned, contours, (x, y), area), drawn = log[0]
    assert was_dev, "the frame reached the module as a host array: the device read path did not run"
    th = oracle.inrange(np.ascontiguousarray(oracle.bgr2lab(frame)[:, :, 1]), 150, 255)
    k5 = np.ones((5, 5), np.uint8)
    assert np.array_equal(threshed, th) and np.array_equal(cleaned, oracle.morph(oracle.CLOSE, oracle.morph(oracle.OPEN, th, k5), k5))
    exp = oracle.find_contours(th, 0, 2)
    assert len(contours) == len(exp) and all(np.array_equal(p, q) for p, q in zip(contours, exp))
    ref_img = frame.copy()
    draw_contours(ref_img, exp, thickness=10)                  # host rasteriser on a plain array
    assert np.array_equal(drawn, ref_img)


def test_feeder_off_reads_in_the_loop(vp, monkeypatch):
    """VP_FEEDER=0: the same device frames, fetched by the reading thread itself (peek -> DMA -> validate): synchronous, so a frame is
    there at the first read after its write."""
    import vision.core.bindings.camera_message_framework as cmfb
    from vision.devmat import DeviceMat
    monkeypatch.setattr(cmfb, "_FEEDER", False)
    vp.default_context()
    d = f"pytnofeed{PID}"
    a = F.s1_buoy(4, 320, 200)
    with BlockAccessor(d, max_entry_size_bytes=a.nbytes) as w, BlockAccessor(d) as r:
        w.write_frame(5, a)
        st, img, t, _ = r.read_frame_device()
        assert st == ReadStatus.SUCCESS and t == 5 and isinstance(img, DeviceMat) and np.array_equal(img, a) and r._feeder is None
