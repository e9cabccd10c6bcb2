"""CPU suite: the module runtime (vision.core.*) — source-string parsing, tuner wire format, the dispatch rules of
the loop (process() vs @sources handlers, alias resolution, copies), posts read back through ModuleReader,
normalisation arithmetic, handlers, retry after a source is deleted, capture-source harness.  Behaviours are
written from the reference lines cited in each test (the reference has no tests of its own)."""
import os
import struct
import sys
import threading
import time

import numpy as np
import pytest

from vision.core.bindings.camera_message_framework import BLOCK_STUB, BlockAccessor
from vision.core.tuners import BoolTuner, DoubleTuner, IntTuner

PID = os.getpid()


@pytest.fixture(autouse=True)
def _argv(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["module.py"])      # ModuleBase parses sys.argv itself (core/base.py:599-635)


def _wait(cond, timeout=5.0):
    t0 = time.time()
    n = 0
    while time.time() - t0 < timeout:
        if cond():
            return True
        n += 1
        time.sleep(0.003 + 0.001 * (n % 7))      # uneven on purpose: a fixed 10 ms poll reads a 200 fps, 4-frame loop at every other frame for ever
    return False


def test_video_source_strings():
    from vision.core.base import VideoSource, sources
    v = VideoSource.create("zed[forward, normal]:i8:u32")         # core/base.py:68-110
    assert v.name == "zed" and v.plane_aliases == ("forward", "normal")
    assert (v.byte_type, v.short_type, v.long_type) == (np.int8, np.uint32, np.float64)
    v = VideoSource.create("forward")
    assert (v.name, v.byte_type, v.short_type, v.long_type, v.plane_aliases) == ("forward", np.uint8, np.float32, np.float64, ())
    assert VideoSource.create("d:f64").long_type == np.float64 and VideoSource.create("d:i64:i32").short_type == np.int32
    assert VideoSource.create(v) is v

    @sources("zed[forward]", "zed[normal]", "downward")            # core/base.py:123-149
    def f(a, b, c):
        pass
    assert f._sources_aliases == ("forward", "normal", "downward")


def test_tuner_wire_format():
    t = IntTuner("thresh_min", 7, 0, 255)                          # core/tuners.py:49-79: '{n}siii'
    assert t.serialize() == struct.pack("10siii", b"thresh_min", 7, 0, 255) and t.byte_size() == struct.calcsize("10siii")
    assert str(t) == "IntTuner_thresh_min"
    t.deserialize(struct.pack("10siii", b"thresh_min", 99, 0, 255))
    assert t.value == 99
    t.deserialize(struct.pack("10siii", b"thresh_min", 999, 0, 1000))   # validator keeps the construction-time range
    assert t.value == 99
    d = DoubleTuner("gain", 1.5)                                   # :82-112: '{n}sddd'
    assert d.serialize() == struct.pack("4sddd", b"gain", 1.5, -10000, 10000)
    b = BoolTuner("on", True)                                      # :115-135: '{n}s?'
    assert b.serialize() == struct.pack("2s?", b"on", True)
    b.deserialize(struct.pack("2s?", b"on", False))
    assert b.value is False
    with pytest.raises(AssertionError):
        IntTuner("has space", 1)
    with pytest.raises(AssertionError):
        IntTuner("x", 1, 5, 2)
    assert IntTuner("a", 1) == IntTuner("a", 2) and IntTuner("a", 1) != DoubleTuner("a", 1.0)


def test_normalize_arithmetic():
    from vision.core.base import VideoSourceMetadata
    m = VideoSourceMetadata()
    m.update(np.zeros((1080, 1920, 3), np.uint8), 0)
    # core/base.py:553-574: both axes are divided by the WIDTH
    assert m.normalize_coord((540, 960)) == (0.0, 0.0)
    assert m.normalize_coord((0, 0)) == ((0 - 540) / 1920, (0 - 960) / 1920)
    assert m.normalize_axis(1920, 0) == 0.5 and m.normalize_axis(1080, 1) == 540 / 1920
    assert m.mark_as_dead() is True and m.mark_as_dead() is False
    a, b = VideoSourceMetadata(), VideoSourceMetadata()
    a.update(np.zeros((2, 2)), 0)
    assert len(b._acquisition_times) == 0                           # windows are per instance here


def _module_class():
    from vision.core.base import ModuleBase, sources
    from vision.core.handlers import HandlerBase, HandlerMixin

    class Echo(HandlerBase):
        def process(self, direction, image, *a, **k):
            self.post("echo", image, "gray")
            return self.normalize((0, 0)), self.tuners["gain"]

    class Demo(ModuleBase, HandlerMixin):
        def __init__(self, srcs, tuners):
            ModuleBase.__init__(self, srcs, tuners, fps=200)
            HandlerMixin.__init__(self, [Echo("echo")])
            self.single, self.pairs, self.stray = [], [], []

        def process(self, direction, image):
            assert threading.current_thread() is not threading.main_thread()
            (self.single if direction.startswith("pytfwd") else self.stray).append((direction, image))
            image[0, 0, 0] = 255                                     # frames are writable private copies
            self.post("seen", image[:, :, 0], "GRAY")
            self.extra = self.handlers["echo"].process(direction, image[:, :, 0])

        @sources("zed[forward]", "zed[normal]")
        def both(self, fwd, normal):
            self.pairs.append((fwd.copy(), normal.copy(), self.tuners["thresh"]))
    return Demo


def test_loop_dispatch_posts_tuners_and_reader():
    from vision.core.base import ModuleReader
    Demo = _module_class()
    fwd, zed = f"pytfwd{PID}", f"pytzed{PID}"   # no underscores: get_active_modules splits on "_" (core/base.py:362-365)
    img = np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3)
    nrm = np.ones((4, 6, 3), np.float32) * 0.5
    dep = np.ones((4, 6), np.float32)
    with BlockAccessor(fwd, max_entry_size_bytes=img.nbytes) as wf, \
            BlockAccessor(zed, max_entry_size_bytes=img.nbytes + nrm.nbytes + dep.nbytes) as wz:
        mod = Demo([fwd, f"{zed}[forward,normal,depth]"], [IntTuner("thresh", 150), DoubleTuner("gain", 2.0)])
        assert mod._name == f"Demo-on-{fwd}-{zed}"                  # core/base.py:646-648
        runner = threading.Thread(target=mod)
        runner.start()
        try:
            wf.write_frame(10, img)
            assert _wait(lambda: len(mod.single) == 1)
            d, got = mod.single[0]
            assert d == fwd and got.shape == (4, 6, 3) and got[0, 0, 0] == 255 and np.array_equal(got.ravel()[1:], img.ravel()[1:])
            assert img[0, 0, 0] == 0                                # the module worked on a copy
            assert mod.extra == (((0 - 2) / 6, (0 - 3) / 6), 2.0)   # handler borrowed normalize / tuners
            # block-provided plane names win over the [alias] list (core/base.py:775-780): "depth" has no handler -> process()
            wz.write_frame(11, [("forward", img), ("normal", nrm), ("depth", dep)])
            assert _wait(lambda: len(mod.pairs) == 1 and len(mod.stray) == 1)
            f2, n2, thr = mod.pairs[0]
            assert np.array_equal(f2, img) and np.array_equal(n2, nrm) and thr == 150
            assert mod.stray[0][0] == "depth" and mod.stray[0][1].shape == (4, 6, 1)
            # unnamed planes fall back to the alias list of the source string
            wz.write_frame(12, [img, nrm * 2, dep])
            assert _wait(lambda: len(mod.pairs) == 2) and np.array_equal(mod.pairs[1][1], nrm * 2)
            # posts are readable by the GUI side, with their colour space and index (core/base.py:832-839)
            name = mod._name
            assert _wait(lambda: os.path.exists(f"{BLOCK_STUB}module_{name}_post%0%seen#GRAY"))
            assert name in ModuleReader.get_active_modules()
            reader = ModuleReader(name)
            posts, tuned = [], []
            reader.register_post_udl(lambda m, n, i, data, cs: posts.append((n, i, cs, data.copy())))
            reader.register_tuner_udl(lambda m, n, i, t: tuned.append((n, t.value)))
            reader.run_forever(fps=200)
            try:
                wf.write_frame(13, img)
                assert _wait(lambda: any(p[0] == "seen" for p in posts) and any(p[0] == "echo" for p in posts))
                seen = next(p for p in posts if p[0] == "seen")
                assert seen[1] == 0 and seen[2] == "GRAY" and seen[3].shape[:2] == (4, 6)
                assert next(p for p in posts if p[0] == "echo")[1:3] == (1, "GRAY")
                # a tuner edit from the GUI side reaches the module (core/base.py:423-428, :246-253)
                reader.update_tuner_value("thresh", 42)
                wz.write_frame(14, [("forward", img), ("normal", nrm), ("depth", dep)])
                assert _wait(lambda: len(mod.pairs) >= 3 and mod.pairs[-1][2] == 42)
                with pytest.raises(RuntimeError):
                    mod.post("bad%name", img)
            finally:
                reader.unblock()
        finally:
            mod.stop()
            runner.join(5)
    assert not runner.is_alive()


def test_module_reenters_after_source_deletion():
    """FRAMEWORK_DELETED on a source -> RuntimeError in read_messages -> _retry -> the manager is re-entered and waits for
    the source to come back (core/base.py:746-752, :691-707; binding :399-413 polls once per second)."""
    from vision.core.base import ModuleBase
    name = f"pyt_gone_{PID}"
    seen = []

    class Plain(ModuleBase):
        def process(self, direction, image):
            seen.append(int(image[0, 0, 0]))

    w = BlockAccessor(name, max_entry_size_bytes=12)
    w.__enter__()
    mod = Plain([name], [], fps=200)
    t = threading.Thread(target=mod)
    t.start()
    try:
        w.write_frame(1, np.full((2, 2, 3), 1, np.uint8))
        assert _wait(lambda: seen == [1])
        w.__exit__(None, None, None)                                 # creator leaves: block deleted + unlinked
        time.sleep(0.2)
        w = BlockAccessor(name, max_entry_size_bytes=12)
        w.__enter__()
        time.sleep(1.3)                                              # reopen poll period
        w.write_frame(2, np.full((2, 2, 3), 2, np.uint8))
        assert _wait(lambda: seen == [1, 2], timeout=5)
    finally:
        mod.stop()
        t.join(5)
        w.__exit__(None, None, None)


def test_class_name_and_performance_mode(monkeypatch):
    from vision.core.base import ModuleBase

    class Bad_Name(ModuleBase):
        pass
    with pytest.raises(RuntimeError):
        Bad_Name(["x"], [])                                          # core/base.py:637-640
    monkeypatch.setattr(sys, "argv", ["m.py", "--enable-performance", "-f", "30", "other:f64"])

    class Fast(ModuleBase):
        pass
    m = Fast(["ignored"], [])
    assert m._name == "Fast-on-other" and m._fps == 30
    m.post("anything", np.zeros((2, 2), np.uint8))                  # no-op in performance mode (:857)
    assert len(m._post_queue) == 0


def test_capture_source_harness_and_image_directory(tmp_path):
    from PIL import Image
    from vision.capture_sources.image_directory import ImageDirectory
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 255, (20, 30, 3), dtype=np.uint8) for _ in range(3)]
    for i, im in enumerate(imgs):
        Image.fromarray(im[:, :, ::-1]).save(tmp_path / f"f{i:02d}.png")   # file holds RGB; the source yields BGR
    d = f"pyt_dir_{PID}"
    src = ImageDirectory(d, str(tmp_path), fps=200)
    t = threading.Thread(target=src.run_event_loop)
    t.start()
    try:
        got = {}
        with BlockAccessor(d) as r:
            def pump():
                st, data, _ = r.read_frame()
                if data is not None:
                    for k, im in enumerate(imgs):
                        if np.array_equal(data, im):
                            got[k] = True
                return len(got) == 3
            assert _wait(pump, timeout=5)
    finally:
        src._quit_flag.set()
        t.join(5)
        src.close()
    assert not os.path.exists(BLOCK_STUB + d)


def test_logical_udl_that_returns_leaves_the_source_running():
    """A logical UDL that returns normally (one-shot hardware set-up) does NOT stop the source; one that raises does, and so does a
    capture UDL that is exhausted (core/capture_source.py:113-127 sets the quit flag in `except` only; :147-170 after the loop)."""
    from vision.core.capture_source import CaptureSource
    d = f"pyt_logical_{PID}"
    src = CaptureSource()
    ran = []
    src.register_logical_udl(lambda limiter, args: ran.append(args), ("set-up",))
    frame = np.zeros((4, 4, 3), np.uint8)

    def capture(limiter, args):
        for n, t in enumerate(limiter.rate(200)):
            frame[0, 0, 0] = n % 251
            yield d, t, frame
    src.register_capture_udl("cam", capture)
    t = threading.Thread(target=src.run_event_loop)
    t.start()
    try:
        assert _wait(lambda: bool(ran), timeout=5)
        seen = set()
        with BlockAccessor(d) as r:
            def pump():
                st, data, _ = r.read_frame()
                if data is not None:
                    seen.add(int(data[0, 0, 0]))
                return len(seen) >= 5                       # frames keep coming long after the logical UDL has returned
            assert _wait(pump, timeout=5)
        assert not src._quit_flag.is_set() and t.is_alive()
    finally:
        src._quit_flag.set()
        t.join(5)
        src.close()
    # a logical UDL that raises stops everything
    src2 = CaptureSource()

    def broken(limiter, args):
        raise ValueError("boom")
    src2.register_logical_udl(broken)
    t2 = threading.Thread(target=src2.run_event_loop)
    t2.start()
    t2.join(5)
    assert not t2.is_alive() and src2._quit_flag.is_set()
    src2.close()


def test_video_capture_source(tmp_path):
    """capture_sources/video.py:9-39: one decoded frame per tick fanned out to every listed direction."""
    from vision.capture_sources.video import Video
    frames = np.random.default_rng(3).integers(0, 255, (4, 12, 16, 3), dtype=np.uint8)
    path = tmp_path / "clip.npy"
    np.save(path, frames)
    a, b = f"pytva{PID}", f"pytvb{PID}"
    src = Video(str(path), [a, b], fps=200, loop=True)
    t = threading.Thread(target=src.run_event_loop)
    t.start()
    try:
        seen = {a: set(), b: set()}
        with BlockAccessor(a) as ra, BlockAccessor(b) as rb:
            def pump():
                for name, r in ((a, ra), (b, rb)):
                    st, data, _ = r.read_frame()
                    if data is not None:
                        for k in range(4):
                            if np.array_equal(data, frames[k]):
                                seen[name].add(k)
                return len(seen[a]) == 4 and len(seen[b]) == 4
            assert _wait(pump, timeout=5)
    finally:
        src._quit_flag.set()
        t.join(5)
        src.close()


def test_overlay_helpers_and_decode_normal():
    """The drawing helpers the handlers import (handlers/bins.py, gate.py: draw_rect / draw_circle / draw_text) and
    utils/transform.py decode_normal (modules/normal.py:26): in-place host stand-ins, not part of the accelerated path."""
    from vision.utils import draw, transform
    m = np.zeros((40, 60, 3), np.uint8)
    draw.draw_rect(m, (5, 5), (20, 15), (0, 255, 0), 1)
    assert m[5, 5:21, 1].min() == 255 and m[15, 5:21, 1].min() == 255 and m[5:16, 5, 1].min() == 255 and m[10, 10].sum() == 0
    draw.draw_rect(m, (30, 5), (35, 8), (7, 7, 7), -1)
    assert (m[5:9, 30:36] == 7).all()
    draw.draw_circle(m, (40, 25), 8, (255, 0, 0), 1)
    assert m[25, 48, 0] == 255 and m[25, 32, 0] == 255 and m[17, 40, 0] == 255 and m[25, 40].sum() == 0
    before = m.copy()
    draw.draw_text(m, "label", (2, 38), 0.5, (0, 0, 255))
    assert (m != before).any()
    draw.draw_circle(m, (1000, 1000), 5)                     # fully outside: no exception
    n = transform.decode_normal(np.array([[[0, 255, 128]]], np.uint8))
    assert n.dtype == np.float32 and n[0, 0, 0] == -1.0 and n[0, 0, 1] == 1.0 and abs(n[0, 0, 2] - (128 / 255 * 2 - 1)) < 1e-6
