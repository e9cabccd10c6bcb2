"""GPU suite: posts by DMA (vision.core.posts, csrc/vp_post.hip, cmf_write_begin / cmf_write_commit).

post() of a device image queues one copy by the copy engine from HBM straight into the post block's ring slot and the flush commits the
slot when the copy has arrived (reference: core/base.py:846-876 queues a host copy, :832-839 flushes it through write_frame's memcpy,
lib/camera_message_framework.cpp:306-374).  Checked here: a reader of the reference's layout (read_frame, ModuleReader) receives bit-equal
images; post() hands over the image as it is at the call, whatever happens to it afterwards; same-name posts replace each other;
commits may lag the flush but never overtake each other; a reader racing a full-rate poster never accepts a half-new slot; and the
ingest side keeps delivering frames to a module that holds on to many of them."""
import os
import struct
import threading
import time
import zlib

import numpy as np
import pytest

import frames as F
import module_harness as MH
from vision.core.bindings.camera_message_framework import BLOCK_STUB, BlockAccessor, ReadStatus

pytestmark = pytest.mark.gpu
PID = os.getpid()


def _read(name):
    with BlockAccessor(name) as r:
        st, data, t = r.read_frame()
        assert st == ReadStatus.SUCCESS
        return np.array(data, copy=True), t


def _uid(name):
    with open(BLOCK_STUB + name, "rb") as fh:
        return struct.unpack("<Q", fh.read(8))[0]


@pytest.mark.parametrize("size", [(640, 360), (1920, 1080)])
def test_posts_of_the_buoy_body_reach_a_reference_layout_reader(vp, oracle, size):
    """The red_buoy body's three posts (threshold mask, cleaned mask, frame with the overlay: modules/red_buoy.py:24,31,51) go out by
    DMA - no host copy is made - and what read_frame returns from each block is bit-equal to the oracle's chain and the host rasteriser's
    overlay, with the shape and colour-space tag the reference would have published."""
    from vision.devmat import DeviceMat
    from vision.utils.draw import draw_contours
    w, h = size
    ctx = vp.default_context()
    me = MH.PlainSelf((h, w), True, tag=f"PostT{w}")
    normal = np.zeros((8, 8, 3), np.float32)
    k5 = np.ones((5, 5), np.uint8)
    try:
        for i in range(3):
            frame = F.s1_buoy(i, w, h)
            img = DeviceMat.from_host(ctx, frame)
            MH.buoy_body(me, img, normal)
            assert img._host is None, "posting downloaded the image"
            me.flush(wait=True)
            names = me.block_names()
            assert set(names) == {"threshed#GRAY", "threshed_cleaned#GRAY", "contours#BGR"}
            # idx in the block name = position in the queue at the first post (what ModuleReader sorts the GUI's panes by)
            assert [names[k].split("%")[1] for k in ("threshed#GRAY", "threshed_cleaned#GRAY", "contours#BGR")] == ["0", "1", "2"]
            th = oracle.inrange(np.ascontiguousarray(oracle.bgr2lab(frame)[:, :, 1]), 150, 255)
            got_th, _ = _read(names["threshed#GRAY"])
            got_cl, _ = _read(names["threshed_cleaned#GRAY"])
            got_ct, _ = _read(names["contours#BGR"])
            assert got_th.shape == (h, w, 1) and np.array_equal(got_th[:, :, 0], th)
            assert np.array_equal(got_cl[:, :, 0], oracle.morph(oracle.CLOSE, oracle.morph(oracle.OPEN, th, k5), k5))
            ref = frame.copy()
            draw_contours(ref, oracle.find_contours(th, 0, 2), thickness=10)
            assert got_ct.shape == (h, w, 3) and np.array_equal(got_ct, ref)
            assert _uid(names["contours#BGR"]) == i + 1                    # one frame published per iteration and block
        assert me.queue.dma_posts == 9 and me.queue.host_posts == 0
    finally:
        me.close()


def test_post_hands_over_the_image_as_it_is_at_the_call(vp, oracle):
    """Snapshot semantics (the reference copies at post()): an in-place device write, a host-side write, dropping the image and reusing
    its allocation - none of them, queued after post(), may reach what is published."""
    from vision.devmat import DeviceMat
    from vision.utils.draw import draw_contours
    from vision.utils.color import bgr_to_gray
    ctx = vp.default_context()
    w, h = 1920, 1080
    me = MH.PlainSelf((h, w), True, tag="PostSnap")
    frame = F.s1_buoy(1, w, h)
    square = [np.array([[100, 100], [1800, 100], [1800, 1000], [100, 1000]], np.int32).reshape(-1, 1, 2)]
    try:
        # (1) device-side overwrite right after the post: the module's stream is fenced behind the copy
        img = DeviceMat.from_host(ctx, frame)
        me.post("a", img)
        draw_contours(img, square, thickness=25)                       # vp_draw_polylines_dev into the very buffer being copied
        assert img._host is None
        me.flush(wait=True)
        got, _ = _read(me.block_names()["a#BGR"])
        assert np.array_equal(got, frame)
        drawn = frame.copy()
        draw_contours(drawn, square, thickness=25)
        assert np.array_equal(np.asarray(img), drawn)                  # ... and the draw itself happened
        # (2) host-side write after the post
        img2 = DeviceMat.from_host(ctx, frame)
        me.post("a", img2)
        img2[0:50, 0:50] = 255
        gray_after = bgr_to_gray(img2)[0]                              # forces the re-upload of the written host copy
        me.flush(wait=True)
        got, _ = _read(me.block_names()["a#BGR"])
        assert np.array_equal(got, frame)
        changed = frame.copy()
        changed[0:50, 0:50] = 255
        assert np.array_equal(np.asarray(gray_after), oracle.bgr2gray(changed))
        # (3) the image is dropped and its allocation reused by later operators before the copy is committed
        img3 = DeviceMat.from_host(ctx, frame)
        me.post("a", img3)
        del img3
        others = [DeviceMat.from_host(ctx, np.full_like(frame, 17 * (k + 1))) for k in range(6)]     # same size class: would take the freed buffer
        me.flush(wait=True)
        got, _ = _read(me.block_names()["a#BGR"])
        assert np.array_equal(got, frame)
        assert all(int(np.asarray(o)[5, 5, 0]) == 17 * (k + 1) for k, o in enumerate(others))
        # (4) a post of an image that was still deferred (morphology not launched yet): post() launches it
        from vision.utils.color import range_threshold
        from vision.utils.transform import morph_remove_noise, rect_kernel
        th = range_threshold(bgr_to_gray(DeviceMat.from_host(ctx, frame))[0], 100, 255)
        opened = morph_remove_noise(th, rect_kernel(5))
        me.post("m", opened, "GRAY")
        me.flush(wait=True)
        got, _ = _read(me.block_names()["m#GRAY"])
        exp = oracle.morph(oracle.OPEN, oracle.inrange(oracle.bgr2gray(frame), 100, 255), np.ones((5, 5), np.uint8))
        assert np.array_equal(got[:, :, 0], exp)
    finally:
        me.close()


def test_same_name_replaces_and_commits_keep_their_order(vp):
    """A second post under one name in one iteration replaces the first (one frame is published, the later image); a commit that lags
    its flush is made before the next frame of the same block; drain() publishes what was flushed and gives up what was only queued;
    host arrays and device images can alternate on one block."""
    from vision.devmat import DeviceMat
    ctx = vp.default_context()
    w, h = 1920, 1080
    imgs = [np.full((h, w, 3), v, np.uint8) for v in (10, 20, 30, 40, 50)]
    me = MH.PlainSelf((h, w), True, tag="PostOrder")
    try:
        me.post("x", DeviceMat.from_host(ctx, imgs[0]))
        me.post("x", DeviceMat.from_host(ctx, imgs[1]))
        me.flush(wait=True)
        name = me.block_names()["x#BGR"]
        got, _ = _read(name)
        assert _uid(name) == 1 and np.array_equal(got, imgs[1])
        # flush without waiting, many times in a row: every frame arrives, in order, none twice
        seen = []
        stop = threading.Event()

        def gui():
            with BlockAccessor(name) as r:
                while not stop.is_set():
                    st, data, _ = r.read_frame()
                    if st == ReadStatus.SUCCESS:
                        assert (data == data[0, 0, 0]).all(), "a half-written slot was accepted"
                        seen.append(int(data[0, 0, 0]))
                    time.sleep(0.0002)
        th = threading.Thread(target=gui)
        th.start()
        try:
            for k in range(60):
                me.post("x", DeviceMat.from_host(ctx, imgs[k % 5]))
                me.flush()                                             # whatever has arrived; the rest at the next post / flush
            me.flush(wait=True)
            time.sleep(0.05)
        finally:
            stop.set()
            th.join()
        assert _uid(name) == 61 and me.queue.pending() == 0
        assert seen and seen[-1] == 10 * ((59 % 5) + 1) and all(v in (10, 20, 30, 40, 50) for v in seen)
        # a host array on the same block while a device post is still open, then a device post again
        me.post("x", DeviceMat.from_host(ctx, imgs[2]))
        me.flush()
        me.post("x", imgs[3])
        me.flush(wait=True)
        got, _ = _read(name)
        assert _uid(name) == 63 and np.array_equal(got, imgs[3])
        me.post("x", DeviceMat.from_host(ctx, imgs[4]))
        me.flush(wait=True)
        assert np.array_equal(_read(name)[0], imgs[4]) and _uid(name) == 64
        # drain: a flushed post is published, a queued one is not
        me.post("x", DeviceMat.from_host(ctx, imgs[0]))
        me.flush()
        me.post("y", DeviceMat.from_host(ctx, imgs[1]))
        yname = me.block_names()["y#BGR"]
        me.queue.drain()
        assert _uid(name) == 65 and np.array_equal(_read(name)[0], imgs[0]) and _uid(yname) == 0
        me.post("y", DeviceMat.from_host(ctx, imgs[2]))                # the abandoned slot is reused
        me.flush(wait=True)
        assert _uid(yname) == 1 and np.array_equal(_read(yname)[0], imgs[2])
        # an image larger than the block was made for: the error write_frame raises
        with pytest.raises(RuntimeError, match="larger than the block"):
            me.post("y", DeviceMat.from_host(ctx, np.zeros((h + 1, w, 3), np.uint8)))
    finally:
        me.close()


def test_a_reader_racing_a_full_rate_poster_never_accepts_a_half_new_slot(vp):
    """Old-reader safety under load: one thread posts four distinct 1080p images round robin as fast as it can (copies by DMA, commits
    deferred), a reference-layout reader copies slots with read_frame all the while: every frame it accepts is bit-equal to one of
    the four (CRC) and carries the time stamp order of its writes."""
    from vision.devmat import DeviceMat
    ctx = vp.default_context()
    w, h = 1920, 1080
    rng = np.random.default_rng(11)
    pool = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(4)]
    for k, p in enumerate(pool):
        p[0, 0, 0] = k
    sums = [zlib.crc32(p) for p in pool]
    dev = [DeviceMat.from_host(ctx, p) for p in pool]
    me = MH.PlainSelf((h, w), True, tag="PostRace")
    bad, accepted = [], []
    stop = threading.Event()
    try:
        me.post("r", dev[0])
        me.flush(wait=True)
        name = me.block_names()["r#BGR"]

        def gui():
            last = 0
            with BlockAccessor(name) as r:
                while not stop.is_set():
                    st, data, t = r.read_frame()
                    if st != ReadStatus.SUCCESS:
                        continue
                    k = int(data[0, 0, 0])
                    if k > 3 or zlib.crc32(data) != sums[k] or t < last:
                        bad.append((k, t))
                    last = t
                    accepted.append(k)
        th = threading.Thread(target=gui)
        th.start()
        try:
            t_end = time.time() + 3.0
            n = 0
            while time.time() < t_end:
                n += 1
                me.post("r", dev[n % 4])
                me.flush()
            me.flush(wait=True)
        finally:
            stop.set()
            th.join()
        assert not bad, bad[:5]
        assert len(accepted) >= 10 and _uid(name) == n + 1
        print(f"{n} posts in 3 s ({n / 3.0:.0f}/s), {len(accepted)} accepted by the reader")
    finally:
        me.close()


def test_module_posts_on_the_runtime_reach_module_reader(vp, oracle):
    """A module on the runtime with posts on (the reference's default): ModuleReader - the GUI's end, read_frame on every post block -
    receives the threshold mask, the cleaned mask and the overlay frame of the buoy harness, each bit-equal to the oracle's result for
    the frame that was written; the loop publishes lagging posts in its idle time and drains them when it stops."""
    from vision.core.base import ModuleReader
    from vision.utils.draw import draw_contours
    MH.module_argv()
    d = f"pytpost{PID}"
    frame = F.s1_buoy(2, 640, 360)
    normal = np.zeros((360, 640, 3), np.float32)
    got = {}
    done = []
    with BlockAccessor(d, max_entry_size_bytes=frame.nbytes + normal.nbytes) as wblk:
        mod = MH.buoy_module(lambda *a: done.append(1))([d], MH.buoy_tuners())
        mod._fps = 200
        runner = threading.Thread(target=mod)
        runner.start()
        reader = None
        try:
            t0 = time.time()
            while len(done) < 3 and time.time() - t0 < 30:
                wblk.write_frame(int(time.monotonic() * 1000), [("forward", frame), ("normal", normal)])
                time.sleep(0.02)
            assert len(done) >= 3
            reader = ModuleReader(mod._name)
            reader.register_post_udl(lambda module, name, idx, image, cs: got.__setitem__(name, (idx, np.array(image, copy=True), cs)))
            reader.run_forever(fps=200)
            t0 = time.time()
            while len(got) < 3 and time.time() - t0 < 10:
                wblk.write_frame(int(time.monotonic() * 1000), [("forward", frame), ("normal", normal)])
                time.sleep(0.02)
            assert mod._posts.dma_posts >= 9 and mod._posts.host_posts == 0
        finally:
            if reader is not None:
                reader.unblock()
            mod.stop()
            runner.join(10)
    assert set(got) == {"threshed", "threshed_cleaned", "contours"}
    th = oracle.inrange(np.ascontiguousarray(oracle.bgr2lab(frame)[:, :, 1]), 150, 255)
    k5 = np.ones((5, 5), np.uint8)
    assert got["threshed"][0] == 0 and got["threshed"][2] == "GRAY" and np.array_equal(got["threshed"][1][:, :, 0], th)
    assert got["threshed_cleaned"][0] == 1 and np.array_equal(got["threshed_cleaned"][1][:, :, 0], oracle.morph(oracle.CLOSE, oracle.morph(oracle.OPEN, th, k5), k5))
    ref = frame.copy()
    draw_contours(ref, oracle.find_contours(th, 0, 2), thickness=10)
    assert got["contours"][0] == 2 and got["contours"][2] == "BGR" and np.array_equal(got["contours"][1], ref)
    assert mod._posts.pending() == 0 and not mod._posts._open


def test_dma_posts_off_takes_the_host_path(vp, monkeypatch):
    """VP_DMA_POSTS=0: the download + write_frame path of round 3, same bytes in the block."""
    import vision.core.posts as posts
    from vision.devmat import DeviceMat
    monkeypatch.setattr(posts, "_DMA_POSTS", False)
    ctx = vp.default_context()
    frame = F.s1_buoy(0, 320, 200)
    me = MH.PlainSelf((200, 320), True, tag="PostHost")
    try:
        me.post("h", DeviceMat.from_host(ctx, frame))
        me.flush()
        assert me.queue.host_posts == 1 and me.queue.dma_posts == 0
        assert np.array_equal(_read(me.block_names()["h#BGR"])[0], frame)
    finally:
        me.close()


def test_a_module_that_keeps_many_frames_still_gets_new_ones(vp):
    """In the reference every frame is a private copy a module may keep for good (core/base.py:765-768: history deques, prev_frame).
    The feeder has four device buffers: beyond two held by the module, frames are handed out in allocations of the module's own, so
    the stream of frames never stalls on what the module holds - and every held frame keeps its contents."""
    vp.default_context()
    d = f"pytkeep{PID}"
    w, h = 640, 360
    frames = [np.full((h, w, 3), 10 + k, np.uint8) for k in range(12)]
    held = []
    with BlockAccessor(d, max_entry_size_bytes=frames[0].nbytes) as wr, BlockAccessor(d) as r:
        for k, f in enumerate(frames):
            wr.write_frame(100 + k, f)
            t0 = time.time()
            while True:
                st, img, t, _ = r.read_frame_device()
                if st == ReadStatus.SUCCESS and t == 100 + k:
                    break
                assert time.time() - t0 < 2.0, f"frame {k} never arrived with {len(held)} frames held"
                time.sleep(0.0005)
            held.append(img)                                           # the module keeps every frame
        assert r._feeder is not None and r._feeder.out <= 2
        for k, img in enumerate(held):
            assert np.array_equal(np.asarray(img), frames[k]), f"held frame {k} changed"


def test_posts_of_frame_planes_smaller_images_and_grey(vp):
    """Shapes a module can post: a plane of a multi-plane frame (an image at an OFFSET inside the frame's one device allocation), a
    2-D mask, an image smaller than the one the block was made for (the block keeps its size, the frame's metadata follow the image),
    and an int16 device image (not uint8: converted on the host like any other array, as the reference's np.array(image, np.uint8))."""
    from vision.devmat import DeviceMat
    ctx = vp.default_context()
    d = f"pytplanes{PID}"
    a = F.s1_buoy(0, 320, 200)
    depth = (np.arange(200 * 320, dtype=np.uint8).reshape(200, 320, 1) * 3).astype(np.uint8)
    me = MH.PlainSelf((200, 320), True, tag="PostPlanes")
    try:
        with BlockAccessor(d, max_entry_size_bytes=a.nbytes + depth.nbytes) as w, BlockAccessor(d) as r:
            w.write_frame(3, [("forward", a), ("depth", depth)])
            t0 = time.time()
            while True:
                st, data, _, _ = r.read_frame_device()
                if st == ReadStatus.SUCCESS or time.time() - t0 > 2:
                    break
                time.sleep(0.001)
            assert st == ReadStatus.SUCCESS
            fwd, dep = data
            assert isinstance(dep, DeviceMat) and dep._off == a.nbytes          # the second plane starts behind the first
            me.post("plane", dep, "GRAY")
            me.post("whole", fwd)
            me.flush(wait=True)
            got, _ = _read(me.block_names()["plane#GRAY"])
            assert got.shape == (200, 320, 1) and np.array_equal(got, depth)
            assert np.array_equal(_read(me.block_names()["whole#BGR"])[0], a)
        # a smaller image into the same block, then a 2-D one
        small = DeviceMat.from_host(ctx, np.ascontiguousarray(a[:50, :64]))
        me.post("whole", small)
        me.flush(wait=True)
        got, _ = _read(me.block_names()["whole#BGR"])
        assert got.shape == (50, 64, 3) and np.array_equal(got, a[:50, :64])
        mask = DeviceMat.from_host(ctx, np.ascontiguousarray(a[:, :, 0]))
        me.post("whole", mask)
        me.flush(wait=True)
        got, _ = _read(me.block_names()["whole#BGR"])
        assert got.shape == (200, 320, 1) and np.array_equal(got[:, :, 0], a[:, :, 0])
        # not uint8 on the device: the host conversion path
        i16 = DeviceMat.from_host(ctx, (a[:, :, 0].astype(np.int16) - 3))
        before = me.queue.host_posts
        me.post("i16", i16)
        me.flush(wait=True)
        assert me.queue.host_posts == before + 1
        assert np.array_equal(_read(me.block_names()["i16#BGR"])[0][:, :, 0], np.array(a[:, :, 0].astype(np.int16) - 3, np.uint8))
    finally:
        me.close()


def test_an_image_posted_over_and_over_does_not_collect_readers(vp):
    """A long-lived image posted at every iteration (a static overlay) must not accumulate pending-reader entries."""
    from vision.devmat import DeviceMat
    ctx = vp.default_context()
    img = DeviceMat.from_host(ctx, F.s1_buoy(0, 320, 200))
    me = MH.PlainSelf((200, 320), True, tag="PostStatic")
    try:
        for _ in range(200):
            me.post("s", img)
            me.flush(wait=True)
        assert len(img._consumers) <= 10
        assert np.array_equal(_read(me.block_names()["s#BGR"])[0], F.s1_buoy(0, 320, 200)) and _uid(me.block_names()["s#BGR"]) == 200
    finally:
        me.close()
