"""Images that stay in HBM between operator calls (vision/devmat.py): what the `vision.utils` mirror returns must behave like the
caller-owned, writable numpy arrays of the reference (utils/color.py:11-32 ... modules/red_buoy.py:39) whichever side holds the data,
and must not move it before Python looks."""
import threading

import numpy as np
import pytest

import frames as F

pytestmark = pytest.mark.gpu


def _chain(image):
    from vision.utils.color import bgr_to_lab, range_threshold
    from vision.utils.feature import outer_contours
    from vision.utils.transform import morph_close_holes, morph_remove_noise, rect_kernel
    lab, (l_, a_, b_) = bgr_to_lab(image)
    th = range_threshold(a_, 150, 255)
    k = rect_kernel(5)
    cl = morph_close_holes(morph_remove_noise(th, k), k)
    return lab, a_, th, cl, outer_contours(th)


def test_chain_stays_on_the_device_and_matches_the_oracle(vp, oracle):
    from vision.devmat import DeviceMat
    img = F.s1_buoy(3, 640, 360)
    lab, a_, th, cl, cs = _chain(img)
    for m in (lab, a_, th, cl):
        assert isinstance(m, DeviceMat) and m._host is None, "an intermediate image was downloaded although nothing read it"
    assert th.binary and cl.binary
    olab = oracle.bgr2lab(img)
    oth = oracle.inrange(np.ascontiguousarray(olab[:, :, 1]), 150, 255)
    k5 = np.ones((5, 5), np.uint8)
    ocl = oracle.morph(oracle.CLOSE, oracle.morph(oracle.OPEN, oth, k5), k5)
    exp = oracle.find_contours(oth, 0, 2)
    assert len(cs) == len(exp) and all(isinstance(c, np.ndarray) and np.array_equal(c, e) for c, e in zip(cs, exp))
    assert np.array_equal(lab, olab) and np.array_equal(a_, olab[:, :, 1]) and np.array_equal(th, oth) and np.array_equal(cl, ocl)
    assert lab.shape == (360, 640, 3) and th.shape == (360, 640) and th.dtype == np.uint8 and th.ndim == 2 and len(th) == 360


def test_host_side_writes_are_seen_by_the_next_operator(vp, oracle):
    from vision.utils.color import range_threshold
    from vision.utils.transform import dilate, rect_kernel
    g = np.ascontiguousarray(F.s1_buoy(1, 320, 200)[:, :, 2])
    k3 = np.ones((3, 3), np.uint8)
    ref = oracle.inrange(g, 150, 255)
    # __setitem__ on the result
    th = range_threshold(g, 150, 255)
    th[10:20, 30:60] = 255
    ref1 = ref.copy(); ref1[10:20, 30:60] = 255
    assert np.array_equal(dilate(th, rect_kernel(3)), oracle.morph(oracle.DILATE, ref1, k3))
    # a writable numpy view handed out by np.asarray, mutated behind the object's back
    th = range_threshold(g, 150, 255)
    view = np.asarray(th)
    assert isinstance(view, np.ndarray) and view.flags.writeable
    view[100:110, :] = 255
    ref2 = ref.copy(); ref2[100:110, :] = 255
    assert np.array_equal(dilate(th, rect_kernel(3)), oracle.morph(oracle.DILATE, ref2, k3))
    # in-place operators and out=
    th = range_threshold(g, 150, 255)
    th |= 1
    np.bitwise_and(th, 0xF0, out=th)
    assert np.array_equal(th, (ref | 1) & 0xF0)
    assert np.array_equal(dilate(th, rect_kernel(3)), oracle.morph(oracle.DILATE, (ref | 1) & 0xF0, k3))   # a grey image now: generic path


def test_behaves_like_an_array(vp, oracle):
    from vision.utils.color import bgr_to_gray, range_threshold
    img = F.s2_bins(0, 200, 120)
    g, (g1,) = bgr_to_gray(img)
    og = oracle.bgr2gray(img)
    assert np.array_equal(g1, og) and g1 is not g
    th = range_threshold(g, 20, 255)
    oth = oracle.inrange(og, 20, 255)
    assert int((th > 0).sum()) == int((oth > 0).sum()) and int(np.count_nonzero(th)) == int(np.count_nonzero(oth))
    assert np.array_equal(th[5:9, 7], oth[5:9, 7]) and th[3, 4] == oth[3, 4]
    assert np.array_equal(th.copy(), oth) and np.array_equal(th.astype(np.float32), oth.astype(np.float32))
    assert np.array_equal(~th, ~oth) and np.array_equal(th & g, oth & og) and np.array_equal(255 - th, 255 - oth)
    assert th.T.shape == (200, 120) and th.flags.c_contiguous and th.mean() == oth.mean()
    assert np.array_equal(np.dstack([th, th, th]), np.dstack([oth, oth, oth]))
    assert np.array_equal(np.where(th)[0], np.where(oth)[0])
    rows = [r for r in th]
    assert len(rows) == 120 and np.array_equal(rows[7], oth[7])
    assert "DeviceMat" in repr(range_threshold(g, 20, 255))


def test_plain_numpy_mode(vp, oracle):
    from vision import devmat
    from vision.utils.color import bgr_to_lab, range_threshold
    img = F.s1_buoy(0, 160, 90)
    devmat.set_lazy(False)
    try:
        lab, planes = bgr_to_lab(img)
        th = range_threshold(planes[1], 150, 255)
        assert type(lab) is np.ndarray and all(type(p) is np.ndarray for p in planes) and type(th) is np.ndarray
        assert np.array_equal(lab, oracle.bgr2lab(img))
        # a DeviceMat made earlier is still accepted
        devmat.set_lazy(True)
        d = range_threshold(planes[1], 150, 255)
        devmat.set_lazy(False)
        from vision.utils.transform import erode, rect_kernel
        assert np.array_equal(erode(d, rect_kernel(3)), oracle.morph(oracle.ERODE, th, np.ones((3, 3), np.uint8)))
    finally:
        devmat.set_lazy(True)


def test_image_made_on_another_thread(vp, oracle):
    """Contexts are per thread (ModuleBase runs process() on its own thread): an image produced under one context is usable under
    another (it travels through the host)."""
    from vision.utils.color import range_threshold
    from vision.utils.transform import dilate, rect_kernel
    g = np.ascontiguousarray(F.s1_buoy(5, 200, 100)[:, :, 2])
    box = {}
    t = threading.Thread(target=lambda: box.update(th=range_threshold(g, 150, 255)))
    t.start(); t.join()
    out = dilate(box["th"], rect_kernel(3))
    assert np.array_equal(out, oracle.morph(oracle.DILATE, oracle.inrange(g, 150, 255), np.ones((3, 3), np.uint8)))


def test_facade_and_drawing_take_device_images(vp, oracle):
    from vision import cv2_facade as cv2
    from vision.utils.draw import draw_contours
    img = F.s2_bins(1, 320, 180)
    hsv = cv2.cvtColor(img, cv2.COLOR_BGR2HSV)
    mask = cv2.inRange(hsv, np.array([10, 20, 60]), np.array([30, 100, 255]))
    opened = cv2.morphologyEx(mask, cv2.MORPH_OPEN, cv2.getStructuringElement(cv2.MORPH_RECT, (5, 5)))
    cs, _ = cv2.findContours(opened, cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)
    ohsv = oracle.bgr2hsv(img)
    omask = oracle.inrange(ohsv, (10, 20, 60), (30, 100, 255))
    oopen = oracle.morph(oracle.OPEN, omask, np.ones((5, 5), np.uint8))
    exp = oracle.find_contours(oopen, 0, 2)
    assert len(cs) == len(exp) and all(np.array_equal(a, b) for a, b in zip(cs, exp))
    vis = cv2.cvtColor(mask, cv2.COLOR_GRAY2BGR)
    over = cv2.addWeighted(img, 0.7, vis, 0.3, 0)
    assert over.shape == img.shape
    n, lab, st, ce = cv2.connectedComponentsWithStats(opened, 8, cv2.CV_32S)
    on, olab, ost, oce = oracle.ccl(oopen, 2)
    assert n == on and np.array_equal(lab, olab) and np.array_equal(st, ost)
    # drawing into a device image materialises it and draws on the host copy
    a = img.copy(); b = img.copy()
    draw_contours(a, cs, thickness=3)
    from vision.utils import draw as D
    for c in cs:                                   # the pure-Python rasteriser of the same statements
        pts = np.asarray(c, np.int64).reshape(-1, 2)
        for i in range(len(pts)):
            D._line(b, pts[i], pts[(i + 1) % len(pts)], np.asarray((0, 0, 255), np.uint8), 3)
    assert np.array_equal(a, b)
    draw_contours(vis, cs, thickness=2)
    assert isinstance(np.asarray(vis), np.ndarray)


def test_polygon_sums_equal_the_sequential_loop(vp):
    from vision.utils import feature
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 17, 500):
        c = rng.integers(0, 1920, (n, 1, 2)).astype(np.int32)
        assert feature._polygon_moments(c) == feature._polygon_moments_float(c.reshape(-1, 2).astype(np.float64))
    sq = np.array([[[10, 10]], [[10, 30]], [[50, 30]], [[50, 10]]], np.int32)
    assert feature.contour_area(sq) == 800.0 and feature.contour_centroid(sq) == (30, 20)


def test_operators_between_device_images_run_when_needed(vp, oracle):
    """Morphology on a device image is launched when its result is needed (or just before its input changes), never earlier; the
    result is what an immediate launch gives."""
    from vision import devmat
    from vision.utils.color import range_threshold
    from vision.utils.transform import dilate, erode, morph_close_holes, morph_remove_noise, rect_kernel
    g = np.ascontiguousarray(F.s1_buoy(4, 320, 200)[:, :, 2])
    k5, k3 = np.ones((5, 5), np.uint8), np.ones((3, 3), np.uint8)
    oth = oracle.inrange(g, 150, 255)
    th = range_threshold(g, 150, 255)
    a = morph_remove_noise(th, rect_kernel(5))
    b = morph_close_holes(a, rect_kernel(5))                 # a pending result as the input of the next
    assert a._pending is not None and b._pending is not None and a.binary and b.binary and a.shape == th.shape
    assert "not computed" in repr(b)
    ob = oracle.morph(oracle.CLOSE, oracle.morph(oracle.OPEN, oth, k5), k5)
    assert np.array_equal(b, ob) and a._pending is None and b._pending is None
    assert np.array_equal(a, oracle.morph(oracle.OPEN, oth, k5))
    # the input is written to on the host after the call: the result is that of the contents at the call
    th = range_threshold(g, 150, 255)
    d = dilate(th, rect_kernel(3))
    e = erode(d, rect_kernel(3))
    th[20:60, 30:90] = 255
    assert d._pending is None and e._pending is not None      # d had to run before th changed; e reads d, which did not change
    assert np.array_equal(d, oracle.morph(oracle.DILATE, oth, k3))
    assert np.array_equal(e, oracle.morph(oracle.ERODE, oracle.morph(oracle.DILATE, oth, k3), k3))
    oth2 = oth.copy(); oth2[20:60, 30:90] = 255
    assert np.array_equal(dilate(th, rect_kernel(3)), oracle.morph(oracle.DILATE, oth2, k3))
    # same through a numpy view handed out earlier (the view is the authoritative copy from then on)
    th = range_threshold(g, 150, 255)
    view = np.asarray(th)
    d = dilate(th, rect_kernel(3))                            # uploads the host copy at the call
    view[:] = 0
    assert np.array_equal(d, oracle.morph(oracle.DILATE, oth, k3))
    # a result nobody looks at costs no launch, and dropping it is fine
    th = range_threshold(g, 150, 255)
    ctx = vp.default_context()
    ctx.profile_begin(64)
    x = morph_remove_noise(th, rect_kernel(5))
    del x
    prof = ctx.profile_end()
    assert not prof or sum(v[1] for v in prof.values()) == 0, prof
    # arguments are checked at the call, not at first use
    with pytest.raises(vp.VpError):
        erode(th, rect_kernel(3), iterations=-1)
    # switch off: launched at the call
    devmat.set_defer(False)
    try:
        y = erode(th, rect_kernel(3))
        assert y._pending is None and np.array_equal(y, oracle.morph(oracle.ERODE, oth, k3))
    finally:
        devmat.set_defer(True)


def test_add_weighted_on_the_device(vp):
    """cv2.addWeighted stand-in (modules/bins.py:20): the device kernel makes the statement of the numpy float64 expression, on host
    arrays, device images and mixtures; ties round to even; the result saturates."""
    from vision import cv2_facade as cv2
    from vision import devmat
    from vision.utils.color import gray_to_bgr, range_threshold
    rng = np.random.default_rng(9)

    def ref(a, al, b, be, g):
        return np.clip(np.rint(np.asarray(a, np.float64) * al + np.asarray(b, np.float64) * be + g), 0, 255).astype(np.uint8)
    a = rng.integers(0, 256, (37, 53, 3)).astype(np.uint8)
    b = rng.integers(0, 256, (37, 53, 3)).astype(np.uint8)
    for al, be, g in ((0.7, 0.3, 0), (0.5, 0.5, 0), (1.5, 1.0, -20.25), (-1.0, 0.25, 300), (0.1, 0.2, 0.5)):
        out = cv2.addWeighted(a, al, b, be, g)
        assert isinstance(out, devmat.DeviceMat) and out._pending is not None
        assert np.array_equal(out, ref(a, al, b, be, g)), (al, be, g)
    # every pair of byte values once: 0.5 / 0.5 makes exact ties (half to even), 0.7 / 0.3 near-ties
    x, y = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    for al, be in ((0.5, 0.5), (0.7, 0.3), (0.3, 0.7)):
        assert np.array_equal(cv2.addWeighted(x, al, y, be, 0), ref(x, al, y, be, 0))
    # bins.py: frame (host) + mask overlay (device); odd sizes; dst given
    img = F.s2_bins(0, 333, 201)
    mask = range_threshold(np.ascontiguousarray(img[:, :, 1]), 60, 200)
    vis = cv2.cvtColor(mask, cv2.COLOR_GRAY2BGR)
    over = cv2.addWeighted(img, 0.7, vis, 0.3, 0)
    assert np.array_equal(over, ref(img, 0.7, np.dstack([np.asarray(mask)] * 3), 0.3, 0))
    dst = np.zeros_like(img)
    assert cv2.addWeighted(img, 0.25, img, 0.25, 1, dst) is dst and np.array_equal(dst, ref(img, 0.25, img, 0.25, 1))
    # shapes that differ, other types: the numpy statement as before
    f = cv2.addWeighted(a.astype(np.float32), 0.5, b.astype(np.float32), 0.5, 0)
    assert isinstance(f, np.ndarray) and np.array_equal(f, ref(a, 0.5, b, 0.5, 0))
    devmat.set_lazy(False)
    try:
        h = cv2.addWeighted(a, 0.7, b, 0.3, 0)
        assert type(h) is np.ndarray and np.array_equal(h, ref(a, 0.7, b, 0.3, 0))
    finally:
        devmat.set_lazy(True)


def test_post_takes_a_copy_without_giving_up_the_device_image(vp, oracle):
    """self.post(name, image) of a device image: one download into an array of its own (later writes to the image do not reach the
    queued post), and the image is still valid on the device for the next operator."""
    from vision.utils.color import range_threshold
    from vision.utils.transform import dilate, rect_kernel
    g = np.ascontiguousarray(F.s1_buoy(1, 200, 120)[:, :, 2])
    th = range_threshold(g, 150, 255)
    c = th.host_copy()
    ref = oracle.inrange(g, 150, 255)
    assert type(c) is np.ndarray and c.flags.writeable and np.array_equal(c, ref) and th._host is None and th._dev_ok and th.binary
    d = dilate(th, rect_kernel(3))
    th[0:5, :] = 255                                   # after the copy: the copy keeps the old contents
    assert np.array_equal(c, ref) and np.array_equal(d, oracle.morph(oracle.DILATE, ref, np.ones((3, 3), np.uint8)))
    c2 = th.host_copy()                                # host copy authoritative now
    assert c2[0, 0] == 255
    e = dilate(th, rect_kernel(3))                     # pending: host_copy computes it (item assignment hands out no alias)
    assert e._pending is not None and np.array_equal(e.host_copy(), np.asarray(e))
    assert c2 is not np.asarray(th)


def test_frames_read_straight_into_page_locked_memory(vp):
    """With a device context on the thread the binding's private read hands out arrays of the reader's own in page-locked memory (the
    seqlock copy is the only copy): writable, untouched by later reads, every plane with its type."""
    import os
    from vision.core.bindings.camera_message_framework import BlockAccessor, ReadStatus
    vp.default_context()
    d = f"pytpriv{os.getpid()}"
    a = F.s1_buoy(0, 320, 200)
    b = F.s1_buoy(1, 320, 200)
    depth = np.arange(200 * 320, dtype=np.float32).reshape(200, 320)
    with BlockAccessor(d, max_entry_size_bytes=a.nbytes + depth.nbytes) as w, BlockAccessor(d) as r:
        w.write_frame(1, [("forward", a), ("depth", depth)])
        st, data, t, private = r.read_frame_private()
        assert st == ReadStatus.SUCCESS and private and t == 1 and r.last_plane_names() == ("forward", "depth")
        fwd, dep = data
        assert fwd.shape == (200, 320, 3) and fwd.dtype == np.uint8 and fwd.flags.writeable and np.array_equal(fwd, a)
        assert dep.shape == (200, 320, 1) and dep.dtype == np.float32 and np.array_equal(dep[:, :, 0], depth)
        w.write_frame(2, [("forward", b), ("depth", depth * 2)])
        st, data2, t, private = r.read_frame_private()
        assert st == ReadStatus.SUCCESS and private and t == 2 and np.array_equal(data2[0], b)
        assert np.array_equal(fwd, a) and np.array_equal(dep[:, :, 0], depth)           # the first frame's arrays are still the first frame
        fwd[:] = 0                                                                       # and writable without consequences for the second
        assert np.array_equal(data2[0], b)
        st, again, _, _ = r.read_frame_private()
        assert st == ReadStatus.NO_NEW_FRAME
        w.write_frame(3, a)                                                              # a single plane comes back as one array
        st, one, t, private = r.read_frame_private()
        assert st == ReadStatus.SUCCESS and private and isinstance(one, np.ndarray) and np.array_equal(one, a)
        # the operators take such a frame like any other
        from vision.utils.color import bgr_to_gray
        g, _ = bgr_to_gray(one)
        assert np.asarray(g).shape == (200, 320)


def test_overlays_drawn_on_the_device(vp):
    """draw_contours / drawContours / draw_polylines into an image that lives on the device and has no host copy: drawn there (nothing
    is downloaded), with the pixels of the host rasteriser - contours, open polylines, thick brushes, points outside the image, one- and
    four-channel images; an image with a host copy is drawn on the host as before."""
    from vision import cv2_facade as cv2
    from vision import devmat
    from vision.utils import draw as D
    from vision.utils.color import gray_to_bgr, range_threshold
    rng = np.random.default_rng(12)
    img = F.s2_bins(0, 333, 201)
    polys = [np.array([[10, 10], [300, 20], [320, 190], [5, 150]], np.int32).reshape(-1, 1, 2), np.array([[-40, 100], [400, 120]], np.int32).reshape(-1, 1, 2),
             rng.integers(-20, 350, (12, 1, 2)).astype(np.int32), np.array([[[150, 100]]], np.int32)]
    for thickness in (1, 2, 4, 9, 17):
        for closed in (True, False):
            over = cv2.addWeighted(img, 0.5, img, 0.5, 0)                   # a device image (pending, then computed by the draw)
            assert isinstance(over, devmat.DeviceMat)
            for p in polys:
                D.draw_polylines(over, p, closed, (0, 255, 0), thickness)
            assert over._host is None, "the overlay was downloaded for drawing"
            ref = img.copy()
            for p in polys:
                D.draw_polylines(ref, p, closed, (0, 255, 0), thickness)
            assert np.array_equal(over, ref), (thickness, closed)
    # bins.py: cv2.drawContours(overlayed, [box_points], 0, (0, 255, 0), 4) for every rectangle
    over = cv2.addWeighted(img, 0.7, img, 0.3, 0)
    ref = np.asarray(cv2.addWeighted(img, 0.7, img, 0.3, 0)).copy()
    boxes = [np.intp(cv2.boxPoints(((100.0 + 30 * i, 90.0), (60.0, 30.0), 20.0 * i + 5))) for i in range(5)]
    for b in boxes:
        assert cv2.drawContours(over, [b], 0, (0, 255, 0), 4) is over
        cv2.drawContours(ref, [b], 0, (0, 255, 0), 4)
    assert over._host is None and np.array_equal(over, ref)
    # one channel (a mask) and the contour tuple of find_contours
    g = np.ascontiguousarray(img[:, :, 1])
    m = range_threshold(g, 60, 200)
    from vision.utils.feature import outer_contours
    cs = outer_contours(m)
    m2 = range_threshold(g, 60, 200)
    D.draw_contours(m2, cs, 128, 3)
    ref = np.asarray(range_threshold(g, 60, 200)).copy()
    D.draw_contours(ref, cs, 128, 3)
    assert m2._host is None and not m2.binary and np.array_equal(m2, ref)
    # an image whose host copy exists is drawn on the host
    vis = gray_to_bgr(g)[0]
    h = np.asarray(vis)
    D.draw_contours(vis, cs, (1, 2, 3), 2)
    assert vis._host is not None and (h == np.asarray(vis)).all() and (np.asarray(vis)[..., 0] == 1).any()


def test_writes_through_an_alias_after_a_re_upload_are_seen(vp, oracle):
    """A writable host alias handed out once stays writable for ever: an operator that re-uploads the host copy must not make the
    device copy trusted again, or a later write through the alias would be lost (roi = m[a:b]; op(m); roi[:] = 0; op(m))."""
    from vision.utils.color import range_threshold
    from vision.utils.transform import dilate, erode, rect_kernel
    g = np.ascontiguousarray(F.s1_buoy(2, 320, 200)[:, :, 2])
    k3 = np.ones((3, 3), np.uint8)
    ref = oracle.inrange(g, 150, 255)
    th = range_threshold(g, 150, 255)
    roi = th[40:90]                                          # a view of the host copy, kept by the caller
    first = dilate(th, rect_kernel(3))
    assert first._pending is None, "an operator on an image with an alias out must run at its call"
    assert np.array_equal(first, oracle.morph(oracle.DILATE, ref, k3))
    roi[:] = 255                                             # no hook sees this write
    ref2 = ref.copy(); ref2[40:90] = 255
    assert np.array_equal(dilate(th, rect_kernel(3)), oracle.morph(oracle.DILATE, ref2, k3))
    roi[:, 100:200] = 0
    ref2[40:90, 100:200] = 0
    assert np.array_equal(erode(th, rect_kernel(3)), oracle.morph(oracle.ERODE, ref2, k3))
    assert np.array_equal(th, ref2)
    # a reshaped alias shares the state
    th3 = range_threshold(g, 150, 255)
    al = th3.reshaped((1,) + th3.shape).reshaped(th3.shape)
    v = np.asarray(th3)
    al2 = th3.reshaped(th3.shape)
    dilate(al2, rect_kernel(3))
    v[0:10] = 255
    ref3 = ref.copy(); ref3[0:10] = 255
    assert np.array_equal(dilate(al2, rect_kernel(3)), oracle.morph(oracle.DILATE, ref3, k3))
    del al


def test_device_draw_forces_operators_deferred_on_the_image(vp, oracle):
    """An in-place device write (draw_contours into an image that lives on the device) is a write like any other: a morphology that
    was deferred on that image - also one registered through a reshaped alias - reads the image as it was at its call."""
    from vision.devmat import DeviceMat
    from vision.utils.color import range_threshold
    from vision.utils.draw import draw_contours
    from vision.utils.transform import dilate, rect_kernel
    g = np.ascontiguousarray(F.s1_buoy(5, 320, 200)[:, :, 2])
    k3 = np.ones((3, 3), np.uint8)
    ref = oracle.inrange(g, 150, 255)
    th = range_threshold(g, 150, 255)
    assert isinstance(th, DeviceMat) and th._host is None
    alias = th.reshaped(th.shape)
    d1 = dilate(th, rect_kernel(3))
    d2 = dilate(alias, rect_kernel(3))
    assert d1._pending is not None and d2._pending is not None
    square = np.array([[[20, 20]], [[120, 20]], [[120, 90]], [[20, 90]]], np.int32)
    draw_contours(th, [square], color=(255, 255, 255), thickness=3)
    assert th._host is None, "the overlay was meant to be drawn by the device"
    assert d1._pending is None and d2._pending is None, "pending readers must run before the device write"
    exp = oracle.morph(oracle.DILATE, ref, k3)
    assert np.array_equal(d1, exp) and np.array_equal(d2, exp)
    assert not np.array_equal(np.asarray(th), ref)           # and the drawing did land
