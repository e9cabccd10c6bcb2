"""BASELINE config 1: red_buoy-style module fed by the image_directory capture source through the CMF ring.

CPU variant (no GPU here): the module's process() uses the CPU oracle as the checker-side stand-in for the operator
calls — this test is about the plumbing (PNG files -> capture thread -> shared-memory ring with the 2-plane `zed`
contract -> ModuleBase loop -> @sources handler -> posts), and asserts that what the module computed from the frames
it received equals the direct-call results on the source frames.
GPU variant (-m gpu): the same module written against vision.utils.* exactly like modules/red_buoy.py:21-38, i.e. the
shipped HIP path behind the runtime, compared with the oracle."""
import os
import sys
import threading
import time

import numpy as np
import pytest

import frames as F
from vision.core.bindings.camera_message_framework import BlockAccessor
from vision.core.tuners import IntTuner

PID = os.getpid()
W, H = 1280, 720


def _make_frames(n):
    out = []
    for i in range(n):
        f = F.s1_buoy(i, W, H)
        f[0, 0] = (i, 110, 40)              # frame id in a background pixel (stays background: a-channel ~120)
        out.append(f)
    return out


def _feeder(direction, frames, stop, fps=100):
    """image_directory-style producer that honours the reference's plane contract for `zed`: BuoyLAB listens on
    zed[forward], zed[normal] (modules/red_buoy.py:18), so every frame carries a BGR u8 plane and an f32 normal plane."""
    from vision.core.capture_source import CaptureSource

    class Zed(CaptureSource):
        pass
    src = Zed()
    normal = np.zeros((H, W, 3), np.float32)

    def udl(limiter, args):
        k = 0
        for t in limiter.rate(fps):
            yield direction, t, (frames[k % len(frames)], normal), ("forward", "normal")
            k += 1
    src.register_capture_udl("zed", udl)
    src._quit_flag = stop
    for t in src._workers:
        t.start()
    return src


def _run(module_factory, n_frames, want):
    direction = f"pytzed{PID}x{int(time.time() * 1000) % 100000}"
    frames = _make_frames(n_frames)
    stop = threading.Event()
    src = _feeder(direction, frames, stop)
    mod = module_factory(direction)
    runner = threading.Thread(target=mod)
    runner.start()
    try:
        t0 = time.time()
        while len(mod.results) < want and time.time() - t0 < 60:
            time.sleep(0.02)
    finally:
        mod.stop()
        runner.join(10)
        stop.set()
        for t in src._workers:
            t.join(5)
        src.close()
    return frames, mod


def test_config1_plumbing_cpu(monkeypatch, oracle):
    monkeypatch.setattr(sys, "argv", ["red_buoy.py"])
    from vision.core.base import ModuleBase, sources

    def factory(direction):
        class BuoyLAB(ModuleBase):
            def __init__(self):
                super().__init__([direction], [IntTuner("thresh_min", 150, 0, 255), IntTuner("thresh_max", 255, 0, 255)], fps=200)
                self.results = {}

            @sources("zed[forward]", "zed[normal]")
            def process_img(self, image, normal):
                assert normal.dtype == np.float32 and normal.shape == (H, W, 3)
                lab = oracle.bgr2lab(image)
                threshed = oracle.inrange(np.ascontiguousarray(lab[:, :, 1]), self.tuners["thresh_min"], self.tuners["thresh_max"])
                self.post("threshed", threshed, "GRAY")
                k = np.ones((5, 5), np.uint8)
                cleaned = oracle.morph(oracle.CLOSE, oracle.morph(oracle.OPEN, threshed, k, fast=True), k, fast=True)
                n, _, stats, cent = oracle.ccl(cleaned, 2, max_k=256, want_labels=False)
                self.results[int(image[0, 0, 0])] = (n, stats.copy(), cent.copy(), self.normalize((H / 2, W / 2)))
        return BuoyLAB()

    frames, mod = _run(factory, 6, 6)
    assert len(mod.results) >= 4                      # latest-wins ring: a slow consumer may skip frames, never reorder
    for idx, (n, stats, cent, centre) in mod.results.items():
        ref = oracle.chain(frames[idx], oracle.MODE_LAB, (0, 150, 0), (255, 255, 255), [oracle.OPEN, oracle.CLOSE], 5, 5, 2, 256,
                           want_labels=False)
        assert n == ref["nlabels"] and n >= 2
        assert np.array_equal(stats, ref["stats"]) and np.array_equal(cent, ref["centroids"])
        assert centre == (0.0, 0.0)


@pytest.mark.gpu
def test_config1_red_buoy_on_gpu(monkeypatch, vp, oracle):
    monkeypatch.setattr(sys, "argv", ["red_buoy.py"])
    from vision.core.base import ModuleBase, sources
    from vision.utils.color import bgr_to_lab, range_threshold
    from vision.utils.feature import connected_components
    from vision.utils.transform import morph_close_holes, morph_remove_noise, rect_kernel

    def factory(direction):
        class BuoyLAB(ModuleBase):
            def __init__(self):
                super().__init__([direction], [IntTuner("thresh_min", 150, 0, 255), IntTuner("thresh_max", 255, 0, 255)], fps=200)
                self.results = {}

            @sources("zed[forward]", "zed[normal]")
            def process_img(self, image, normal):          # body follows modules/red_buoy.py:21-38
                lab, (lab_l, lab_a, lab_b) = bgr_to_lab(image)
                threshed = range_threshold(lab_a, self.tuners["thresh_min"], self.tuners["thresh_max"])
                self.post("threshed", threshed, "GRAY")
                kernel = rect_kernel(5)
                cleaned = morph_remove_noise(threshed, kernel)
                cleaned = morph_close_holes(cleaned, kernel)
                self.post("threshed_cleaned", cleaned, "GRAY")
                n, labels, stats, cent = connected_components(cleaned, max_labels=256)
                self.results[int(image[0, 0, 0])] = (threshed, cleaned, n, labels, stats, cent)
        return BuoyLAB()

    frames, mod = _run(factory, 4, 4)
    assert len(mod.results) >= 2
    for idx, (threshed, cleaned, n, labels, stats, cent) in mod.results.items():
        ref = oracle.chain(frames[idx], oracle.MODE_LAB, (0, 150, 0), (255, 255, 255), [oracle.OPEN, oracle.CLOSE], 5, 5, 2, 256)
        assert np.array_equal(threshed, ref["threshed"]) and np.array_equal(cleaned, ref["cleaned"])
        assert n == ref["nlabels"] and np.array_equal(labels, ref["labels"])
        assert np.array_equal(stats, ref["stats"]) and np.array_equal(cent, ref["centroids"])
