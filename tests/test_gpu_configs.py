"""GPU suite: BASELINE.json configs 3 and 4 as parity cases (they are not bench lines).
config 3: bins (HSV -> inRange C3 -> OPEN 5x5 -> CCL) and a gate-style echo on two camera directions concurrently, one HIP
          stream (= one libvp context) per direction, 1080p.
config 4: 4K frames in a 32-deep batch, contiguous shard per rank (frames [4g, 4g+4) -> GPU g): here one rank's shard."""
import sys
import threading

import numpy as np
import pytest

import frames as F

pytestmark = pytest.mark.gpu


def test_config3_two_directions_two_streams(vp, oracle):
    from vision.utils.chain import run_chain
    W, H = 1920, 1080
    bins_frames = np.stack([F.s2_bins(i, W, H) for i in range(3)])
    gate_frames = np.stack([F.s1_buoy(i, W, H) for i in range(3)])
    results, streams, errors = {}, {}, []

    def forward():   # bins chain
        try:
            streams["forward"] = vp.lib().vp_get_stream(vp.default_context().handle)
            for _ in range(3):
                results["forward"] = run_chain(bins_frames, vp.BGR2HSV, (10, 20, 60), (30, 100, 255), [(vp.MORPH_OPEN, 5, 5)], ccl=1, max_labels=1024)
        except Exception as e:   # pragma: no cover
            errors.append(e)

    def downward():  # gate echoes frames; give its stream real work with the buoy chain
        try:
            streams["downward"] = vp.lib().vp_get_stream(vp.default_context().handle)
            for _ in range(3):
                results["downward"] = run_chain(gate_frames, vp.BGR2LAB, (0, 150, 0), (255, 255, 255), [(vp.MORPH_OPEN, 5, 5), (vp.MORPH_CLOSE, 5, 5)],
                                                ccl=1, max_labels=1024)
        except Exception as e:   # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=forward), threading.Thread(target=downward)]
    [t.start() for t in ts]
    [t.join(120) for t in ts]
    assert not errors, errors
    assert streams["forward"] != streams["downward"]          # one context / HIP stream per direction
    k = np.ones((5, 5), np.uint8)
    for i in range(3):
        th = oracle.inrange(oracle.bgr2hsv(bins_frames[i]), (10, 20, 60), (30, 100, 255))
        cl = oracle.morph(oracle.OPEN, th, k, fast=True)
        n, lab, st, ce = oracle.ccl(cl, 2)
        r = results["forward"]
        assert np.array_equal(r["threshed"][i], th) and np.array_equal(r["cleaned"][i], cl)
        assert r["nlabels"][i] == n and np.array_equal(r["labels"][i], lab) and np.array_equal(r["stats"][i][:n], st)
        ref = oracle.chain(gate_frames[i], oracle.MODE_LAB, (0, 150, 0), (255, 255, 255), [oracle.OPEN, oracle.CLOSE], 5, 5, 2, 1024)
        r = results["downward"]
        assert np.array_equal(r["cleaned"][i], ref["cleaned"]) and np.array_equal(r["labels"][i], ref["labels"])


def test_config3_bins_and_gate_modules_on_two_directions(vp, oracle, capsys):
    """BASELINE config 3 with the modules themselves: the bins harness (modules/bins.py:11-81) on one camera direction and the gate
    harness (modules/gate.py:8-21: post + normalize + latency) on another, both on the runtime at the same time, 1080p - every module's
    loop thread has its own libvp context, i.e. its own HIP stream; bins' masks and contours equal the oracle's, the gate's post
    (a device image, published by DMA) comes back bit-equal through a reference-layout reader."""
    import glob
    import os
    import time
    import module_harness as MH
    from vision import cv2_facade
    from vision.core.bindings.camera_message_framework import BlockAccessor
    had_cv2 = sys.modules.get("cv2")
    cv2_facade.install()
    MH.module_argv()
    W, H = 1920, 1080
    pid = os.getpid()
    d_bins, d_gate = f"pytc3bins{pid}", f"pytc3gate{pid}"
    bins_frame, gate_frame = F.s2_bins(1, W, H), F.s1_buoy(1, W, H)
    seen = {"bins": [], "gate": []}
    streams = {}

    def on_bins(mod, direction, img, out):
        streams["bins"] = vp.lib().vp_get_stream(vp.default_context().handle)
        cleaned, contours, valid, overlayed = out
        seen["bins"].append((np.array(cleaned, copy=True), [np.array(c, copy=True) for c in contours], len(valid)))

    def on_gate(mod, direction, image, norm, lat):
        streams["gate"] = vp.lib().vp_get_stream(vp.default_context().handle)
        seen["gate"].append(norm)

    try:
        with BlockAccessor(d_bins, max_entry_size_bytes=bins_frame.nbytes) as wb, BlockAccessor(d_gate, max_entry_size_bytes=gate_frame.nbytes) as wg:
            bins_mod = MH.bins_module(on_bins)([d_bins], [])
            gate_mod = MH.gate_module(on_gate)([d_gate], MH.gate_tuners())
            bins_mod._fps = gate_mod._fps = 500
            runners = [threading.Thread(target=bins_mod), threading.Thread(target=gate_mod)]
            [t.start() for t in runners]
            post = None
            try:
                t0 = time.time()
                while (len(seen["bins"]) < 3 or len(seen["gate"]) < 3) and time.time() - t0 < 60:
                    now = int(time.monotonic() * 1000)
                    wb.write_frame(now, bins_frame)
                    wg.write_frame(now, gate_frame)
                    time.sleep(0.01)
                hits = glob.glob(f"/dev/shm/auv_visiond_module_{gate_mod._name}_post%*%post_{d_gate}#BGR")
                assert hits, "the gate module published no post block"
                with BlockAccessor(hits[0][len("/dev/shm/auv_visiond_"):]) as rd:
                    t0 = time.time()
                    while post is None and time.time() - t0 < 5:
                        st, data, _ = rd.read_frame()
                        if data is not None:
                            post = np.array(data, copy=True)
                        time.sleep(0.005)
            finally:
                bins_mod.stop()
                gate_mod.stop()
                [t.join(15) for t in runners]
    finally:
        if had_cv2 is None:
            sys.modules.pop("cv2", None)
    capsys.readouterr()
    assert len(seen["bins"]) >= 3 and len(seen["gate"]) >= 3
    assert streams["bins"] != streams["gate"], "the two modules share a HIP stream"
    th = oracle.inrange(oracle.bgr2hsv(bins_frame), (10, 20, 60), (30, 100, 255))
    cl = oracle.morph(oracle.OPEN, th, np.ones((5, 5), np.uint8), fast=True)
    exp = oracle.find_contours(cl, oracle.RETR_EXTERNAL, oracle.CHAIN_APPROX_SIMPLE)
    cleaned, contours, nvalid = seen["bins"][-1]
    assert np.array_equal(cleaned, cl) and len(contours) == len(exp) and all(np.array_equal(a, b) for a, b in zip(contours, exp))
    assert seen["gate"][-1] == ((600 - H / 2) / W, (800 - W / 2) / W)
    assert post is not None and np.array_equal(post, gate_frame)
    assert gate_mod._posts.dma_posts >= 3 and gate_mod._posts.host_posts == 0


def test_config4_4k_batch_shard(vp, oracle):
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    import bench
    from vision.utils.chain import run_chain
    W, H = 3840, 2160
    lo, hi = bench.shard_of(32, 3, 8)                          # rank 3 of 8 owns frames [12, 16) of every 32-deep batch
    assert (lo, hi) == (12, 16)
    shard = np.stack([F.s1_buoy(100 + i, W, H) for i in range(lo, hi)])
    morph = [(vp.MORPH_OPEN, 5, 5), (vp.MORPH_CLOSE, 5, 5)]
    out = run_chain(shard, vp.BGR2LAB, (0, 150, 0), (255, 255, 255), morph, ccl=1, max_labels=2048)
    ref = oracle.chain(shard[0], oracle.MODE_LAB, (0, 150, 0), (255, 255, 255), [oracle.OPEN, oracle.CLOSE], 5, 5, 2, 2048)
    assert np.array_equal(out["threshed"][0], ref["threshed"]) and np.array_equal(out["cleaned"][0], ref["cleaned"])
    assert out["nlabels"][0] == ref["nlabels"] and np.array_equal(out["labels"][0], ref["labels"])
    assert np.array_equal(out["stats"][0][:ref["nlabels"]], ref["stats"])
    for i in range(hi - lo):                                   # size-independent properties on the rest of the shard
        nl = int(out["nlabels"][i])
        lab = out["labels"][i]
        assert np.array_equal(lab > 0, out["cleaned"][i] > 0) and lab.max() == nl - 1
        assert np.array_equal(np.bincount(lab.ravel(), minlength=nl), out["stats"][i][:nl, 4])


def test_one_call_over_more_than_2_pow_32_pixels(vp):
    """Maximum sizes: 2,100 frames of 1080p in one vp_chain_run (4.35e9 pixels, 12 GiB in, 37 GiB out, ~40 GiB of workspace) — every
    frame's masks, labels and statistics must equal those of the same frame in a batch of 8 (64-bit indexing throughout).  Needs
    ~100 GiB of free device memory; skipped on anything smaller than the 288 GB part."""
    import torch
    free, _total = torch.cuda.mem_get_info()
    if free < 120 * 2**30:
        pytest.skip("needs ~100 GiB of free device memory")
    W, H, B, D = 1920, 1080, 2100, 8
    ctx = vp.Context(0)
    try:
        desc = vp.make_chain_desc(W, H, vp.BGR2LAB, (0, 150, 0), (255, 255, 255), [(vp.MORPH_OPEN, 5, 5), (vp.MORPH_CLOSE, 5, 5)], ccl=1, max_labels=64)
        small = torch.from_numpy(np.stack([F.s1_buoy(i, W, H) for i in range(D)])).cuda()

        def run(d):
            n = d.shape[0]
            t = {"thr": torch.empty((n, H, W), dtype=torch.uint8, device="cuda"), "cln": torch.empty((n, H, W), dtype=torch.uint8, device="cuda"),
                 "lab": torch.empty((n, H, W), dtype=torch.int32, device="cuda"), "st": torch.zeros((n, 64, 5), dtype=torch.int32, device="cuda"),
                 "ce": torch.zeros((n, 64, 2), dtype=torch.float64, device="cuda"), "nl": torch.zeros((n,), dtype=torch.int32, device="cuda")}
            b = vp.ChainBuffers()
            b.bgr = d.data_ptr()
            b.threshed, b.cleaned, b.labels, b.stats, b.centroids, b.nlabels = (t[k].data_ptr() for k in ("thr", "cln", "lab", "st", "ce", "nl"))
            torch.cuda.synchronize()
            ctx.chain_run(desc, b, n)
            ctx.synchronize()
            return t

        ref = run(small)
        big = small.repeat((B + D - 1) // D, 1, 1, 1)[:B].contiguous()
        out = run(big)
        reps = (B + D - 1) // D
        for k in ref:
            exp = ref[k].repeat((reps,) + (1,) * (ref[k].dim() - 1))[:B]
            assert torch.equal(out[k], exp), k
            del exp
    finally:
        ctx.close()
        torch.cuda.empty_cache()
