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
