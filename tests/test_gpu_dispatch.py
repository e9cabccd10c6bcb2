"""vision.dispatch on a real device: one rank's share of config 4's 32-deep batches (rank 3 of 8 owns frames [12, 16)), and a whole
batch through one device with two feeder threads; results against the oracle.  The N-device figure itself needs an N-GPU node."""
import os

import numpy as np
import pytest

import frames as F

pytestmark = pytest.mark.gpu


def _check(oracle, frames, lo, hi, res, max_labels):
    for k in range(hi - lo):
        ref = oracle.chain(frames[lo + k], oracle.MODE_LAB, (0, 150, 0), (255, 255, 255), [oracle.OPEN, oracle.CLOSE], 5, 5, 2, max_labels)
        n = ref["nlabels"]
        assert int(res["nlabels"][k]) == n
        assert np.array_equal(res["stats"][k][:n], ref["stats"])
        if "labels" in res:
            assert np.array_equal(res["labels"][k], ref["labels"]) and np.array_equal(res["cleaned"][k], ref["cleaned"])


def test_one_ranks_share_and_whole_batches(vp, oracle):
    from vision.dispatch import BatchDispatcher
    W, H, B = 480, 270, 32
    chain = dict(color_mode=vp.BGR2LAB, lo=(0, 150, 0), hi=(255, 255, 255), morph=[(vp.MORPH_OPEN, 5, 5), (vp.MORPH_CLOSE, 5, 5)], ccl=1,
                 max_labels=128, want=("stats", "labels", "cleaned"))
    batches = [np.stack([F.s1_buoy(50 * b + i, W, H) for i in range(B)]) for b in range(3)]
    with BatchDispatcher([0], B, H, W, chain=chain, rank=3, world=8, ring=2) as d:        # config 4: this process is GPU 3 of 8
        assert d.slices == [(12, 16)]
        for b in batches:
            d.submit(b)
        for b in batches:
            bid, ((lo, hi, res),) = d.collect()
            assert (lo, hi) == (12, 16)
            _check(oracle, b, lo, hi, res, 128)
        bound = d.bound_cpus[0]
        assert bound <= os.sched_getaffinity(0)            # bound to the GPU's NUMA node when the platform names one, else left alone
    chain["want"] = ("stats",)
    with BatchDispatcher([0], B, H, W, chain=chain, ring=2) as d:                           # the whole batch on one device
        ids = [d.submit(b) for b in batches]
        for want_id, b in zip(ids, batches):
            bid, ((lo, hi, res),) = d.collect()
            assert bid == want_id and (lo, hi) == (0, B)
            _check(oracle, b, lo, hi, res, 128)


def test_device_is_found_on_the_pci_bus(vp):
    import ctypes as C
    buf = C.create_string_buffer(32)
    assert vp.lib().vp_device_count() >= 1
    vp.check(vp.lib().vp_device_pci_bus_id(0, buf, 32))
    addr = buf.value.decode()
    assert len(addr.split(":")) == 3 and os.path.isdir(os.path.join("/sys/bus/pci/devices", addr)), addr
