"""vision.dispatch: host-fed multi-device dispatch of frame batches (BASELINE config 4), driven here with a stand-in for the device so
that the sharding, the feeder threads, the ordering of results, NUMA placement and the one-process-per-device mode (world_size 2 over
gloo) are exercised on a CPU box.  tests/test_gpu_dispatch.py runs the same code on a real device."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import frames as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleRunner:
    """Stand-in for vision.utils.chain.ChainRunner: same surface (.input, .run()), the oracle's chain as the "device"."""
    seen = []                                            # (device, thread name, cpu affinity) of every instance, for the placement test

    def __init__(self, device, n_frames, height, width):
        from oracle import oracle as orc
        self.orc = orc
        self.input = np.zeros((n_frames, height, width, 3), np.uint8)
        self.n = n_frames
        aff = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set()
        OracleRunner.seen.append((device, threading.current_thread().name, frozenset(aff)))

    def run(self):
        orc = self.orc
        outs = [orc.chain(self.input[i], orc.MODE_LAB, (0, 150, 0), (255, 255, 255), [orc.OPEN, orc.CLOSE], 5, 5, 2, 64, want_labels=False)
                for i in range(self.n)]
        stats = np.zeros((self.n, 64, 5), np.int32)
        for i, o in enumerate(outs):
            stats[i, :o["nlabels"]] = o["stats"]
        return {"nlabels": np.array([o["nlabels"] for o in outs], np.int32), "stats": stats}


def test_shards_cover_the_batch():
    from vision.dispatch import shard_of
    sys.path.insert(0, ROOT)
    import bench
    for n in (0, 1, 7, 32, 33):
        for shards in (1, 2, 3, 8):
            assert [shard_of(n, g, shards) for g in range(shards)] == [bench.shard_of(n, g, shards) for g in range(shards)]
    assert shard_of(32, 3, 8) == (12, 16)                 # config 4: frames [4g, 4g + 4) -> GPU g


def test_numa_lookup_from_sysfs(tmp_path):
    from vision import dispatch as D
    (tmp_path / "bus/pci/devices/0000:05:00.0").mkdir(parents=True)
    (tmp_path / "bus/pci/devices/0000:05:00.0/numa_node").write_text("1\n")
    (tmp_path / "bus/pci/devices/0000:85:00.0").mkdir(parents=True)
    (tmp_path / "bus/pci/devices/0000:85:00.0/numa_node").write_text("-1\n")
    (tmp_path / "devices/system/node/node1").mkdir(parents=True)
    (tmp_path / "devices/system/node/node1/cpulist").write_text("0-1,4,6-7\n")
    assert D.numa_node_of_pci("0000:05:00.0", str(tmp_path)) == 1
    assert D.numa_node_of_pci("0000:85:00.0", str(tmp_path)) is None and D.numa_node_of_pci("0000:ff:00.0", str(tmp_path)) is None
    assert D.cpus_of_numa_node(1, str(tmp_path)) == {0, 1, 4, 6, 7}
    allowed = os.sched_getaffinity(0)
    assert D.cpus_near_device(0, str(tmp_path), lambda d: "0000:05:00.0") == {0, 1, 4, 6, 7} & allowed
    assert D.cpus_near_device(1, str(tmp_path), lambda d: "0000:85:00.0") == set()


def test_batches_through_three_stand_in_devices(oracle, tmp_path):
    """7 frames per batch over 3 devices (3 + 2 + 2), two feeder threads each, feeders of device 0 bound to the CPUs a fake sysfs
    reports: every frame's result equals the oracle's, batches come back in submission order."""
    from vision.dispatch import BatchDispatcher
    allowed = sorted(os.sched_getaffinity(0))
    near = set(allowed[:2])
    (tmp_path / "bus/pci/devices/0000:05:00.0").mkdir(parents=True)
    (tmp_path / "bus/pci/devices/0000:05:00.0/numa_node").write_text("0\n")
    (tmp_path / "devices/system/node/node0").mkdir(parents=True)
    (tmp_path / "devices/system/node/node0/cpulist").write_text(",".join(str(c) for c in sorted(near)) + "\n")
    OracleRunner.seen.clear()
    H, W, B = 36, 64, 7
    batches = [np.stack([F.s1_buoy(10 * b + i, W, H) for i in range(B)]) for b in range(4)]
    with BatchDispatcher([0, 1, 2], B, H, W, ring=2, make_runner=OracleRunner, sysfs=str(tmp_path),
                         pci_lookup=lambda d: "0000:05:00.0" if d == 0 else None) as d:
        assert d.slices == [(0, 3), (3, 5), (5, 7)]
        ids = [d.submit(b) for b in batches]
        for want_id, frames in zip(ids, batches):
            bid, parts = d.collect()
            assert bid == want_id and [(lo, hi) for lo, hi, _ in parts] == [(0, 3), (3, 5), (5, 7)]
            for lo, hi, res in parts:
                for k in range(hi - lo):
                    ref = oracle.chain(frames[lo + k], oracle.MODE_LAB, (0, 150, 0), (255, 255, 255), [oracle.OPEN, oracle.CLOSE], 5, 5, 2, 64, want_labels=False)
                    assert int(res["nlabels"][k]) == ref["nlabels"] and np.array_equal(res["stats"][k][:ref["nlabels"]], ref["stats"])
        assert d.bound_cpus[0] == near and d.bound_cpus[1] == set() and d.bound_cpus[2] == set()
    assert len(OracleRunner.seen) == 6                                            # one runner (context + staging) per feeder thread
    for dev, name, aff in OracleRunner.seen:
        assert name.startswith(f"vp-feed-d{dev}-")
        assert aff == (frozenset(near) if dev == 0 else frozenset(allowed))        # bound before the runner (its pinned buffers) is made
    assert os.sched_getaffinity(0) == set(allowed)                                 # the submitting thread keeps its own affinity


def test_frame_stack_in_batches(tmp_path):
    from vision.dispatch import batches_of
    stack = np.arange(10 * 4 * 6 * 3, dtype=np.uint8).reshape(10, 4, 6, 3)
    path = str(tmp_path / "clip.npy")
    np.save(path, stack)
    got = list(batches_of(path, 4))
    assert len(got) == 2 and np.array_equal(got[0], stack[:4]) and np.array_equal(got[1], stack[4:8])


def test_runner_failure_reaches_the_caller():
    from vision.dispatch import BatchDispatcher

    class Broken:
        def __init__(self, *a):
            raise RuntimeError("no device")
    with pytest.raises(RuntimeError, match="no device"):
        BatchDispatcher([0], 4, 8, 8, make_runner=Broken, bind_numa=False)


def test_a_failed_batch_does_not_wedge_the_dispatcher():
    """A feeder that raises on one batch: collect() raises for that batch once every device has answered, the batch leaves the
    books, the next collect() serves the next batch; close() returns although batches were never collected."""
    from vision.dispatch import BatchDispatcher

    class Flaky:
        def __init__(self, device, n, h, w):
            self.input = np.zeros((n, h, w, 3), np.uint8)

        def run(self):
            if int(self.input[0, 0, 0, 0]) == 13:
                raise ValueError("bad batch")
            return {"first": self.input[:, 0, 0, 0].copy()}
    good = np.zeros((4, 8, 8, 3), np.uint8)
    bad = np.full((4, 8, 8, 3), 13, np.uint8)
    with BatchDispatcher([0, 1], 4, 8, 8, make_runner=Flaky, bind_numa=False, ring=1) as d:
        d.submit(good); d.submit(bad); d.submit(good + 1)
        bid, parts = d.collect()
        assert bid == 0 and [p[:2] for p in parts] == [(0, 2), (2, 4)]
        with pytest.raises(ValueError, match="bad batch"):
            d.collect()
        bid, parts = d.collect()                         # the batch after the failed one
        assert bid == 2 and all((p[2]["first"] == 1).all() for p in parts)
        with pytest.raises(RuntimeError, match="nothing submitted"):
            d.collect()
        d.submit(good)                                   # left uncollected: close() must not hang on it
    assert not d._threads


@pytest.mark.timeout(240)
def test_two_ranks_gloo(oracle):
    """One process per device, as bench.py is launched: each rank dispatches its share of every batch; together they cover every
    frame exactly once and every result equals the oracle's."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(31500 + os.getpid() % 2000), os.path.join(ROOT, "tests", "_dispatch_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=230)
    assert out.returncode == 0, out.stderr[-3000:]
    ranks = json.loads(next(l for l in out.stdout.splitlines() if l.startswith("RESULT "))[len("RESULT "):])
    merged = {}
    for r in ranks:
        assert not (set(r["results"]) & set(merged)), "a frame was processed by two ranks"
        merged.update(r["results"])
    assert sorted(merged) == sorted(f"{b}:{i}" for b in range(3) for i in range(7))
    for key, (nl, stats) in merged.items():
        b, i = (int(x) for x in key.split(":"))
        ref = oracle.chain(F.s1_buoy(100 * b + i, 64, 36), oracle.MODE_LAB, (0, 150, 0), (255, 255, 255), [oracle.OPEN, oracle.CLOSE], 5, 5, 2, 64, want_labels=False)
        assert nl == ref["nlabels"] and np.array_equal(np.array(stats)[:nl], ref["stats"])
