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


def test_video_capture_source_feeds_the_dispatcher(oracle, tmp_path):
    """BASELINE config 4's wiring end to end on the CPU: a frame stack played by vision.capture_sources.video (one frame per tick into a
    shared-memory block, capture_sources/video.py:9-29) is read back, gathered into batches and sharded over three stand-in devices by
    BatchDispatcher; every frame's result equals the oracle's and comes back in capture order."""
    import time
    from vision.capture_sources.video import Video
    from vision.core.bindings.camera_message_framework import BlockAccessor, ReadStatus
    from vision.dispatch import BatchDispatcher
    n, h, w, batch = 12, 36, 64, 6
    stack = np.stack([F.s1_buoy(i, w, h) for i in range(n)])
    path = tmp_path / "clip.npy"
    np.save(path, stack)
    d = f"pytv4{os.getpid()}"
    src = Video(str(path), [d], fps=100, loop=True)          # (the clip repeats: the reader attaches whenever it does)
    t = threading.Thread(target=src.run_event_loop)
    got = []
    with BatchDispatcher([0, 1, 2], batch, h, w, make_runner=OracleRunner, bind_numa=False, ring=2) as disp:
        t.start()
        try:
            with BlockAccessor(d) as r:
                pending, submitted, t0 = [], 0, time.time()
                while len(got) < n // batch and time.time() - t0 < 30:
                    st, frame, stamp = r.read_frame()
                    if st == ReadStatus.SUCCESS and frame is not None:
                        pending.append(np.array(frame, copy=True))
                    elif st == ReadStatus.FRAMEWORK_DELETED:
                        break
                    if len(pending) == batch:
                        disp.submit(np.stack(pending)); submitted += 1
                        got.append((np.stack(pending), disp.collect()))
                        pending = []
                    time.sleep(0.001)
        finally:
            src._quit_flag.set()
            t.join(5)
            src.close()
    assert got, "no batch made it through"
    for frames, (bid, parts) in got:
        assert [p[:2] for p in parts] == [(0, 2), (2, 4), (4, 6)]
        for lo, hi, res in parts:
            for k in range(lo, hi):
                ref = oracle.chain(frames[k], oracle.MODE_LAB, (0, 150, 0), (255, 255, 255), [oracle.OPEN, oracle.CLOSE], 5, 5, 2, 64, want_labels=False)
                assert int(res["nlabels"][k - lo]) == ref["nlabels"]
                assert np.array_equal(res["stats"][k - lo][:ref["nlabels"]], ref["stats"])
    # the latest-wins ring may drop frames when the reader is slow; what was read is in capture order and from the clip
    flat = [fr for frames, _ in got for fr in frames]
    seen = [int(np.argmax([np.array_equal(fr, s) for s in stack])) for fr in flat]
    assert all(np.array_equal(stack[i], fr) for i, fr in zip(seen, flat))
    steps = [(b - a) % n for a, b in zip(seen, seen[1:])]
    assert all(1 <= st < n for st in steps) and sum(steps) < 3 * n, seen        # forwards through the (repeating) clip, never backwards


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
