"""CPU suite: the N > 1 path of bench.py with world_size 2 over gloo — frames are sharded contiguously with no
data-path collective, every rank reports the same MAX-over-ranks time, shards are disjoint and complete."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_of():
    sys.path.insert(0, ROOT)
    import bench
    for n in (0, 1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            cuts = [bench.shard_of(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    assert bench.shard_of(32, 3, 8) == (12, 16)      # config 4: frames [4g, 4g+4) -> GPU g


@pytest.mark.timeout(180)
def test_two_ranks_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + os.getpid() % 2000), os.path.join(ROOT, "tests", "_rank_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stderr[-2000:]
    line = next(l for l in out.stdout.splitlines() if l.startswith("RESULT "))
    ranks = json.loads(line[len("RESULT "):])
    assert [r["rank"] for r in ranks] == [0, 1]
    assert ranks[0]["range"] == [0, 4] and ranks[1]["range"] == [4, 7]
    covered = sorted(int(k) for r in ranks for k in r["results"])
    assert covered == list(range(7))
    assert ranks[0]["elapsed"] == ranks[1]["elapsed"] and ranks[0]["elapsed"] >= 3 * 0.04   # MAX over ranks (rank 1 sleeps longer)


@pytest.mark.timeout(180)
def test_host_fed_leg_two_ranks_gloo():
    """The leg of bench.py that every rank runs for N > 1 (host frames -> pinned staging -> own device, all ranks at once after a
    barrier; reference capture_sources/video.py:9-29 is the fan-out it stands for): world_size 2 over gloo with stand-in devices -
    every rank feeds exactly its slice of every batch, rank 0 reports per-rank and aggregate rates."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(31500 + os.getpid() % 2000), os.path.join(ROOT, "tests", "_hostfed_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(next(l for l in out.stdout.splitlines() if l.startswith("RESULT "))[len("RESULT "):])
    assert [r["rank"] for r in rec["ranks"]] == [0, 1]
    assert [r["frames_of_each_batch"] for r in rec["ranks"]] == [[0, 4], [4, 7]]
    assert all(r["frames_per_s"] > 0 for r in rec["ranks"]) and rec["aggregate_frames_per_s"] > 0 and rec["batches_per_s"] > 0
    # a batch is complete when its slowest slice is: the aggregate cannot exceed batch / slowest rank's time per batch
    slow = max(r["seconds"] for r in rec["ranks"])
    assert abs(rec["aggregate_frames_per_s"] - 5 * 7 / slow) < 0.1 * rec["aggregate_frames_per_s"]
