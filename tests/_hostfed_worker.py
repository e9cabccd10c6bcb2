"""Worker of tests/test_multirank.py::test_host_fed_leg_two_ranks_gloo: bench.host_fed_leg - the leg every rank of a multi-GPU bench run
executes after a barrier - under torch.distributed.run with gloo, the device replaced by a stand-in that checks what it is fed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import bench

rank, local_rank, world, dist = bench.init_distributed("gloo")
seen = []


class Runner:
    """(device, n_frames, h, w) -> .input / .run(): records the first byte of every frame it was handed (frame k of a batch is filled
    with 7 k + 1 by the leg, so a wrong slice shows)."""

    def __init__(self, device, n, h, w):
        self.input = np.zeros((n, h, w, 3), np.uint8)

    def run(self):
        seen.append(self.input[:, 0, 0, 0].tolist())
        return {"first": self.input[:, 0, 0, 0].copy()}


rec = bench.host_fed_leg(dist, rank, world, 0, 64, 36, batch=7, batches=5, ring=2, make_runner=Runner, bind_numa=False)
lo, hi = bench.shard_of(7, rank, world)
assert rec["frames_of_each_batch"] == [lo, hi] and len(seen) == 6            # 5 timed batches + the first one
assert all(s == [(7 * k + 1) % 256 for k in range(lo, hi)] for s in seen), seen
if rank == 0:
    print("RESULT " + json.dumps(rec), flush=True)
dist.destroy_process_group()
