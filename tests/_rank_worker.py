"""Worker of tests/test_multirank.py: run under torch.distributed.run with the gloo backend.  Exercises exactly the N > 1
helpers bench.py uses (init_distributed, shard_of, timed_steps: barrier + MAX over ranks) on a CPU stand-in step."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import bench
import frames as F
from oracle import oracle as orc

rank, local_rank, world, dist = bench.init_distributed("gloo")
assert dist is not None and world == int(os.environ["WORLD_SIZE"])
N_FRAMES = 7                                    # deliberately not divisible by the world size
lo, hi = bench.shard_of(N_FRAMES, rank, world)
mine = [F.s1_buoy(i, 128, 72) for i in range(lo, hi)]
results = {}


def step():
    for i, f in zip(range(lo, hi), mine):
        out = orc.chain(f, orc.MODE_LAB, (0, 150, 0), (255, 255, 255), [orc.OPEN, orc.CLOSE], 5, 5, 2, 64, want_labels=False)
        results[i] = int(out["nlabels"])
    time.sleep(0.02 * (rank + 1))               # ranks finish at different times: the MAX must be reported


elapsed = bench.timed_steps(step, lambda: None, 3, dist)
gathered = [None] * world
dist.all_gather_object(gathered, {"rank": rank, "range": [lo, hi], "results": results, "elapsed": elapsed})
if rank == 0:
    print("RESULT " + json.dumps(gathered), flush=True)
dist.destroy_process_group()
