import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")]
import importlib, types
import test_gpu_fuzz as T
from oracle import oracle as O
from vision import _vp
_vp.lib(); _vp.default_context()
bad = 0
for seed in range(24, 424):
    try:
        T.test_random_chain(_vp, O, seed)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, str(e)[:200])
        if bad > 5: break
print("done, failures:", bad)
