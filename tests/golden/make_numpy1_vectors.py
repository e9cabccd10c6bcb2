"""Golden vectors for the numpy part of `thresh_color_distance` (reference utils/color.py:91-103), made by running those numpy
statements under a real numpy 1.x, the major version the reference is written for (it uses np.int0, removed in numpy 2).

What is uncertain about these statements is not arithmetic but numpy's promotion rules: `weights_cp[i]` is a float64 numpy scalar,
`np.float32(split[i]) - color[i]` a float32 array minus a Python int; under numpy 1.x value-based casting keeps the product float32,
and `np.percentile` of a float32 array interpolates as numpy does.  This script states nothing about cv2 (the final `cv2.inRange` is
not part of it) - it records what numpy itself computes, so that the oracle and the HIP path are pinned to the interpreter's answer
instead of to a reading of the promotion rules.

    /opt/conda/bin/python3.9 tests/golden/make_numpy1_vectors.py        # numpy 1.26.4 in this image; writes numpy1_color_distance.npz
"""
import os
import sys

import numpy as np

assert int(np.__version__.split(".")[0]) == 1, f"needs numpy 1.x, this is {np.__version__}"
out = {}
rng = np.random.default_rng(20261004)
cases = [((37, 53), (100, 150, 60), [], (1, 1, 1), None, 70.0), ((64, 64), (12, 250, 3), [0], (1, 2, 3), None, 40.0),
         ((33, 130), (200.5, 17.25, 90), [1, 2], (0.5, 1, 1), None, 25.0), ((48, 96), (100, 150, 60), [], (1, 1, 1), 20, 500.0),
         ((48, 96), (30, 40, 220), [2], (3, 1, 2), 73.5, 90.0), ((5, 7), (0, 0, 0), [], (1e-3, 1, 1000), 50, 1e9)]
for k, (shape, color, ignore_channels, weights, auto_distance_percentile, distance) in enumerate(cases):
    split = [rng.integers(0, 256, shape).astype(np.uint8) for _ in range(3)]
    # ---- the statements of the reference, numpy only -------------------------------------------------------------------------
    weights_cp = list(weights)
    for idx in ignore_channels:
        weights_cp[idx] = 0
    weights_cp /= np.linalg.norm(weights)
    dists = np.zeros(split[0].shape, dtype=np.float32)
    for i in range(3):
        if i in ignore_channels:
            continue
        dists += weights_cp[i] * (np.float32(split[i]) - color[i])**2
    if auto_distance_percentile:
        distance = min(np.percentile(dists, auto_distance_percentile), distance**2)
    else:
        distance = distance**2
    sq = np.uint8(np.sqrt(dists))
    # ---------------------------------------------------------------------------------------------------------------------------
    assert dists.dtype == np.float32
    out[f"c{k}_split"] = np.stack(split)
    out[f"c{k}_args"] = np.array([*color, *weights, -1.0 if auto_distance_percentile is None else auto_distance_percentile, cases[k][5]], np.float64)
    out[f"c{k}_ignore"] = np.array(ignore_channels, np.int32)
    out[f"c{k}_weights_cp"] = np.asarray(weights_cp, np.float64)
    out[f"c{k}_dists"] = dists
    out[f"c{k}_distance"] = np.array([distance], np.float64)
    out[f"c{k}_sq"] = sq
out["numpy_version"] = np.array([np.__version__])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "numpy1_color_distance.npz"), **out)
print("written with numpy", np.__version__)
