"""Seeded sweep of single-image contour extraction against the oracle's sequential border following: random sizes (widths that
allow 8-, 16- and only 32-row strips), densities from isolated pixels to nearly full, blobs with holes, both retrieval modes and
approximations, device-resident and host masks.   usage: python tools/fuzz_contours.py [first seed] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from oracle import oracle as O
from vision.utils import feature
from vision.utils.color import range_threshold

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    h = int(rng.integers(1, 400))
    w = int(rng.choice([rng.integers(1, 700), 8 * rng.integers(1, 80), 64 * rng.integers(1, 12), 2 * rng.integers(1, 300) + 1]))
    kind = seed % 4
    if kind == 0:
        m = (rng.random((h, w)) < rng.choice([0.01, 0.1, 0.3, 0.5, 0.7, 0.95])).astype(np.uint8) * 255
    elif kind == 1:
        yy, xx = np.mgrid[0:h, 0:w]
        m = np.zeros((h, w), np.uint8)
        for _ in range(int(rng.integers(1, 12))):
            cx, cy, rx, ry = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(1, max(2, w / 3)), rng.uniform(1, max(2, h / 3))
            ring = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2
            m[(ring <= 1) & (ring >= rng.choice([0.0, 0.3, 0.7]))] = 255
    elif kind == 2:
        m = np.full((h, w), 255, np.uint8)
        m[rng.random((h, w)) < 0.02] = 0
        if h > 4 and w > 4:
            m[h // 3: h // 3 + max(1, h // 5), w // 4: w // 4 + max(1, w // 3)] = 0
    else:
        m = np.kron((rng.random(((h + 7) // 8, (w + 7) // 8)) < 0.5).astype(np.uint8) * 255, np.ones((8, 8), np.uint8))[:h, :w]
        m = np.ascontiguousarray(m)
    mode, method = int(rng.integers(0, 2)), int(rng.integers(1, 3))
    src = range_threshold(m, 128, 255) if seed % 3 == 0 else m          # a device-resident mask every third time
    got, gh = feature.find_contours(src, mode, method, with_holes=True)
    exp, eh = O.find_contours(m, mode, method, with_holes=True)
    ok = len(got) == len(exp) and all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(got, exp)) and np.array_equal(gh, eh)
    if not ok:
        bad += 1
        print(f"FAIL seed {seed}: {h}x{w} kind {kind} mode {mode} method {method}: {len(got)} vs {len(exp)} contours", flush=True)
        if bad > 5:
            break
print(f"contour sweep, seeds {first}..{first + count - 1}: {bad} failures")
