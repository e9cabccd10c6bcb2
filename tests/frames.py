"""Deterministic synthetic frames of SURVEY §8d (S1 buoy, S2 bins, S3 adversarial, S4 flat)."""
import numpy as np


def s1_buoy(i, w=1920, h=1080, k=12):
    rng = np.random.default_rng(1000 + i)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.empty((h, w, 3), np.float32)
    base = (150.0, 110.0, 40.0)
    gx, gy = rng.uniform(-20, 20, 2)
    grad = gx * (xx / w - 0.5) + gy * (yy / h - 0.5)
    for c in range(3):
        img[:, :, c] = base[c] + grad + rng.normal(0, 6, (h, w)).astype(np.float32)
    for _ in range(k):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        rx, ry = rng.uniform(15, 120, 2) * (w / 1920.0)
        col = (rng.uniform(20, 60), rng.uniform(20, 70), rng.uniform(170, 255))
        m = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0
        for c in range(3):
            img[:, :, c][m] = col[c] + rng.normal(0, 3)
    out = np.clip(img, 0, 255).astype(np.uint8)
    salt = rng.random((h, w)) < 0.001
    out[salt] = 255
    return out


def s2_bins(i, w=1920, h=1080, k=6):
    rng = np.random.default_rng(2000 + i)
    out = rng.integers(0, 40, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for _ in range(k):
        cx, cy = rng.uniform(0.1 * w, 0.9 * w), rng.uniform(0.1 * h, 0.9 * h)
        hw = rng.uniform(40, 160) * (w / 1920.0)
        hh = hw / 2
        th = rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        m = (np.abs(u) <= hw) & (np.abs(v) <= hh)
        out[m] = (140, 170, 200)  # BGR beige: HSV (15, 76, 200)
    return out


def s3_noise(i, w=1920, h=1080):
    return np.random.default_rng(3000 + i).integers(0, 256, (h, w, 3), dtype=np.uint8)


def s4_flat(value, w=1920, h=1080):
    return np.full((h, w, 3), value, np.uint8)


def random_mask(rng, h, w, p=None):
    p = rng.random() if p is None else p
    return ((rng.random((h, w)) < p) * 255).astype(np.uint8)
