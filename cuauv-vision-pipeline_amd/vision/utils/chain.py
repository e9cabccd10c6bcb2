"""Fused colour -> inRange -> rect morphology -> CCL chain on host arrays (one H2D, one launch
sequence, compact results back).  Not a reference function: it is the batched form of what
modules/red_buoy.py:21-38 and modules/bins.py:13-27 do per frame, bound to vp_chain_run_host."""
import numpy as np

from vision import _vp


def run_chain(frames, color_mode, lo, hi, morph=(), ccl=1, numbering=_vp.CCL_BLOCK2X2, max_labels=256,
              want=("threshed", "cleaned", "labels", "stats")):
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    if frames.ndim == 3:
        frames = frames[None]
    n, h, w, c = frames.shape
    if c != 3:
        raise ValueError("expected (n, h, w, 3) BGR frames")
    desc = _vp.make_chain_desc(w, h, color_mode, lo, hi, morph, ccl, numbering, max_labels)
    bufs = _vp.ChainBuffers()
    out = {}
    bufs.bgr = frames.ctypes.data
    if "threshed" in want:
        out["threshed"] = np.empty((n, h, w), np.uint8)
        bufs.threshed = out["threshed"].ctypes.data
    if "cleaned" in want:
        out["cleaned"] = np.empty((n, h, w), np.uint8)
        bufs.cleaned = out["cleaned"].ctypes.data
    if ccl:
        if "labels" in want:
            out["labels"] = np.empty((n, h, w), np.int32)
            bufs.labels = out["labels"].ctypes.data
        if "stats" in want:
            out["stats"] = np.zeros((n, max_labels, 5), np.int32)
            out["centroids"] = np.zeros((n, max_labels, 2), np.float64)
            bufs.stats = out["stats"].ctypes.data
            bufs.centroids = out["centroids"].ctypes.data
        out["nlabels"] = np.zeros((n,), np.int32)
        bufs.nlabels = out["nlabels"].ctypes.data
    ctx = _vp.default_context()
    ctx.chain_run_host(desc, bufs, n)
    return out
