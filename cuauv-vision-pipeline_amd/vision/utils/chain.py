"""Fused colour -> inRange -> rect morphology -> CCL chain on host arrays (one H2D, one launch
sequence, compact results back).  Not a reference function: it is the batched form of what
modules/red_buoy.py:21-38 and modules/bins.py:13-27 do per frame, bound to vp_chain_run_host."""
import numpy as np

from vision import _vp


def run_chain(frames, color_mode, lo, hi, morph=(), ccl=1, numbering=_vp.CCL_BLOCK2X2, max_labels=256,
              want=("threshed", "cleaned", "labels", "stats"), contours=None):
    """contours: None, or a dict for _vp.make_contour_desc (source "cleaned" | "threshed", mode, method, max_contours,
    max_points): out["contours"][f] = (contours in cv2 order, hole flags) as utils/feature.py:5-40 would return for frame f;
    capacities grow and the call repeats when a frame needs more."""
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
    if contours is None:
        ctx.chain_run_host(desc, bufs, n)
        return out
    cdesc = _vp.make_contour_desc(**contours)
    while True:
        arrs, cb = _vp.contour_arrays(lambda shape, dt: np.zeros(shape, dt), n, cdesc)
        ctx.chain_run_contours_host(desc, bufs, cdesc, cb, n)
        need_c, need_p = int(arrs["info"][:, 0].max()), int(arrs["info"][:, 1].max())
        if need_c <= cdesc.max_contours and need_p <= cdesc.max_points:
            break
        cdesc.max_contours = max(cdesc.max_contours, 2 * need_c)
        cdesc.max_points = max(cdesc.max_points, 2 * need_p)
    out["contours"] = _vp.contour_lists(arrs, cdesc)
    out["contour_features"] = _vp.contour_features(arrs, cdesc)
    return out


class ChainRunner:
    """Host-fed chain with page-locked staging: `runner.input` is an (n, h, w, 3) uint8 pinned array the producer fills
    in place (e.g. straight from the CMF read), `runner.run()` moves it over PCIe, runs the chain and brings the requested
    outputs back into pinned arrays (views valid until the next run).  With stats-only outputs the rate is bound by the
    6.2 MB/frame upload; full masks + labels add 12.4 MB/frame of download."""

    def __init__(self, n, height, width, color_mode, lo, hi, morph=(), ccl=1, numbering=_vp.CCL_BLOCK2X2, max_labels=256,
                 want=("stats",), device=0, contours=None):
        self.ctx = _vp.Context(device)
        self.n, self.h, self.w = int(n), int(height), int(width)
        self.desc = _vp.make_chain_desc(width, height, color_mode, lo, hi, morph, ccl, numbering, max_labels)
        self.input = _vp.pinned_empty(self.ctx, (n, height, width, 3), np.uint8)
        self.out = {}
        self.bufs = _vp.ChainBuffers()
        self.bufs.bgr = self.input.ctypes.data
        shapes = {"threshed": ((n, height, width), np.uint8), "cleaned": ((n, height, width), np.uint8)}
        if ccl:
            shapes.update({"labels": ((n, height, width), np.int32), "stats": ((n, max_labels, 5), np.int32),
                           "centroids": ((n, max_labels, 2), np.float64), "nlabels": ((n,), np.int32)})
        for name, (shape, dt) in shapes.items():
            if name in want or name == "nlabels" or (name == "centroids" and "stats" in want):
                self.out[name] = _vp.pinned_empty(self.ctx, shape, dt)
                setattr(self.bufs, name, self.out[name].ctypes.data)

        self.cdesc = None
        if contours is not None:
            self.cdesc = _vp.make_contour_desc(**contours)
            self.carrs, self.cbufs = _vp.contour_arrays(lambda shape, dt: _vp.pinned_empty(self.ctx, shape, dt), n, self.cdesc)

    def run(self):
        if self.cdesc is None:
            self.ctx.chain_run_host(self.desc, self.bufs, self.n)
        else:
            self.ctx.chain_run_contours_host(self.desc, self.bufs, self.cdesc, self.cbufs, self.n)
            self.out["contours"] = _vp.contour_lists(self.carrs, self.cdesc)   # None where a frame exceeded the capacities
            self.out["contour_features"] = _vp.contour_features(self.carrs, self.cdesc)
        return self.out
