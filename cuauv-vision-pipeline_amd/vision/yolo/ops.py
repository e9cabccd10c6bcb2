"""letterbox / nms / nms_rotated bound to libvp (vp_letterbox_*, vp_nms_*).  Inputs may be numpy arrays (host entry points) or
torch tensors on the GPU (device entry points, no host hop, result tensors on the same device)."""
import ctypes as C

import numpy as np

from vision import _vp


def _is_torch_cuda(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def _on_torch_stream(ctx, launch):
    """Runs `launch` so that it is ordered with the caller's torch work: on torch's current stream when that is a real stream
    handle; around the legacy default stream (handle 0, which libvp cannot adopt) with explicit synchronisation."""
    import torch
    handle = torch.cuda.current_stream().cuda_stream
    if handle:
        ctx.set_stream(handle)
        try:
            launch()
        finally:
            ctx.set_stream(None)
    else:
        torch.cuda.current_stream().synchronize()
        launch()
        ctx.synchronize()


def letterbox(image, new_shape=(640, 640), pad_value=114):
    """(h, w, 3) BGR uint8 -> ((3, H, W) float32 RGB in [0, 1], (r, left, top))."""
    H, W = int(new_shape[0]), int(new_shape[1])
    geom = (C.c_float * 3)()
    ctx = _vp.default_context()
    if _is_torch_cuda(image):
        import torch
        if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3:
            raise ValueError("expected an (h, w, 3) uint8 tensor")
        image = image.contiguous()
        out = torch.empty((3, H, W), dtype=torch.float32, device=image.device)
        _on_torch_stream(ctx, lambda: _vp.check(_vp.lib().vp_letterbox_dev(ctx.handle, image.data_ptr(), image.shape[1], image.shape[0], W, H,
                                                                            int(pad_value), out.data_ptr(), geom), ctx.handle))
        return out, tuple(float(g) for g in geom)
    from vision.devmat import DeviceMat
    if isinstance(image, DeviceMat) and image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] == 3 and image._host is None \
            and image.device_valid_for(ctx):
        # a frame the runtime put into HBM (or an operator's result): letterboxed where it lies, the result is a tensor on that device
        import torch
        src = image.dev_ptr                              # (launches whatever still has to produce the image, on the context's stream)
        ctx.synchronize()                                # ... which must be through before torch's stream reads it
        out = torch.empty((3, H, W), dtype=torch.float32, device=torch.device("cuda", ctx.device))
        _on_torch_stream(ctx, lambda: _vp.check(_vp.lib().vp_letterbox_dev(ctx.handle, src, image.shape[1], image.shape[0], W, H,
                                                                            int(pad_value), out.data_ptr(), geom), ctx.handle))
        return out, tuple(float(g) for g in geom)
    image = np.ascontiguousarray(image, dtype=np.uint8)
    if image.ndim != 3 or image.shape[2] != 3 or image.size == 0:
        raise ValueError("expected a non-empty (h, w, 3) uint8 image")
    out = np.empty((3, H, W), np.float32)
    _vp.check(_vp.lib().vp_letterbox_u8_f32(ctx.handle, _vp.ptr(image), image.shape[1], image.shape[0], W, H, int(pad_value), _vp.ptr(out), geom),
              ctx.handle)
    return out, tuple(float(g) for g in geom)


def _nms(boxes, scores, thr, rotated, max_det):
    bs = 5 if rotated else 4
    ctx = _vp.default_context()
    if _is_torch_cuda(boxes):
        import torch
        boxes = boxes.float().contiguous()
        scores = scores.float().contiguous()
        n = int(boxes.shape[0])
        if boxes.dim() != 2 or boxes.shape[1] != bs or scores.shape[0] != n:
            raise ValueError("boxes / scores shapes")
        keep = torch.empty((max(max_det, 1),), dtype=torch.int32, device=boxes.device)
        nk = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
        _on_torch_stream(ctx, lambda: _vp.check(_vp.lib().vp_nms_dev(ctx.handle, boxes.data_ptr(), scores.data_ptr(), n, float(thr), int(rotated),
                                                                      int(max(max_det, 1)), keep.data_ptr(), nk.data_ptr()), ctx.handle))
        return keep[: min(int(nk.item()), max_det)].long()
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, bs)
    scores = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    n = boxes.shape[0]
    if scores.shape[0] != n:
        raise ValueError("boxes / scores shapes")
    keep = np.empty(max(max_det, 1), np.int32)
    nk = C.c_int32(0)
    _vp.check(_vp.lib().vp_nms_f32(ctx.handle, _vp.ptr(boxes), _vp.ptr(scores), n, float(thr), int(rotated), int(max_det), _vp.ptr(keep),
                                   C.byref(nk)), ctx.handle)
    return keep[: nk.value].astype(np.int64)


def nms(boxes, scores, iou_threshold=0.45, max_det=300):
    """Greedy non-maximum suppression on (n, 4) x1, y1, x2, y2 boxes: indices kept, best score first."""
    return _nms(boxes, scores, iou_threshold, 0, max_det)


def nms_rotated(boxes, scores, threshold=0.45, max_det=300):
    """(n, 5) x, y, w, h, angle boxes, probabilistic IoU: a box is dropped when a higher-scored box overlaps it by >= threshold."""
    return _nms(boxes, scores, threshold, 1, max_det)


def scale_boxes(xyxy, geom):
    """Map letterboxed x1, y1, x2, y2 back to the source frame: subtract the padding, divide by the scale."""
    r, left, top = geom
    b = np.asarray(xyxy, np.float32).copy()
    b[..., [0, 2]] = (b[..., [0, 2]] - left) / r
    b[..., [1, 3]] = (b[..., [1, 3]] - top) / r
    return b


def order_points(points):
    """(tl, tr, bl, br) of four corner points — what handlers/torpedoes.py:81 expects from its `order_points` helper: split by the
    sum / difference of the coordinates."""
    p = np.asarray(points, np.float64).reshape(4, 2)
    s, d = p.sum(1), p[:, 0] - p[:, 1]
    return tuple(p[np.argmin(s)]), tuple(p[np.argmax(d)]), tuple(p[np.argmin(d)]), tuple(p[np.argmax(s)])
