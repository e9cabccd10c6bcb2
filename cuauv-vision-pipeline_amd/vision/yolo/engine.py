"""The surface of `ultralytics.YOLO` that modules/yolo.py:49-57,112-114 touches - `YOLO(path)`, `.to(device)`, `.task`,
`.track(image, verbose=False)[0].summary()` - around the PyTorch network of vision/yolo/model.py, with the two steps ultralytics runs
on the CPU / in Python moved to the HIP kernels of this repository:

    frame (h, w, 3) BGR uint8  --vp_letterbox_dev-->  (3, H, W) float RGB in [0, 1]   (scale to fit 640, pad 114 to a multiple of 32)
      --network (PyTorch-ROCm)-->  (4 + nc + 1, anchors): x, y, w, h, class probabilities, angle
      --confidence filter, best class, class offset-->  candidates  --vp_nms_dev (rotated, probabilistic IoU)-->  kept, best first
      --regularise (w >= h, angle in [0, pi)), undo the letterbox, corners, track ids-->  `Results.summary()` entries

ultralytics and the reference's weight file (`obb_v14.pt`) exist neither in its tree nor in this image: without a state dict the
network is randomly initialised (seeded) and its detections mean nothing; the path around it is what is built and tested
(tests/test_gpu_yolo_module.py compares every step with a plain PyTorch restatement on the same raw network output).
"""
import math
import os

import numpy as np

from vision.yolo import ops

DEFAULT_NAMES = ("torpedo_board", "shark_hole", "saw_hole", "pole_red", "pole_white", "shark", "saw", "gate_behind", "bin_shark", "bin_saw",
                 "bin", "spoon", "cup", "pink_basket", "yellow_basket")     # the classes modules/yolo.py:130-151 dispatches on


def letterbox_shape(h, w, imgsz=640, stride=32):
    """(H, W) of the network input for an (h, w) frame: longer side scaled to `imgsz`, the other padded up to a multiple of `stride`
    (the minimal-rectangle letterbox a PyTorch model gets)."""
    r = min(imgsz / h, imgsz / w)
    nh, nw = int(round(h * r)), int(round(w * r))
    return nh + (imgsz - nh) % stride, nw + (imgsz - nw) % stride


def xywhr_to_corners(b):
    """(n, 5) x, y, w, h, angle -> (n, 4, 2) corners: centre +/- the two half-axes, in the order ++, +-, --, -+."""
    import torch
    ctr, w, h, ang = b[:, :2], b[:, 2:3], b[:, 3:4], b[:, 4:5]
    cos, sin = torch.cos(ang), torch.sin(ang)
    v1 = torch.cat([w / 2 * cos, w / 2 * sin], 1)
    v2 = torch.cat([-h / 2 * sin, h / 2 * cos], 1)
    return torch.stack([ctr + v1 + v2, ctr + v1 - v2, ctr - v1 - v2, ctr - v1 + v2], 1)


def regularize(b):
    """Same boxes with w >= h and the angle in [0, pi)."""
    import torch
    x, y, w, h, t = b.unbind(1)
    swap = h > w
    w_, h_ = torch.where(swap, h, w), torch.where(swap, w, h)
    t = torch.where(swap, t + math.pi / 2, t) % math.pi
    return torch.stack([x, y, w_, h_, t], 1)


class Tracker:
    """Stand-in for the tracker behind `model.track` (identities only; no handler reads them): a detection keeps the id of the
    previous frame's detection of the same class whose axis-aligned bounds overlap it most (IoU >= 0.3), otherwise gets a new id."""

    def __init__(self):
        self.prev = []            # (id, class, x1, y1, x2, y2)
        self.next_id = 1

    def update(self, classes, bounds):
        ids, used = [], set()
        for c, (x1, y1, x2, y2) in zip(classes, bounds):
            best, best_iou = None, 0.3
            for k, (pid, pc, a1, b1, a2, b2) in enumerate(self.prev):
                if pc != c or k in used:
                    continue
                iw, ih = min(x2, a2) - max(x1, a1), min(y2, b2) - max(y1, b1)
                if iw <= 0 or ih <= 0:
                    continue
                inter = iw * ih
                iou = inter / ((x2 - x1) * (y2 - y1) + (a2 - a1) * (b2 - b1) - inter)
                if iou >= best_iou:
                    best, best_iou = k, iou
            if best is None:
                ids.append(self.next_id)
                self.next_id += 1
            else:
                used.add(best)
                ids.append(self.prev[best][0])
        self.prev = [(i, c, *b) for i, c, b in zip(ids, classes, bounds)]
        return ids


class Results:
    """One frame's detections, best first: `boxes` (n, 5) x, y, w, h, angle in frame pixels, `conf`, `cls`, `corners` (n, 4, 2)."""

    def __init__(self, names, boxes, conf, cls, corners, track_ids, orig_shape):
        self.names, self.boxes, self.conf, self.cls, self.corners, self.track_ids, self.orig_shape = names, boxes, conf, cls, corners, track_ids, orig_shape

    def __len__(self):
        return len(self.conf)

    def summary(self, normalize=False, decimals=5):
        h, w = self.orig_shape
        sx, sy = (w, h) if normalize else (1, 1)
        out = []
        for i in range(len(self.conf)):
            c = self.corners[i]
            entry = {"name": self.names[int(self.cls[i])], "class": int(self.cls[i]), "confidence": round(float(self.conf[i]), decimals),
                     "box": {f"{ax}{k + 1}": round(float(c[k][j]) / s, decimals) for k in range(4) for j, (ax, s) in enumerate((("x", sx), ("y", sy)))}}
            if self.track_ids is not None:
                entry["track_id"] = int(self.track_ids[i])
            out.append(entry)
        return out


class YOLO:
    task = "obb"

    def __init__(self, weights=None, names=DEFAULT_NAMES, imgsz=640, conf=0.25, iou=0.7, max_det=300, seed=0):
        import torch
        from vision.yolo.model import YOLOv8nOBB
        self.names = {i: n for i, n in enumerate(names)}
        self.imgsz, self.conf, self.iou, self.max_det = int(imgsz), float(conf), float(iou), int(max_det)
        gen = torch.Generator().manual_seed(seed)
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(seed)
            self.model = YOLOv8nOBB(len(names)).eval()
        del gen
        self.weights = str(weights) if weights else None
        if self.weights and os.path.exists(self.weights):
            self.model.load_state_dict(torch.load(self.weights, map_location="cpu"))
        self.device = torch.device("cpu")
        self.tracker = Tracker()
        self.graphs = os.environ.get("VP_YOLO_GRAPHS", "1") != "0"
        self._graphs = {}

    def to(self, device):
        import torch
        self.device = torch.device(device)
        self.model.to(self.device)
        self._graphs = {}
        return self

    # -- the steps, separately callable (the tests compare each with a restatement) ------------------------------------------------
    def preprocess(self, image):
        """frame -> ((1, 3, H, W) float tensor on the model's device, (scale, left, top)); the letterbox runs as a HIP kernel when the
        model is on the GPU."""
        import torch
        from vision.devmat import DeviceMat, to_host_readonly
        shape = letterbox_shape(image.shape[0], image.shape[1], self.imgsz)
        if self.device.type == "cuda":
            if isinstance(image, DeviceMat) and image._host is None:
                t, geom = ops.letterbox(image, shape)        # already in HBM (the runtime's frames are): no host hop
                if not hasattr(t, "is_cuda"):
                    t = torch.from_numpy(t)
                t = t.to(self.device)
            else:
                host = np.array(to_host_readonly(image), copy=True) if isinstance(image, DeviceMat) else np.ascontiguousarray(image)
                t, geom = ops.letterbox(torch.from_numpy(host).to(self.device), shape)
        else:
            raise RuntimeError("the detector's pre- and post-processing run on the GPU: move the model there with .to('cuda')")
        return t.unsqueeze(0), geom

    def postprocess(self, pred, geom, orig_shape, track=False):
        """(4 + nc + 1, anchors) raw output of one frame -> Results."""
        import torch
        nc = len(self.names)
        p = pred.t()                                                  # (A, 4 + nc + 1)
        conf, cls = p[:, 4:4 + nc].max(1)
        keep = conf > self.conf
        p, conf, cls = p[keep], conf[keep], cls[keep]
        if p.shape[0] > 30000:
            top = conf.topk(30000).indices
            p, conf, cls = p[top], conf[top], cls[top]
        boxes = torch.cat([p[:, :4], p[:, -1:]], 1)
        if boxes.shape[0]:
            shifted = boxes.clone()
            shifted[:, :2] += cls[:, None].to(boxes.dtype) * 7680.0    # other classes far away: suppression within a class only
            idx = ops.nms_rotated(shifted, conf, self.iou, self.max_det)
            boxes, conf, cls = boxes[idx], conf[idx], cls[idx]
        boxes = regularize(boxes)
        r, left, top_ = geom
        boxes = boxes.clone()
        boxes[:, 0] = (boxes[:, 0] - left) / r
        boxes[:, 1] = (boxes[:, 1] - top_) / r
        boxes[:, 2:4] /= r
        corners = xywhr_to_corners(boxes)
        boxes_h, conf_h, cls_h, corners_h = boxes.cpu().numpy(), conf.cpu().numpy(), cls.cpu().numpy(), corners.cpu().numpy()
        ids = None
        if track:
            bounds = [(float(c[:, 0].min()), float(c[:, 1].min()), float(c[:, 0].max()), float(c[:, 1].max())) for c in corners_h]
            ids = self.tracker.update([int(c) for c in cls_h], bounds)
        return Results(self.names, boxes_h, conf_h, cls_h, corners_h, ids, tuple(orig_shape[:2]))

    def forward(self, x):
        """One pass of the network.  At batch 1 the ~200 small kernels of YOLOv8n are launch-bound, so on the GPU the pass is captured
        once per input shape into a HIP graph (torch.cuda.CUDAGraph) and replayed: the input is copied into the graph's fixed buffer, the
        returned tensor is the graph's output buffer (valid until the next forward).  `self.graphs = False` runs the eager pass."""
        import torch
        with torch.no_grad():
            if self.device.type != "cuda" or not self.graphs:
                return self.model(x)
            key = tuple(x.shape)
            g = self._graphs.get(key)
            if g is None:
                static_in = x.clone()
                side = torch.cuda.Stream(self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    for _ in range(2):                     # lazy initialisations (MIOpen's choice of algorithm, anchors) before the capture
                        self.model(static_in)
                torch.cuda.current_stream(self.device).wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_out = self.model(static_in)
                g = self._graphs[key] = (graph, static_in, static_out)
            graph, static_in, static_out = g
            static_in.copy_(x)
            graph.replay()
            return static_out

    def predict(self, image, verbose=False, track=False):
        x, geom = self.preprocess(image)
        pred = self.forward(x)
        return [self.postprocess(pred[0], geom, image.shape, track)]

    def track(self, image, verbose=False, persist=True):
        return self.predict(image, verbose=verbose, track=True)

    __call__ = predict
