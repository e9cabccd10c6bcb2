"""BASELINE config 5 end to end: the detector module (reference modules/yolo.py:37-165) with the PyTorch-ROCm YOLOv8n-OBB of
vision/yolo/model.py between the HIP letterbox and rotated-NMS kernels, and the torpedo-board handler (handlers/torpedoes.py).
ultralytics and the weight file are outside the reference tree and this image: the network is randomly initialised (seeded), so what
is checked is the path around it - every step against a plain PyTorch / numpy restatement on the same data - not detections."""
import os
import sys
import threading
import time

import numpy as np
import pytest

import frames as F

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims"))
PID = os.getpid()


def _model(**kw):
    from vision.yolo.engine import YOLO
    return YOLO(None, conf=kw.pop("conf", 0.0), **kw).to("cuda")       # a random network is never confident: keep every candidate


def test_steps_against_restatements(vp):
    import torch
    from test_gpu_yolo import _probiou
    from vision.yolo.engine import letterbox_shape, regularize, xywhr_to_corners
    m = _model(max_det=300)
    img = F.s1_buoy(4, 1280, 720)
    x, (r, left, top) = m.preprocess(img)
    assert tuple(x.shape) == (1, 3) + letterbox_shape(720, 1280) == (1, 3, 384, 640)
    # letterbox: scale to fit, centred, padded with 114 / 255, BGR -> RGB, [0, 1]
    assert abs(r - 0.5) < 1e-6 and left == 0 and top == 12
    assert torch.allclose(x[0, :, :12], torch.full((3, 12, 640), 114 / 255, device="cuda")) and torch.allclose(x[0, :, 372:], torch.full((3, 12, 640), 114 / 255, device="cuda"))
    ref = torch.nn.functional.interpolate(torch.from_numpy(img[:, :, ::-1].copy()).cuda().permute(2, 0, 1)[None].float() / 255, size=(360, 640),
                                          mode="bilinear", align_corners=False, antialias=False)[0]
    assert float((x[0, :, 12:372] - ref).abs().max()) <= 1.5 / 255       # 8-bit fixed-point resize vs float bilinear
    pred = m.forward(x)[0]
    m.graphs = False
    eager = m.forward(x)[0]
    m.graphs = True
    assert torch.equal(pred, eager) and torch.equal(m.forward(x)[0], eager)          # captured pass = eager pass, also when replayed
    assert tuple(pred.shape) == (4 + 15 + 1, 384 * 640 // 64 + 384 * 640 // 256 + 384 * 640 // 1024)
    res = m.postprocess(pred, (r, left, top), img.shape, track=True)
    # restatement of the post-processing in plain PyTorch: best class, class-wise probabilistic-IoU suppression, regularise, unletterbox
    p = pred.t()
    conf, cls = p[:, 4:19].max(1)
    order = conf.argsort(descending=True, stable=True)
    p, conf, cls = p[order], conf[order], cls[order]
    boxes = torch.cat([p[:, :4], p[:, -1:]], 1)
    shifted = boxes.clone()
    shifted[:, :2] += cls[:, None].float() * 7680.0
    iou = _probiou(shifted, shifted).triu_(diagonal=1)
    worst = iou.max(0)[0]
    keep = (worst < m.iou).nonzero().squeeze(1)
    doubtful = ((worst - m.iou).abs() < 1e-4).any()
    keep = keep[: m.max_det]
    exp = regularize(boxes[keep])
    exp[:, 0] = (exp[:, 0] - left) / r
    exp[:, 1] = (exp[:, 1] - top) / r
    exp[:, 2:4] /= r
    if not bool(doubtful):                       # (a candidate within 1e-4 of the threshold may legitimately fall either way)
        assert len(res) == len(keep)
        assert np.allclose(res.boxes, exp.cpu().numpy(), rtol=0, atol=1e-3)
        assert np.array_equal(res.cls, cls[keep].cpu().numpy()) and np.allclose(res.conf, conf[keep].cpu().numpy())
        assert np.allclose(res.corners, xywhr_to_corners(exp).cpu().numpy(), atol=1e-2)
    assert len(res) <= m.max_det and (np.diff(res.conf) <= 1e-7).all()         # best first
    s = res.summary()
    assert len(s) == len(res) and set(s[0]) == {"name", "class", "confidence", "box", "track_id"} and set(s[0]["box"]) == {f"{a}{k}" for a in "xy" for k in "1234"}
    assert s[0]["name"] == m.names[s[0]["class"]]
    # a second frame: same objects keep their identities
    ids0 = {(e["class"], e["track_id"]) for e in s}
    s2 = m.track(img)[0].summary()
    assert {(e["class"], e["track_id"]) for e in s2} == ids0


def test_detector_module_with_torpedoes_handler(vp, monkeypatch):
    """The module on the runtime (shared-memory block -> loop thread -> fwd_process): records of the three torpedo classes reach the
    handler, whose shm outputs equal a recomputation from those very records; with the object inactive the handler posts a grey image."""
    monkeypatch.setattr(sys, "argv", ["yolo.py"])
    import shm
    from vision.core.bindings.camera_message_framework import BlockAccessor
    from vision.handlers.torpedoes import TorpedoesOBB
    import module_harness as MH
    TUNERS = MH.detector_tuners()
    from vision.yolo.ops import order_points
    shm.active_objects.yolo_torpedoes_board.set(True)
    shm.active_objects.yolo_torpedoes_board_direction.set("forward")
    calls = []

    class Recording(TorpedoesOBB):
        def process(self, direction, img, boards, sharks, saws):
            calls.append((direction, img.shape, list(boards), list(sharks), list(saws)))
            return super().process(direction, img, boards, sharks, saws)

    d = f"pytyolo{PID}"
    frame = F.s1_buoy(6, 640, 360)
    depth = np.zeros((360, 640), np.float32)      # zed carries several planes; a one-plane frame is cached under the block's name (core/base.py:765-803)
    with BlockAccessor(d, max_entry_size_bytes=frame.nbytes + depth.nbytes) as w:
        mod = MH.detector_module(_model)([d], TUNERS, [Recording("torpedoes")])
        mod._fps = 100
        runner = threading.Thread(target=mod)
        runner.start()
        try:
            t0 = time.time()
            while not calls and time.time() - t0 < 60:
                w.write_frame(int(time.monotonic() * 1000), [("forward", frame), ("depth", depth)])
                time.sleep(0.05)
        finally:
            mod.stop()
            runner.join(10)
    assert calls, "the handler was never called"
    direction, shape, boards, sharks, saws = calls[0]
    assert direction == "forward" and shape == frame.shape
    assert all(b.name == "torpedo_board" for b in boards) and all(b.name == "shark_hole" for b in sharks) and all(b.name == "saw_hole" for b in saws)
    assert boards or sharks or saws, "conf = 0 keeps every candidate: some must carry a torpedo class"
    g = shm.yolo_torpedoes_board.get()
    H, W = frame.shape[:2]
    for prefix, lst in (("board", boards), ("shark", sharks), ("saw", saws)):
        best = max(lst, key=lambda x: x.confidence) if lst else None
        if best is None or best.confidence < 0.1:
            assert getattr(g, f"{prefix}_visible") == 0
            continue
        tl, tr, bl, br = order_points([(best.x1, best.y1), (best.x2, best.y2), (best.x3, best.y3), (best.x4, best.y4)])
        assert getattr(g, f"{prefix}_visible") == 1 and getattr(g, f"{prefix}_confidence") == best.confidence
        assert getattr(g, f"{prefix}_top_left_x") == (tl[0] - W / 2) / W and getattr(g, f"{prefix}_top_left_y") == (tl[1] - H / 2) / W
        assert getattr(g, f"{prefix}_bottom_right_x") == (br[0] - W / 2) / W
    # the object switched off: only the grey debug image is posted, no records are built
    shm.active_objects.yolo_torpedoes_board.set(False)
    n_before = len(calls)
    posted = []
    mod2 = MH.detector_module(_model)([d + "b"], MH.detector_tuners(), [Recording("torpedoes")])
    record = lambda name, image, cs="BGR": posted.append((name, np.asarray(image).shape))   # noqa: E731
    monkeypatch.setattr(mod2, "post", record)
    monkeypatch.setattr(mod2.handlers["torpedoes"], "post", record)          # handlers borrow the parent's post at registration
    mod2._current_direction = "forward"
    mod2.fwd_process(frame.copy())
    assert len(calls) == n_before and ("torpedoes handler", (360, 640)) in posted


def test_rate_of_the_detector_path(vp):
    """Not a benchmark line (BASELINE names no metric for config 5): the per-frame cost of letterbox + network + post-processing at
    1080p, printed for DESIGN.md."""
    import torch
    m = _model(conf=0.25)                        # the reference's default threshold: a random network then yields no candidates
    img = F.s1_buoy(0)
    for _ in range(3):
        m.track(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m.track(img)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    x, g = m.preprocess(img)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(20):
        m.preprocess(img)
    torch.cuda.synchronize(); tp = (time.perf_counter() - t1) / 20
    m.graphs = False
    m.track(img); torch.cuda.synchronize(); t2 = time.perf_counter()
    for _ in range(20):
        m.track(img)
    torch.cuda.synchronize(); te = (time.perf_counter() - t2) / 20
    print(f"\nconfig 5, 1080p frame: {1e3 * dt:.2f} ms per frame end to end ({1 / dt:.0f} frames/s; {1e3 * te:.2f} ms with the network run eagerly), "
          f"of which upload + HIP letterbox {1e3 * tp:.2f} ms")
    assert dt < 0.5
