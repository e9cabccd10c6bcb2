"""Repeatability of the chain over a large batch: every copy of a frame must give the same labels / stats as its first copy.
usage: stress_ccl.py [batch] [rounds]   (VP_CCL_LEVELS=1 for the one-level kernels)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import frames as F
from vision import _vp
W, H, D = 1920, 1080, 8
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = _vp.Context(0)
small = torch.from_numpy(np.stack([F.s1_buoy(i, W, H) for i in range(D)])).cuda()
big = small.repeat((B + D - 1) // D, 1, 1, 1)[:B].contiguous()
t = {"thr": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"), "cln": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"),
     "lab": torch.empty((B, H, W), dtype=torch.int32, device="cuda"), "st": torch.zeros((B, 64, 5), dtype=torch.int32, device="cuda"),
     "ce": torch.zeros((B, 64, 2), dtype=torch.float64, device="cuda"), "nl": torch.zeros((B,), dtype=torch.int32, device="cuda")}
b = _vp.ChainBuffers(); b.bgr = big.data_ptr()
b.threshed, b.cleaned, b.labels, b.stats, b.centroids, b.nlabels = (t[k].data_ptr() for k in ("thr", "cln", "lab", "st", "ce", "nl"))
LO = int(os.environ.get("LO", "150"))            # 118: background speckle (hundreds of components, real unions in the strip-local pass)
MORPH = [] if os.environ.get("NOMORPH") else [(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)]
desc = _vp.make_chain_desc(W, H, _vp.BGR2LAB, (0, LO, 0), (255, 255, 255), MORPH, ccl=1, max_labels=64)
bad_total = 0
if os.environ.get("REFSMALL"):   # the same frames as a batch of D in a call of its own: the big batch must reproduce it
    ts = {"thr": torch.empty((D, H, W), dtype=torch.uint8, device="cuda"), "cln": torch.empty((D, H, W), dtype=torch.uint8, device="cuda"),
          "lab": torch.empty((D, H, W), dtype=torch.int32, device="cuda"), "st": torch.zeros((D, 64, 5), dtype=torch.int32, device="cuda"),
          "ce": torch.zeros((D, 64, 2), dtype=torch.float64, device="cuda"), "nl": torch.zeros((D,), dtype=torch.int32, device="cuda")}
    bs = _vp.ChainBuffers(); bs.bgr = small.data_ptr()
    bs.threshed, bs.cleaned, bs.labels, bs.stats, bs.centroids, bs.nlabels = (ts[k].data_ptr() for k in ("thr", "cln", "lab", "st", "ce", "nl"))
    ctx.chain_run(desc, bs, D); ctx.synchronize()
    ctx.chain_run(desc, b, B); ctx.synchronize()
    for k in ("thr", "cln", "lab", "st", "nl"):
        for f0 in range(0, B, D):
            cur = t[k][f0:f0 + D]
            eq = cur == ts[k][:cur.shape[0]]
            if not bool(eq.all()):
                for i in range(cur.shape[0]):
                    if not bool(eq[i].all()):
                        bad_total += 1
                        d = (~eq[i]).nonzero()
                        print(f"vs small batch, {k}: frame {f0 + i} differs in {d.shape[0]} entries, first at {d[0].tolist()}: got {cur[i][tuple(d[0].tolist())].item()} want {ts[k][i][tuple(d[0].tolist())].item()}", flush=True)
                        if bad_total > 30: sys.exit(1)
    print("vs small batch: mismatching frames:", bad_total)
for r in range(R):
    t["lab"].fill_(-7)
    torch.cuda.synchronize()
    ctx.chain_run(desc, b, B); ctx.synchronize()
    for k in ("lab", "st", "nl"):
        x = t[k]
        ref = x[:D]
        for f0 in range(D, B, D):
            cur = x[f0:f0 + D]
            eq = (cur == ref[:cur.shape[0]])
            if not bool(eq.all()):
                for i in range(cur.shape[0]):
                    if not bool(eq[i].all()):
                        bad_total += 1
                        d = (~eq[i]).nonzero()
                        msg = f"round {r} {k}: frame {f0 + i} differs from frame {i} in {d.shape[0]} entries"
                        if k == "lab":
                            ys, xs = d[:, 0], d[:, 1]
                            msg += f"; rows {int(ys.min())}..{int(ys.max())} cols {int(xs.min())}..{int(xs.max())}; got {cur[i][ys[0], xs[0]].item()} want {ref[i][ys[0], xs[0]].item()}"
                        print(msg, flush=True)
                        if bad_total > 40: sys.exit(1)
print("mismatching frames:", bad_total)
