"""Phase costs inside k_ccl2_local / k_ccl2_merge from a probe build (tools/build_probe.sh; run with VP_LIB=.../libvp_probe.so):
average counter ticks per block between probe points (clock64), S1 chain at batch 128."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import frames as F
from vision import _vp
W, H, B = 1920, 1080, 128
ctx = _vp.Context(0)
L = _vp.lib()
L.vp_debug_probe.argtypes = [C.c_void_p]
d = torch.from_numpy(np.stack([F.s1_buoy(i % 8, W, H) for i in range(B)])).cuda()
t = {"thr": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"), "cln": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"),
     "lab": torch.empty((B, H, W), dtype=torch.int32, device="cuda"), "st": torch.zeros((B, 256, 5), dtype=torch.int32, device="cuda"),
     "ce": torch.zeros((B, 256, 2), dtype=torch.float64, device="cuda"), "nl": torch.zeros((B,), dtype=torch.int32, device="cuda")}
b = _vp.ChainBuffers(); b.bgr = d.data_ptr()
b.threshed, b.cleaned, b.labels, b.stats, b.centroids, b.nlabels = (t[k].data_ptr() for k in ("thr", "cln", "lab", "st", "ce", "nl"))
lo = tuple(int(x) for x in os.environ.get("LO", "0,150,0").split(","))
desc = _vp.make_chain_desc(W, H, _vp.BGR2LAB, lo, (255, 255, 255), [(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)], ccl=1, max_labels=256)
out = np.zeros(32)
for _ in range(3): ctx.chain_run(desc, b, B)
ctx.synchronize(); L.vp_debug_probe(out.ctypes.data)
K = 10
ctx.profile_begin(K * 24)
for _ in range(K): ctx.chain_run(desc, b, B)
pr = ctx.profile_end()
L.vp_debug_probe(out.ctypes.data)
print({k: round(1e3 * v[0] / v[1], 1) for k, v in pr.items()})
names = [["staged + bg box", "count + scan", "indices + row leaders", "unions", "roots + list indices", "second walk", "list written"],
         ["sizes scanned", "records in LDS", "boundary unions", "roots + keys", "ranks", "stats moved + table", "rows", "zero rows"]]
for k, nm in enumerate(names):
    blocks = max(out[16 * k + 15], 1)
    tot = out[16 * k:16 * k + 8].sum()
    print(["k_ccl2_local", "k_ccl2_merge"][k], f"ticks per block, over the {int(blocks)} blocks of the last launch that ran to the end: total {tot / blocks:.0f}")
    for i, n in enumerate(nm):
        print(f"   {n:28s} {out[16 * k + i] / blocks:9.1f}  {100 * out[16 * k + i] / max(tot, 1):5.1f} %")
