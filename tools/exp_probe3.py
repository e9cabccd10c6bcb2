"""Phase costs inside k_ccl3_link / k_ccl3_label from a probe build (tools/build_probe.sh; VP_LIB=.../libvp_probe.so): microseconds per
block summed over its items, noise at the density given.  usage: exp_probe3.py [lo]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import frames as F
from vision import _vp
W, H, B, ML = 1920, 1080, 128, 65536
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 190
ctx = _vp.Context(0)
L = _vp.lib()
L.vp_debug_probe3.argtypes = [C.c_void_p]
d = torch.from_numpy(np.stack([F.s3_noise(i % 8, W, H) for i in range(B)])).cuda()
t = {"thr": torch.empty((B, H, W), dtype=torch.uint8, device="cuda"), "lab": torch.empty((B, H, W), dtype=torch.int32, device="cuda"),
     "st": torch.zeros((B, ML, 5), dtype=torch.int32, device="cuda"), "ce": torch.zeros((B, ML, 2), dtype=torch.float64, device="cuda"),
     "nl": torch.zeros((B,), dtype=torch.int32, device="cuda")}
b = _vp.ChainBuffers(); b.bgr = d.data_ptr(); b.threshed = t["thr"].data_ptr(); b.labels = t["lab"].data_ptr()
b.stats, b.centroids, b.nlabels = t["st"].data_ptr(), t["ce"].data_ptr(), t["nl"].data_ptr()
desc = _vp.make_chain_desc(W, H, _vp.BGR2GRAY, (lo, 0, 0), (255, 255, 255), [], ccl=1, max_labels=ML)
out = np.zeros(48)
for _ in range(2): ctx.chain_run(desc, b, B)
ctx.synchronize(); L.vp_debug_probe3(out.ctypes.data)
ctx.profile_begin(24)
ctx.chain_run(desc, b, B)
pr = ctx.profile_end()
L.vp_debug_probe3(out.ctypes.data)
print({k: round(1e3 * v[0] / v[1], 1) for k, v in pr.items()})
names = [["bits staged", "parents set", "contacts", "flatten + roots", "dump issued", "acc cleared"],
         ["", "", "", "", "", "", "boundary rows staged", "boundary unions"],
         ["bases + staged", "local roots ranked", "root labels", "accumulate", "emit rows / table", "table flushed", "label stores", "totals", "staging: loads to LDS", "item set-up + wait for the block"]]
for k, nm in enumerate(names):
    blocks = max(out[16 * k + 15], 1)
    tot = out[16 * k:16 * k + 15].sum()
    print(["k_ccl3_link", "k_ccl3_bound", "k_ccl3_label"][k], f"us per block over {int(blocks)} blocks: total {tot / blocks / 100:.1f}")
    for i, n in enumerate(nm):
        if not n: continue
        print(f"   {n:24s} {out[16 * k + i] / blocks / 100:9.1f}  {100 * out[16 * k + i] / max(tot, 1):5.1f} %")
