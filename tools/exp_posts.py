"""Rates with posts on (the reference's default mode): module bodies outside the runtime and the red_buoy / bins modules on the runtime.
usage: python tools/exp_posts.py [calls]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "shims")):
    sys.path.insert(0, p)
import module_harness as MH

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
out = {"dma_posts": os.environ.get("VP_DMA_POSTS", "1")}
out["process_body_red_buoy"] = MH.body_rates("buoy", calls=calls)
out["process_body_bins"] = MH.body_rates("bins", calls=calls)
out["runtime_e2e_red_buoy_posts_off"] = MH.runtime_rate("buoy", seconds=3.0)
out["runtime_e2e_red_buoy_posts_on"] = MH.runtime_rate("buoy", seconds=3.0, flags=())
out["runtime_e2e_bins_posts_on"] = MH.runtime_rate("bins", seconds=2.0, flags=())
out["runtime_e2e_gate_posts_on"] = MH.runtime_rate("gate", seconds=2.0, flags=())
print(json.dumps(out, indent=1))
