mkdir -p gpurun_out/unpack
for v in 0 8 1 3; do
  VP_MORPH_DBG=$v python bench.py --steps 20 --regions 3 --no-extras --no-cpu-baseline > gpurun_out/unpack/dbg_$v.json 2> gpurun_out/unpack/dbg_$v.err || exit 1
  python - <<PY
import json
d = json.load(open("gpurun_out/unpack/dbg_$v.json"))
print("DBG=$v", d["ms_per_step"], {k: round(x["avg_us"], 1) for k, x in d["kernels"].items() if "morph" in k})
PY
done
