"""The contract of bench.py's output as the driver reads it: stdout is ONE line, a JSON object with the keys the round's measurement rules
name (metric, value, roofline, cpu_baseline, ...), whatever the side measurements print (the modules they run announce themselves)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_stdout_is_one_json_line_with_the_contract_keys():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--regions", "1", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:5]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "u8" and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0 and c["sample"]
    for key in ("process_body_red_buoy", "process_body_bins", "runtime_e2e_red_buoy", "runtime_e2e_bins", "host_fed_1080p", "host_fed_4k"):
        assert key in d["extras"], (key, d["extras"].get("runtime_rates_error"))
    assert d["value"] > 0 and abs(d["value"] - 128 * 1e3 / d["ms_per_step"]) / d["value"] < 0.01
