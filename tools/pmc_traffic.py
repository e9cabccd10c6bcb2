"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv output into per-kernel HBM-side bytes per launch.

Correction per MI355X_MICROARCH.md (HBM / rocprofv3 section): counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of wide coalesced streaming reads -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.

Kernels are keyed by their FULL name including template arguments ("k_color_thresh_flat<0, 2, true, true>"), so two
instantiations of one template (e.g. the statistics-only variant without the mask store) are never averaged together.
The output carries `_meta`: the digest of the kernel sources the counters were collected from and the profiled command;
bench.py reports a traffic figure only when that digest equals the build it is running.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [profiled command ...]
"""
import collections
import csv
import importlib.util
import json
import os
import sys


def full_name(kernel_name):
    """'void k<0, 2, true>(unsigned char const*, ...)' -> 'k<0, 2, true>' (template arguments kept, parameter list dropped)."""
    s = kernel_name.strip()
    if s.startswith("void "):
        s = s[5:]
    depth = 0
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return s[:i].strip()
    return s


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[full_name(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def source_digest():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("vp_build", os.path.join(root, "cuauv-vision-pipeline_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_digest()


def main():
    fetch_csv, write_csv, out = sys.argv[1], sys.argv[2], sys.argv[3]
    f = load(fetch_csv, "FETCH_SIZE")
    w = load(write_csv, "WRITE_SIZE")
    res = {"_meta": {"csrc_sha256": source_digest(), "command": " ".join(sys.argv[4:]) or None,
                     "correction": "bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE); separate --pmc passes per counter",
                     "keyed_by": "full kernel name incl. template arguments"}}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        fv, fn = f.get(k, (0.0, 0))
        wv, wn = w.get(k, (0.0, 0))
        fb = 2.0 * 1024.0 * fv
        wb = 1024.0 * wv
        res[k] = {"fetch_bytes_corrected": int(fb), "write_bytes": int(wb), "traffic_bytes_per_launch": int(fb + wb),
                  "raw_FETCH_SIZE_KiB": fv, "raw_WRITE_SIZE_KiB": wv, "launches_seen": [fn, wn]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
