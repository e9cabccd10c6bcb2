#!/bin/bash
# Run on the GPU box from the repo root after tools/collect_evidence.sh:  tools/collect_side_evidence.sh <tag>   (e.g. r02/c)
# The figures of DESIGN.md section 5 that are not in the bench line: other workloads, the per-operator API, the dispatcher, latency.
tag=${1:-r02/x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/evidence
mkdir -p $out/$(dirname $tag)
cd $root
python3 tools/exp_cases.py $out/${tag}_cases.json > $out/${tag}_cases.txt 2> $out/${tag}_cases.err
python3 tools/exp_process.py 400 > $out/${tag}_process.txt 2>&1
python3 tools/exp_dispatch.py > $out/${tag}_dispatch.txt 2>&1
python3 tools/exp_runtime.py > $out/${tag}_runtime.txt 2>&1
VP_PRIVATE_READS=0 python3 tools/exp_runtime.py >> $out/${tag}_runtime.txt 2>&1
python3 tools/exp_configs.py > $out/${tag}_configs.txt 2>&1
python3 tools/exp_latency.py > $out/${tag}_latency.txt 2>&1
python3 tools/exp_hostmem.py > $out/${tag}_hostmem.txt 2>&1
python3 -m pytest tests/test_gpu_yolo_module.py -q -s -k rate > $out/${tag}_yolo.txt 2>&1
tail -3 $out/${tag}_process.txt
