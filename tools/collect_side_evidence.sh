#!/bin/bash
# Run on the GPU box from the repo root after tools/collect_evidence.sh:  tools/collect_side_evidence.sh <tag>   (e.g. r03/a)
# The figures of DESIGN.md section 5 that are not in the bench line: other workloads, crowded frames, modules on the runtime, the
# dispatcher against the link, latency, and the N-rank code path rehearsed on one GPU.
tag=${1:-r03/x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/evidence
mkdir -p $out/$(dirname $tag)
cd $root
python3 tools/exp_cases.py $out/${tag}_cases.json > $out/${tag}_cases.txt 2> $out/${tag}_cases.err
for lo in 230 190 128; do python3 tools/exp_noise.py $lo; done > $out/${tag}_noise.txt 2> /dev/null
for lo in 230 190 128; do VP_C3_IDS=8192 python3 tools/exp_noise.py $lo; done > $out/${tag}_noise_8rows.txt 2> /dev/null
python3 tools/exp_process.py 400 > $out/${tag}_process.txt 2>&1
python3 tools/exp_hostfed.py 1080p > $out/${tag}_hostfed.txt 2>&1
python3 tools/exp_hostfed.py 4k >> $out/${tag}_hostfed.txt 2>&1
for m in buoy bins gate; do python3 tools/exp_runtime.py 4 $m 2>&1 | tail -1; done > $out/${tag}_runtime.txt
VP_FEEDER=0 python3 tools/exp_runtime.py 4 buoy 2>&1 | tail -1 >> $out/${tag}_runtime.txt
VP_DEVICE_FRAMES=0 python3 tools/exp_runtime.py 4 buoy 0.0002 2>&1 | tail -1 >> $out/${tag}_runtime.txt
python3 tools/exp_configs.py > $out/${tag}_configs.txt 2>&1
python3 tools/exp_latency.py > $out/${tag}_latency.txt 2>&1
python3 -m pytest tests/test_gpu_yolo_module.py -q -s -k rate > $out/${tag}_yolo.txt 2>&1
# the N-rank code path of bench.py on one GPU (two ranks on cuda:0, gloo for the barrier and the MAX): not a scaling number
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --rehearse-on-one-gpu \
    --no-extras --no-cpu-baseline --steps 20 > $out/${tag}_rehearse_2ranks_one_gpu.json 2> $out/${tag}_rehearse.err
tail -3 $out/${tag}_runtime.txt
