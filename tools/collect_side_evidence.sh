#!/bin/bash
# Run on the GPU box from the repo root after tools/collect_evidence.sh:  tools/collect_side_evidence.sh <tag>   (e.g. r04/a)
# The figures of DESIGN.md section 5 that are not in the bench line: other workloads, crowded frames, modules on the runtime, the
# dispatcher against the link, latency, and the N-rank code path rehearsed on one GPU.
tag=${1:-r04/x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/evidence
mkdir -p $out/$(dirname $tag)
cd $root
python3 tools/exp_cases.py $out/${tag}_cases.json > $out/${tag}_cases.txt 2> $out/${tag}_cases.err
for lo in 230 190 128; do python3 tools/exp_noise.py $lo; done > $out/${tag}_noise.txt 2> /dev/null
for lo in 230 190 128; do VP_C3_IDS=8192 python3 tools/exp_noise.py $lo; done > $out/${tag}_noise_8rows.txt 2> /dev/null
python3 tools/exp_process.py 400 > $out/${tag}_process.txt 2>&1
# posts on (the reference's default): module bodies and modules on the runtime, by DMA and by the download + write_frame path of round 3
python3 tools/exp_posts.py 2> /dev/null | python3 -c "import sys; t = sys.stdin.read(); print(t[t.rindex(chr(10) + '{' + chr(10)) + 1:])" > $out/${tag}_posts_dma.json
VP_DMA_POSTS=0 python3 tools/exp_posts.py 2> /dev/null | python3 -c "import sys; t = sys.stdin.read(); print(t[t.rindex(chr(10) + '{' + chr(10)) + 1:])" > $out/${tag}_posts_host_copy.json
python3 tools/exp_postcost.py 2>&1 | grep -v "^\[\|amdgpu.ids" > $out/${tag}_postcost.txt
python3 tools/exp_contours_single.py 200 2> /dev/null | tail -1 > $out/${tag}_contours_single.json
python3 tools/exp_opkernels.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_opkernels.txt
python3 tools/exp_bits_cache.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_bits_cache.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $out/kt -o kt -- python3 $root/tools/exp_bits_cache.py > /dev/null 2>&1)
python3 tools/kernel_medians.py $(find $out/kt -name '*kernel_trace.csv' | head -1) > $out/${tag}_contour_kernel_medians.txt 2>&1; rm -rf $out/kt
python3 tools/exp_hostfed.py 1080p > $out/${tag}_hostfed.txt 2>&1
python3 tools/exp_hostfed.py 4k >> $out/${tag}_hostfed.txt 2>&1
for m in buoy bins gate; do python3 tools/exp_runtime.py 4 $m 2>&1 | tail -1; done > $out/${tag}_runtime.txt
VP_FEEDER=0 python3 tools/exp_runtime.py 4 buoy 2>&1 | tail -1 >> $out/${tag}_runtime.txt
VP_DEVICE_FRAMES=0 python3 tools/exp_runtime.py 4 buoy 0.0002 2>&1 | tail -1 >> $out/${tag}_runtime.txt
python3 tools/exp_configs.py > $out/${tag}_configs.txt 2>&1
python3 tools/exp_latency.py > $out/${tag}_latency.txt 2>&1
python3 -m pytest tests/test_gpu_yolo_module.py -q -s -k rate > $out/${tag}_yolo.txt 2>&1
# the N-rank code path of bench.py on one GPU (two ranks on cuda:0, gloo for the barrier and the MAX), with the host-fed leg every rank
# runs for N > 1: not a scaling number
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --rehearse-on-one-gpu \
    --no-cpu-baseline --steps 20 2> $out/${tag}_rehearse.err | grep '^{"metric"' > $out/${tag}_rehearse_2ranks_one_gpu.json
tail -3 $out/${tag}_runtime.txt
