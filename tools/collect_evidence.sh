#!/bin/bash
# Run on the GPU box from the repo root:  tools/collect_evidence.sh <tag>     (e.g. r04/a)
# Produces gpurun_out/evidence/<tag>_{bench.json,kernel_stats.csv,traffic.json}; copy them into profiles/ afterwards.
# rocprofv3 passes are separate, as MI355X_MICROARCH.md prescribes: one --kernel-trace --stats pass, one --pmc pass per counter.
set -e
tag=${1:-r02/x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/evidence
mkdir -p $out/$(dirname $tag)
cd $root
python3 bench.py --steps 30 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
# the headline kernel table comes from the timed chain ALONE (--no-extras --no-cpu-baseline): the extras launch the same kernels from
# the contour code, on two streams and on other inputs, and their durations must not be averaged into the chain's
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python3 $root/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1
cp $(find $out/ks -name '*kernel_stats.csv' | head -1) $out/${tag}_kernel_stats.csv
# (and the same with the extras, under its own name: the kernels of the module bodies, the contour code and the crowded-frame path)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kx -o kx -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find $out/kx -name '*kernel_stats.csv' | head -1) $out/${tag}_kernel_stats_with_extras.csv
# counter passes see the timed chain only (--no-extras: no second instantiation of any kernel, no contour code reusing the labelling kernels)
pmc_cmd="bench.py --steps 3 --warmup 1 --regions 1 --no-cpu-baseline --no-extras"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf -o pf -- python3 $root/$pmc_cmd > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw -o pw -- python3 $root/$pmc_cmd > /dev/null 2>&1
cd $root
python3 tools/pmc_traffic.py $(find $out/pf -name '*counter_collection.csv' | head -1) $(find $out/pw -name '*counter_collection.csv' | head -1) $out/${tag}_traffic.json rocprofv3 --pmc "FETCH_SIZE|WRITE_SIZE" -- python3 $pmc_cmd > /dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kb -o kb -- python3 $root/tools/exp_balance.py > $out/${tag}_balance.txt 2>&1
cp $(find $out/kb -name '*kernel_stats.csv' | head -1) $out/${tag}_balance_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kc -o kc -- python3 $root/tools/exp_contours.py > $out/${tag}_contours.txt 2>&1
cp $(find $out/kc -name '*kernel_stats.csv' | head -1) $out/${tag}_contours_kernel_stats.csv
cd $root
rm -rf $out/ks $out/kx $out/pf $out/pw $out/kb $out/kc
cat $out/${tag}_bench.json
