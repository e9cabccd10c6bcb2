#!/bin/bash
# Run on the GPU box from the repo root:  tools/collect_noise_evidence.sh <tag>     (e.g. r04/a)
# Crowded-frame labelling (tools/exp_noise.py: 128 frames of 1080p noise, grey >= lo -> CCL with labels and statistics) at 2 % / 10 % /
# 50 % density: plain timings, then HBM-side bytes per launch of every kernel from separate rocprofv3 --pmc passes (FETCH_SIZE,
# WRITE_SIZE; MI355X_MICROARCH.md's corrections are applied by tools/pmc_traffic.py).
set -e
tag=${1:-r04/x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/evidence
mkdir -p $out/$(dirname $tag)
cd $root
: > $out/${tag}_noise.txt
for lo in 230 190 128; do python3 tools/exp_noise.py $lo 2>/dev/null | tail -1 >> $out/${tag}_noise.txt; done
cat $out/${tag}_noise.txt
cd /tmp && export TMPDIR=/tmp
for lo in 190 128; do
  K=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/nf$lo -o pf -- python3 $root/tools/exp_noise.py $lo > /dev/null 2>&1
  K=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/nw$lo -o pw -- python3 $root/tools/exp_noise.py $lo > /dev/null 2>&1
  python3 $root/tools/pmc_traffic.py $(find $out/nf$lo -name '*counter_collection.csv' | head -1) $(find $out/nw$lo -name '*counter_collection.csv' | head -1) \
      $out/${tag}_noise_traffic_lo$lo.json rocprofv3 --pmc "FETCH_SIZE|WRITE_SIZE" -- python3 tools/exp_noise.py $lo > /dev/null
  rm -rf $out/nf$lo $out/nw$lo
done
echo done
