#!/bin/bash
# usage (on the GPU box): tools/prof_kernels.sh <tag> <python script> [args...]  -> per-kernel table from rocprofv3 --kernel-trace --stats
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag -o p -- python3 "$@" > $root/gpurun_out/$tag.log 2>&1
cd $root
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/$tag/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f'{r["Name"][:70]:70s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:10.1f} us {float(r["Percentage"]):6.2f}%')
PY
