#!/bin/bash
# A/B measurement builds: libvp from another git revision of csrc/ beside the working tree's, loaded with VP_LIB=<path>.
#   tools/build_ab.sh <git-ref> <tag>   ->  cuauv-vision-pipeline_amd/lib/libvp_<tag>.so
# Not part of the product build.
set -e
ref=${1:-HEAD}; tag=${2:-prev}
root=$(cd $(dirname $0)/.. && pwd)
tmp=$(mktemp -d)
git -C $root archive $ref cuauv-vision-pipeline_amd/csrc include | tar -x -C $tmp
c=$tmp/cuauv-vision-pipeline_amd/csrc
srcs=$(ls $c/vp_*.hip $c/vp_tables.cpp)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -Wno-unused-value -Wno-unused-result -x hip $srcs \
  -o $root/cuauv-vision-pipeline_amd/lib/libvp_$tag.so
rm -rf $tmp
echo $root/cuauv-vision-pipeline_amd/lib/libvp_$tag.so
