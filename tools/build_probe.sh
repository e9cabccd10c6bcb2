#!/bin/bash
# Measurement build of libvp with phase probes in the two-level labelling kernels (-DVP_PROBE): lib/libvp_probe.so, loaded with
# VP_LIB=<path>.  Not part of the product build.
set -e
root=$(cd $(dirname $0)/.. && pwd)
c=$root/cuauv-vision-pipeline_amd/csrc
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -DVP_PROBE -Wno-unused-value -Wno-unused-result -x hip \
  $c/vp_api.hip $c/vp_color.hip $c/vp_morph.hip $c/vp_ccl.hip $c/vp_balance.hip $c/vp_yolo.hip $c/vp_filter.hip $c/vp_feed.hip $c/vp_post.hip $c/vp_tables.cpp \
  -o $root/cuauv-vision-pipeline_amd/lib/libvp_probe.so
