#!/bin/bash
# Measurement build of libvp with phase stamps in the one-block contour bookkeeping (k_ct_jump, LDS form; -DVP_CT_PROBE, optional
# -DCTJ_HOPS=n): lib/libvp_ctprobe.so, loaded with VP_LIB=<path> by tools/exp_ct_probe.py.  Not part of the product build.
set -e
root=$(cd $(dirname $0)/.. && pwd)
c=$root/cuauv-vision-pipeline_amd/csrc
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -DVP_CT_PROBE $VP_CT_PROBE_DEFS -Wno-unused-value -Wno-unused-result -x hip \
  $c/vp_api.hip $c/vp_color.hip $c/vp_morph.hip $c/vp_ccl.hip $c/vp_balance.hip $c/vp_yolo.hip $c/vp_filter.hip $c/vp_feed.hip $c/vp_post.hip $c/vp_tables.cpp \
  -o $root/cuauv-vision-pipeline_amd/lib/libvp_ctprobe.so
