#!/bin/bash
# Experiment builds of libpcv_hip.so that differ in ONE compile-time knob of ONE translation unit, for A/B runs on the
# GPU box in a single gpurun call: PCV_HIP_LIBRARY=point_cloud_viewer_amd/libpcv_hip_<tag>.so python bench.py ...
# usage: tools/build_variants.sh <tag> <file.hip> "<-DKNOB=VALUE ...>"   (run after `make` in csrc)
set -e
TAG=$1; SRC=$2; DEFS=$3
cd "$(dirname "$0")/../point_cloud_viewer_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-result"
OBJ=/tmp/pcv_variant_${TAG}.o
/opt/rocm/bin/hipcc $FLAGS $DEFS -c $SRC -o $OBJ
OTHERS=$(ls *.o | grep -v "^exp_" | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpcv_hip_${TAG}.so $OTHERS $OBJ -lpthread
ls -la ../libpcv_hip_${TAG}.so
