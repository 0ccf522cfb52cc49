#!/bin/bash
# sort tuning sweep on the GPU box: rebuild pcv_sort.hip with each flag set and time K3 alone (tools/sort_bench.py)
cd $GRAFT_REPO_ROOT
for cfg in "${@}"; do
  touch point_cloud_viewer_amd/csrc/pcv_sort.hip
  make -C point_cloud_viewer_amd/csrc EXTRA="$cfg" 2>&1 | grep -E "error|spill"
  echo "== $cfg"
  timeout 120 python tools/sort_bench.py 2>&1 | grep -E "path keys|uniform|Error"
done
