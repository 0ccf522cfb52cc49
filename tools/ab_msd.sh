#!/bin/bash
# A/B of the record sort's digit order (PCV_SORT_MSD, experiment library): alternating processes, several pool placements each
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export PCV_HIP_LIBRARY=exp
timeout 300 python -m pytest tests/test_gpu_single_chain.py -x -q -m gpu -k "alternative_kernels and MSD" 2>&1 | tail -3
for rep in 1 2; do for m in 0 1; do
  echo "== PCV_SORT_MSD=$m rep $rep"
  PCV_SORT_MSD=$m timeout 300 python tools/placement_probe.py 100000000 4 2>&1 | tail -4
done; done
