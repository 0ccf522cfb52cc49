#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_query.py tests/test_gpu_query_fuzz.py tests/test_gpu_disk.py -m gpu -q --tb=short > gpurun_out/r06e_new.log 2>&1; echo "new rc=$?"; tail -12 gpurun_out/r06e_new.log
bash tools/profile_query.sh r06e > gpurun_out/r06e_qprof.log 2>&1; tail -3 gpurun_out/r06e_qprof.log
grep -E "cull_nodes|visible|query_flags" gpurun_out/r06e_query_kernel_stats.csv
PCV_HIP_LIBRARY=exp PCV_CULL_DEBUG=1 timeout 300 python bench.py --query --verify-frusta 0 --verify-cull-frusta 0 --query-steps 2 --warmup 1 2>&1 >/dev/null | grep "pcv cull" | sort | uniq -c
