#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_query.py tests/test_gpu_query_fuzz.py tests/test_gpu_sharded.py "tests/test_gpu_single_chain.py::test_wide_rank_geometries_against_the_oracle" -m gpu -q --tb=short > gpurun_out/r06c_new.log 2>&1; echo "new rc=$?"; tail -30 gpurun_out/r06c_new.log
bash tools/r06_run.sh r06c bench
