#!/bin/bash
# round 6, second GPU call: the new tests first (ingest, wide rank geometries), then the whole suite, then the default line
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_single_chain.py -m gpu -q --tb=short -x > gpurun_out/r06b_new.log 2>&1; echo "new rc=$?"; tail -30 gpurun_out/r06b_new.log
bash tools/r06_run.sh r06b tests bench
