#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_single_chain.py tests/test_gpu_build.py tests/test_gpu_fuzz.py -m gpu -q --tb=short -x > gpurun_out/r06h_new.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r06h_new.log
STEPS=8 TAG=r06h_ab RUNS="aligned:PCV_CHAIN_DIAG=0 r05loads:PCV_CHAIN_DIAG=8 aligned2:PCV_CHAIN_DIAG=0 r05loads2:PCV_CHAIN_DIAG=8 nocolor:PCV_CHAIN_DIAG=1" bash tools/ab_quick.sh
