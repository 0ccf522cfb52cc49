#!/bin/bash
# timing-only breakdown of the chain pass (exp library): each line = the pass with parts cut out, timed on its own
mkdir -p gpurun_out
for cfg in "PCV_CHAIN_DIAG=8" "PCV_CHAIN_DIAG=2" "PCV_CHAIN_DIAG=5" "PCV_CHAIN_DIAG=1" "PCV_CHAIN_DIAG=4" "PCV_CHAIN_DIAG=5 PCV_CHAIN_LDS=0" "PCV_CHAIN_DIAG=4 PCV_CHAIN_LDS=0" "PCV_CHAIN_DIAG=2 PCV_SPEC_BIN=1024" "PCV_CHAIN_DIAG=4 PCV_SPEC_BIN=1024"; do
  echo "== $cfg"
  env PCV_HIP_LIBRARY=exp $cfg timeout 120 python bench.py --steps 2 --warmup 1 --no-e2e --no-cpu-baseline --no-parity --no-kernel-events 2>&1 >/dev/null | grep PCV_CHAIN_DIAG | tail -4
done
