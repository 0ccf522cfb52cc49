#!/bin/bash
# timing-only breakdown of the chain pass (exp library): each line = the pass with parts cut out (PCV_CHAIN_DIAG: 1 no walk,
# 2 stop after the deal, 4 no record stores, 8 stop after the coordinate loads), timed on its own next to the full pass.
# CONFIGS="ENV=VAL,ENV2=VAL2 ..." (comma-separated inside one configuration)
mkdir -p gpurun_out
for cfg in ${CONFIGS:-PCV_CHAIN_DIAG=1}; do
  echo "== $cfg"
  env PCV_HIP_LIBRARY=exp PCV_CHAIN_V=4 ${cfg//,/ } timeout 120 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-parity --no-kernel-events 2>&1 >/dev/null | grep PCV_CHAIN_DIAG | tail -4 | tr '\n' ' '; echo
done
