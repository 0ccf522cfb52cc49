#!/bin/bash
for spin in 0 300 0 300 100; do echo "== PCV_POOL_SPIN_US=$spin"; PCV_HIP_LIBRARY=exp PCV_POOL_SPIN_US=$spin python tools/ingest_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "batch 500000|batch 1000000|one-shot"; done
