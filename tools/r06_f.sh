#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_sharded.py -m gpu -q --tb=short > gpurun_out/r06f_new.log 2>&1; echo "sharded tests rc=$?"; tail -3 gpurun_out/r06f_new.log
timeout 600 python bench.py --force-sharded --config3 --points 100000000 --no-n1 --steps 10 --warmup 3 > gpurun_out/r06f_sharded_w1.json 2> gpurun_out/r06f_sharded_w1.err; echo "sharded rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06f_sharded_w1.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d.get('sharded_stage_ms'), d['kernel_ms_per_step'])
PY
STEPS=6 TAG=r06f_ab RUNS="base:PCV_CHAIN_DIAG=0 nocolor:PCV_CHAIN_DIAG=1 nostore:PCV_CHAIN_DIAG=4 nocolor_nostore:PCV_CHAIN_DIAG=5 base2:PCV_CHAIN_DIAG=0 nowalk:PCV_CHAIN_DIAG=2 nowalk_all:PCV_CHAIN_DIAG=7" bash tools/ab_quick.sh
