#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_single_chain.py tests/test_gpu_build.py tests/test_gpu_fuzz.py tests/test_gpu_sharded.py -m gpu -q --tb=short -x > gpurun_out/r06g_new.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r06g_new.log
STEPS=8 TAG=r06g_ab RUNS="late:PCV_COLOR_LATE=1 early:PCV_COLOR_LATE=0 late2:PCV_COLOR_LATE=1 early2:PCV_COLOR_LATE=0" bash tools/ab_quick.sh
timeout 600 python bench.py --force-sharded --config3 --points 100000000 --no-n1 --steps 10 --warmup 3 > gpurun_out/r06g_sharded_w1.json 2> gpurun_out/r06g_sharded_w1.err; echo "sharded rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06g_sharded_w1.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d.get('sharded_stage_ms'), d['kernel_ms_per_step'])
PY
