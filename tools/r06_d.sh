#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sharded.py "tests/test_gpu_single_chain.py::test_wide_rank_geometries_against_the_oracle" -m gpu -q --tb=short > gpurun_out/r06d_new.log 2>&1; echo "new rc=$?"; tail -30 gpurun_out/r06d_new.log
bash tools/profile_query.sh r06d > gpurun_out/r06d_qprof.log 2>&1; tail -5 gpurun_out/r06d_qprof.log
head -30 gpurun_out/r06d_query_kernel_stats.csv
timeout 600 python bench.py --force-sharded --config3 --points 100000000 --no-n1 --steps 10 --warmup 3 > gpurun_out/r06d_sharded_w1.json 2> gpurun_out/r06d_sharded_w1.err; echo "sharded rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06d_sharded_w1.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d.get('sharded_stage_ms'), d['kernel_ms_per_step'])
PY
