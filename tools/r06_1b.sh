#!/bin/bash
# Round 6, full-size records on ONE GPU: (a) 1 B points unsharded, (b) the `--gpus N` command's code path at world size 1 on the
# config-3 cloud of 1 B points with its own N = 1 reference (n1_same_cloud: digest must be equal), (c) the config-3 rehearsal with
# 8 virtual ranks, both ownership modes, merged octree against the single-GPU build and the CPU oracle
mkdir -p gpurun_out
timeout 900 python bench.py --points 1000000000 --steps 5 --warmup 2 --no-legs --no-e2e --no-cpu-baseline --no-parity --digest --full-line > gpurun_out/r06_bench_1B_single_gpu.json 2> gpurun_out/r06_bench_1B_single_gpu.err; echo "1B rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_1B_single_gpu.json').read().strip().splitlines()[-1])
print('1B single:', d['value'], d['ms_per_step'], d.get('tree_digest'), d['config']['nodes'], d['kernel_ms_per_step'])
PY
timeout 1500 python bench.py --force-sharded --config3 --n1-same-cloud --steps 5 --warmup 2 > gpurun_out/r06_gpusN_path_world1_1B.json 2> gpurun_out/r06_gpusN_path_world1_1B.err; echo "gpusN path rc=$?"; tail -2 gpurun_out/r06_gpusN_path_world1_1B.err
tail -1 gpurun_out/r06_gpusN_path_world1_1B.json | head -c 3500; echo
timeout 2400 python bench.py --virtual-ranks 8 --shard-mode both --verify --full-line > gpurun_out/r06_parity_config3_virtual8_1B.json 2> gpurun_out/r06_parity_config3_virtual8_1B.err; echo "virtual8 rc=$?"; tail -2 gpurun_out/r06_parity_config3_virtual8_1B.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_parity_config3_virtual8_1B.json').read().strip().splitlines()[-1])
print({k:(v if len(str(v))<400 else str(v)[:400]) for k,v in d.items() if k in ('value','ms_per_step','parity','modes','single_gpu_build')})
PY
