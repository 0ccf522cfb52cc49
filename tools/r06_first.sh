#!/bin/bash
# round 6, first GPU call: GPU suite, the default line (is the last line < 4 KB and complete?), the --gpus N code path at world 1
# on a config-3 cloud with its own N = 1 reference
bash tools/r06_run.sh r06a tests bench
timeout 600 python bench.py --force-sharded --config3 --points 100000000 --n1-same-cloud --steps 10 --warmup 3 > gpurun_out/r06a_sharded_w1.json 2> gpurun_out/r06a_sharded_w1.err; echo "sharded rc=$?"; tail -3 gpurun_out/r06a_sharded_w1.err
tail -1 gpurun_out/r06a_sharded_w1.json | head -c 3000
