#!/bin/bash
# Round 6, the records of the final build in ONE gpurun call: GPU suite, smoke, rocprofv3 passes of the build and of the query
# path (-> the JSON / CSV files bench.py and DESIGN.md quote), the kernel timeline of one step, the default line.
set -x
bash tools/r06_run.sh r06f tests smoke
bash tools/profile_bench.sh r06f_prof > gpurun_out/r06f_prof.log 2>&1; tail -3 gpurun_out/r06f_prof.log
cp gpurun_out/r06f_prof_bench_traffic.json profiles/r06_bench_100M_traffic.json   # (on the GPU box: what the bench leg below quotes;
cp gpurun_out/r06f_prof_bench_valu.json profiles/r06_bench_100M_valu.json         #  copy the same files into profiles/ at home afterwards)
bash tools/profile_query.sh r06f > gpurun_out/r06f_qprof.log 2>&1; tail -3 gpurun_out/r06f_qprof.log
cp gpurun_out/r06f_query_counters.json profiles/r06_query_counters.json
bash tools/step_timeline.sh r06f --no-legs --no-parity > /dev/null 2>&1; head -60 gpurun_out/r06f_timeline.txt
bash tools/r06_run.sh r06f bench
