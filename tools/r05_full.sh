set -x
bash tools/r05_run.sh r05f tests smoke
bash tools/profile_bench.sh r05f_prof > gpurun_out/r05f_prof.log 2>&1; tail -3 gpurun_out/r05f_prof.log
cp gpurun_out/r05f_prof_bench_traffic.json profiles/r05_bench_100M_traffic.json
cp gpurun_out/r05f_prof_bench_valu.json profiles/r05_bench_100M_valu.json
bash tools/profile_query.sh r05f > gpurun_out/r05f_qprof.log 2>&1; tail -3 gpurun_out/r05f_qprof.log
cp gpurun_out/r05f_query_counters.json profiles/r05_query_counters.json
bash tools/step_timeline.sh r05f --no-legs --no-parity > /dev/null 2>&1; head -60 gpurun_out/r05f_timeline.txt
bash tools/r05_run.sh r05f bench
