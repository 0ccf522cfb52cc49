#!/bin/bash
# After tools/r06_full.sh (one gpurun call): the records it left in gpurun_out/ go to profiles/ under their committed names.
set -e
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
cp $G/r06f_bench_default.json $P/r06_bench_100M.json
cp $G/r06f_bench_default_detail.json $P/r06_bench_100M_detail.json
cp $G/r06f_prof_f64.csv $P/r06_bench_100M_f64_counters.csv
cp $G/r06f_prof_kernel_stats.csv $P/r06_bench_100M_kernel_stats.csv
cp $G/r06f_prof_kernel_stats_traffic.json $P/r06_bench_100M_kernel_stats_traffic.json
cp $G/r06f_prof_kernel_stats_sq.csv $P/r06_bench_100M_sq_counters.csv
cp $G/r06f_prof_bench_traffic.json $P/r06_bench_100M_traffic.json
cp $G/r06f_prof_bench_valu.json $P/r06_bench_100M_valu.json
cp $G/r06f_prof_isa_mix.json $P/r06_bench_100M_isa_mix.json
cp $G/r06f_gputest.log $P/r06_gputest.log
cp $G/r06f_smoke.log $P/r06_smoke.log
cp $G/r06f_query_counters.json $P/r06_query_counters.json
cp $G/r06f_query_kernel_stats.csv $P/r06_query_kernel_stats.csv
cp $G/r06f_query_kernel_stats_sq.csv $P/r06_query_sq_counters.csv
cp $G/r06f_timeline.txt $P/r06_step_timeline.txt
python - <<'PY'
import json
d = json.loads(open('profiles/r06_bench_100M.json').read().strip().splitlines()[-1])
t = json.load(open('profiles/r06_bench_100M_traffic.json'))
print('line', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'matches', d['roofline'].get('profile_matches_build'), 'hash', t.get('build_hash'))
PY
