#!/bin/bash
# rocprofv3 passes over `bench.py --query` (BASELINE config 4) on the GPU box, each in its OWN run (counters never share a
# run with --stats): kernel durations, FETCH_SIZE, WRITE_SIZE, SQ counters of the query kernels.
# usage: gpurun -- 'bash tools/profile_query.sh TAG'  ->  gpurun_out/TAG_query_kernel_stats.csv, _sq.csv, _traffic.json
TAG=${1:-qprof}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
source $GRAFT_REPO_ROOT/tools/run_limited.sh
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --query --verify-frusta 0 --verify-cull-frusta 0 --steps 3 --warmup 1"
run_limited 200 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_qs -o r -- $B "$@" > $OUT/${TAG}_query_stats.log 2>&1
run_limited 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/${TAG}_qf -o r -- $B "$@" > /dev/null 2>&1
run_limited 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/${TAG}_qw -o r -- $B "$@" > /dev/null 2>&1
run_limited 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d /tmp/${TAG}_qq -o r -- $B "$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py --stats /tmp/${TAG}_qs/r_results.db --fetch /tmp/${TAG}_qf/r_results.db --write /tmp/${TAG}_qw/r_results.db --sq /tmp/${TAG}_qq/r_results.db \
  --command "python bench.py --query --verify-frusta 0 --verify-cull-frusta 0 --steps 3 --warmup 1 (BASELINE config 4: 10 000 frusta x the 100 M-point octree)" -o $OUT/${TAG}_query_kernel_stats > /dev/null
grep -E "kernel,|cull_nodes|visible_nodes|query_|nodes_in_location|shape_setup|cull_points" $OUT/${TAG}_query_kernel_stats.csv
grep -E "kernel,|cull_nodes|visible_nodes|query_|cull_points" $OUT/${TAG}_query_kernel_stats_sq.csv
