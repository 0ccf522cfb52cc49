#!/bin/bash
# Counter passes over one short bench.py run each: usage  bash tools/pmc_pass.sh TAG "SET1 counters" "SET2 counters" ...
# (each set is its own rocprofv3 --pmc run, with --kernel-trace only) -> gpurun_out/TAG_pmc.json
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e --no-kernel-events --no-parity --steps 1 --warmup 1 ${PMC_BENCH_ARGS}"
i=0; DBS=""
for set in "$@"; do
  i=$((i+1)); rm -rf /tmp/${TAG}_pmc$i
  rocprofv3 --pmc $set --kernel-trace -d /tmp/${TAG}_pmc$i -o r -- $B > $OUT/${TAG}_pmc$i.log 2>&1 || echo "set $i failed: $set"
  [ -f /tmp/${TAG}_pmc$i/r_results.db ] && DBS="$DBS /tmp/${TAG}_pmc$i/r_results.db"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_table.py $DBS -o $OUT/${TAG}_pmc.json
