#!/bin/bash
# Counter passes over one short bench.py run each: usage  bash tools/pmc_pass.sh TAG "SET1 counters" "SET2 counters" ...
# (each set is its own rocprofv3 --pmc run, with --kernel-trace only, under its own timeout: a set the hardware cannot
# collect makes rocprofv3 abort and then hang in its signal handler) -> gpurun_out/TAG_pmc.json, rewritten after every pass
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
source $GRAFT_REPO_ROOT/tools/run_limited.sh
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e --no-kernel-events --no-parity --steps 1 --warmup 1 ${PMC_BENCH_ARGS}"
i=0; DBS=""
for set in "$@"; do
  i=$((i+1)); rm -rf /tmp/${TAG}_pmc$i
  run_limited ${PMC_TIMEOUT:-100} rocprofv3 --pmc $set --kernel-trace -d /tmp/${TAG}_pmc$i -o r -- $B > $OUT/${TAG}_pmc$i.log 2>&1 || echo "set $i failed: $set"
  if [ -f /tmp/${TAG}_pmc$i/r_results.db ]; then
    DBS="$DBS /tmp/${TAG}_pmc$i/r_results.db"
    (cd $GRAFT_REPO_ROOT && python tools/pmc_table.py $DBS -o $OUT/${TAG}_pmc.json > $OUT/${TAG}_pmc_table.log 2>&1)
  fi
done
cat $OUT/${TAG}_pmc_table.log
