#!/bin/bash
# One more rocprofv3 --pmc pass over bench.py: LDS counters of the sort / settle kernels (bank conflicts, active cycles)
#   gpurun -- 'bash tools/profile_lds.sh TAG'  ->  gpurun_out/TAG_lds.csv
TAG=${1:-lds}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e --no-kernel-events --no-parity"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o r -- $B --steps 2 --warmup 1 "$@" > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/${TAG}_lds -o r -- $B --steps 1 --warmup 0 "$@" > $OUT/${TAG}_ldsrun.log 2>&1
tail -3 $OUT/${TAG}_ldsrun.log
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py --stats $OUT/${TAG}_stats/r_results.db --sq $OUT/${TAG}_lds/r_results.db -o $OUT/${TAG}_tmp > /dev/null && mv $OUT/${TAG}_tmp_sq.csv $OUT/${TAG}_lds.csv
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_lds $OUT/${TAG}_tmp*
cat $OUT/${TAG}_lds.csv | cut -c1-220
