#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run through gpurun): per-kernel durations (--kernel-trace --stats) and,
# in separate passes as MI355X_MICROARCH.md prescribes, the HBM counters FETCH_SIZE and WRITE_SIZE.
# usage: gpurun -- 'bash tools/profile_bench.sh TAG [extra bench.py flags]'   -> gpurun_out/TAG_{stats,fetch,write}/
TAG=${1:-prof}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-kernel-events "$@" > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_fetch -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-kernel-events "$@" > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_write -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-kernel-events "$@" > $OUT/${TAG}_write.log 2>&1
tail -1 $OUT/${TAG}_stats.log
# optional 4th pass (PCV_PROFILE_SQ=1): SQ counters for the issue-bound kernels (VALU instructions, wave / wait cycles)
if [ -n "$PCV_PROFILE_SQ" ]; then
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $OUT/${TAG}_sq -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-kernel-events "$@" > $OUT/${TAG}_sq.log 2>&1
  ls $OUT/${TAG}_sq
fi
# summarise on the box and drop the databases (gpurun copies at most 64 MiB back)
SQ_ARG=""
[ -n "$PCV_PROFILE_SQ" ] && SQ_ARG="--sq $OUT/${TAG}_sq/r_results.db"
cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py --stats $OUT/${TAG}_stats/r_results.db --fetch $OUT/${TAG}_fetch/r_results.db --write $OUT/${TAG}_write/r_results.db $SQ_ARG -o $OUT/${TAG}_kernel_stats > /dev/null
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_sq
ls -la $OUT
