#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run through gpurun), each in its OWN run as MI355X_MICROARCH.md
# prescribes (counters never share a run with --stats; gpurun refuses other trace domains next to --pmc):
#   1 --kernel-trace --stats                       per-kernel durations
#   2 --pmc FETCH_SIZE          3 --pmc WRITE_SIZE  HBM bytes (FETCH_SIZE doubled per the gfx950 correction)
#   4 --pmc SQ_* (issue / wait) + GRBM_GUI_ACTIVE  5 --pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64, _CVT + GRBM_GUI_ACTIVE (clock)
# usage: gpurun -- 'bash tools/profile_bench.sh TAG [extra bench.py flags]'
#   -> gpurun_out/TAG_kernel_stats.csv, _sq.csv, _f64.csv, _traffic.json + the two JSONs bench.py reads
#      (TAG_bench_traffic.json, TAG_bench_valu.json); copy what should be judged into profiles/.
TAG=${1:-prof}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
source $GRAFT_REPO_ROOT/tools/run_limited.sh
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e --no-kernel-events --no-parity"
run_limited 200 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o r -- $B --steps 5 --warmup 2 "$@" > $OUT/${TAG}_stats.log 2>&1
run_limited 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_fetch -o r -- $B --steps 1 --warmup 0 "$@" > $OUT/${TAG}_fetch.log 2>&1
run_limited 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_write -o r -- $B --steps 1 --warmup 0 "$@" > $OUT/${TAG}_write.log 2>&1
run_limited 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/${TAG}_sq -o r -- $B --steps 1 --warmup 0 "$@" > $OUT/${TAG}_sq.log 2>&1
run_limited 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE --kernel-trace -d $OUT/${TAG}_f64 -o r -- $B --steps 1 --warmup 0 "$@" > $OUT/${TAG}_f64.log 2>&1
tail -1 $OUT/${TAG}_stats.log
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py --stats $OUT/${TAG}_stats/r_results.db --fetch $OUT/${TAG}_fetch/r_results.db --write $OUT/${TAG}_write/r_results.db --sq $OUT/${TAG}_sq/r_results.db -o $OUT/${TAG}_kernel_stats > /dev/null
python tools/rocpd_summary.py --stats $OUT/${TAG}_stats/r_results.db --sq $OUT/${TAG}_f64/r_results.db -o $OUT/${TAG}_f64tmp > /dev/null && mv $OUT/${TAG}_f64tmp_sq.csv $OUT/${TAG}_f64.csv && rm -f $OUT/${TAG}_f64tmp*
python tools/make_bench_profile_json.py --tag $OUT/${TAG} "$@"
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_sq $OUT/${TAG}_f64
ls -la $OUT | grep ${TAG}
