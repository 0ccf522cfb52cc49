#!/bin/bash
# VALU issue cost per instruction class + the clock sustained under each: tools/f64_rate.bin plain and
# under a GRBM_GUI_ACTIVE counter pass -> gpurun_out/<TAG>_f64_rate.json (copy to profiles/). usage: bash tools/f64_rate.sh TAG
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
source $GRAFT_REPO_ROOT/tools/run_limited.sh
cd /tmp && export TMPDIR=/tmp
$GRAFT_REPO_ROOT/tools/f64_rate.bin > $OUT/${TAG}_f64_rate_plain.jsonl 2> $OUT/${TAG}_f64_rate.err
rm -rf /tmp/${TAG}_f64clk
run_limited 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/${TAG}_f64clk -o r -- $GRAFT_REPO_ROOT/tools/f64_rate.bin > $OUT/${TAG}_f64_rate_pmc.jsonl 2>> $OUT/${TAG}_f64_rate.err
cd $GRAFT_REPO_ROOT
python tools/f64_rate_merge.py $OUT/${TAG}_f64_rate_plain.jsonl /tmp/${TAG}_f64clk/r_results.db $OUT/${TAG}_f64_rate.json
