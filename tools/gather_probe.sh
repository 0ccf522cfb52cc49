#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/gather_probe.bin (separate rocprofv3 --pmc passes, --kernel-trace only) -> gpurun_out/TAG_gather_probe.json
TAG=${1:-gp}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
source $GRAFT_REPO_ROOT/tools/run_limited.sh
cd /tmp && export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/tools/gather_probe.bin
run_limited 120 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_gs -o r -- $P > $OUT/${TAG}_gather_probe_algorithmic.json 2> $OUT/${TAG}_gather_stats.log
run_limited 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/${TAG}_gf -o r -- $P > /dev/null 2>&1
run_limited 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/${TAG}_gw -o r -- $P > /dev/null 2>&1
run_limited 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d /tmp/${TAG}_ge -o r -- $P > /dev/null 2>&1
run_limited 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d /tmp/${TAG}_gh -o r -- $P > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
DBS=""; for d in gf gw ge gh; do [ -f /tmp/${TAG}_$d/r_results.db ] && DBS="$DBS /tmp/${TAG}_$d/r_results.db"; done
python tools/pmc_table.py $DBS --min-us 0 -o $OUT/${TAG}_gather_pmc.json > /dev/null 2>&1
python tools/rocpd_summary.py --stats /tmp/${TAG}_gs/r_results.db --fetch /tmp/${TAG}_gf/r_results.db --write /tmp/${TAG}_gw/r_results.db -o $OUT/${TAG}_gather_kernel_stats > /dev/null 2>&1
cat $OUT/${TAG}_gather_kernel_stats.csv; tail -1 $OUT/${TAG}_gather_probe_algorithmic.json
python - <<PY
import json
try:
    d = json.load(open("$OUT/${TAG}_gather_pmc.json"))
    for k, v in d["kernels"].items():
        print(k, {a: b for a, b in v.items() if not a.startswith("_") or a in ("_launches", "_avg_us")})
except Exception as e:
    print("pmc table:", e)
PY
