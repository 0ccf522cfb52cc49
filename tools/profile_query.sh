#!/bin/bash
# rocprofv3 passes over `bench.py --query` (BASELINE config 4) on the GPU box, each in its OWN run (counters never share a
# run with --stats), each under a process-group watchdog: kernel durations, FETCH_SIZE, WRITE_SIZE, SQ counters of the query
# kernels. usage: gpurun -- 'bash tools/profile_query.sh TAG'
#   -> gpurun_out/TAG_query_kernel_stats.csv, _sq.csv, _traffic.json and TAG_query_counters.json (what bench.py quotes as
#      query.roofline.profile; copy to profiles/<round>_query_counters.json)
TAG=${1:-qprof}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
source $GRAFT_REPO_ROOT/tools/run_limited.sh
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --query --verify-frusta 0 --verify-cull-frusta 0 --query-steps 3 --warmup 1"
run_limited 200 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_qs -o r -- $B "$@" > $OUT/${TAG}_query_stats.log 2>&1
run_limited 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/${TAG}_qf -o r -- $B "$@" > /dev/null 2>&1
run_limited 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/${TAG}_qw -o r -- $B "$@" > /dev/null 2>&1
run_limited 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d /tmp/${TAG}_qq -o r -- $B "$@" > /dev/null 2>&1
run_limited 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT --kernel-trace -d /tmp/${TAG}_q64 -o r -- $B "$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py --stats /tmp/${TAG}_qs/r_results.db --fetch /tmp/${TAG}_qf/r_results.db --write /tmp/${TAG}_qw/r_results.db --sq /tmp/${TAG}_qq/r_results.db \
  --command "python bench.py --query --verify-frusta 0 --verify-cull-frusta 0 --query-steps 3 --warmup 1 (BASELINE config 4: 10 000 frusta x the 100 M-point octree)" -o $OUT/${TAG}_query_kernel_stats > /dev/null
python tools/pmc_table.py /tmp/${TAG}_qf/r_results.db /tmp/${TAG}_qw/r_results.db /tmp/${TAG}_qq/r_results.db /tmp/${TAG}_q64/r_results.db --min-us 5 -o $OUT/${TAG}_query_pmc.json > /dev/null
python - <<PY
import json, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from bench import build_hash
d = json.load(open("$OUT/${TAG}_query_pmc.json"))
keep = {}
for k, v in d["kernels"].items():
    name = k.split("<")[0]
    if name not in ("cull_nodes_kernel", "cull_nodes_sparse_kernel", "visible_nodes_kernel", "query_flags_kernel", "query_compact_kernel", "cull_points_kernel", "shape_setup_kernel"):
        continue
    e = dict(v)
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:  # KiB counters; FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM)
        e["hbm_bytes_per_launch"] = round(2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024)
    if v.get("GRBM_GUI_ACTIVE") and v.get("_avg_us_under_counters"):
        c = v["GRBM_GUI_ACTIVE"] / (v["_avg_us_under_counters"] * 1e3)
        e["sustained_clock_GHz"] = round(c / 8 if c > 4 else c, 3)
    if v.get("SQ_INSTS_VALU") and v.get("SQ_WAVE_CYCLES"):
        e["wait_share_of_wave_cycles"] = round(v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"], 3)
        e["issue_wait_share_of_wave_cycles"] = round(v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"], 3)
    keep[name] = e
json.dump({"build_hash": build_hash(), "command": "python bench.py --query --verify-frusta 0 --verify-cull-frusta 0 --query-steps 3 --warmup 1",
           "note": "per launch, separate rocprofv3 --pmc passes", "kernels": keep}, open("$OUT/${TAG}_query_counters.json", "w"), indent=1)
print(json.dumps(keep, indent=0)[:1500])
PY
