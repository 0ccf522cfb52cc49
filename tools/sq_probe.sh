#!/bin/bash
# One rocprofv3 SQ-counter pass over a short bench.py run (own run, --kernel-trace only, as gpurun requires):
#   gpurun -- 'ENV=... bash tools/sq_probe.sh TAG [bench flags]'  ->  gpurun_out/TAG_sq.csv (per-kernel averages per launch)
#   PROBE_CMD="python tools/query_probe.py" profiles another command from the repo root instead of bench.py
TAG=${1:-sq}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $OUT/${TAG}_sqdb -o r -- ${PROBE_CMD:-python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-kernel-events} "$@" > $OUT/${TAG}_sq.log 2>&1
cd $GRAFT_REPO_ROOT && python - "$OUT/${TAG}_sqdb/r_results.db" "$OUT/${TAG}_sq.csv" <<'PY'
import sqlite3, sys
from collections import defaultdict
sys.path.insert(0, "tools")
from rocpd_summary import short
cur = sqlite3.connect(sys.argv[1]).cursor()
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for name, cn, val in cur.execute("select kernel_name,counter_name,value from counters_collection"):
    c = acc[short(name)][cn]
    c[0] += 1
    c[1] += val
names = sorted({cn for k in acc.values() for cn in k})
rows = []
for k, c in acc.items():
    avg = {cn: (c[cn][1] / c[cn][0] if cn in c and c[cn][0] else 0.0) for cn in names}
    if avg.get("SQ_INSTS_VALU", 0) >= 1e6 and "kernel" in k and not k.startswith("at::"):
        rows.append((k, max(v[0] for v in c.values()), avg))
rows.sort(key=lambda r: -r[2].get("SQ_INSTS_VALU", 0))
with open(sys.argv[2], "w") as f:
    f.write("kernel,launches," + ",".join(names) + ",valu_per_wave\n")
    for k, launches, avg in rows:
        per = avg.get("SQ_INSTS_VALU", 0) / max(avg.get("SQ_WAVES", 1), 1)
        f.write(",".join([k, str(launches)] + [f"{avg[cn]:.0f}" for cn in names] + [f"{per:.1f}"]) + "\n")
print(open(sys.argv[2]).read())
PY
rm -rf $OUT/${TAG}_sqdb
