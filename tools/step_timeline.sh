#!/bin/bash
# Kernel timeline of the LAST build step of a short bench.py run (rocprofv3 --kernel-trace):
#   gpurun -- 'bash tools/step_timeline.sh TAG [bench flags]'  ->  gpurun_out/TAG_timeline.txt
# one line per kernel launch: start offset from the step's first kernel (us), duration (us), gap to the previous end (us)
TAG=${1:-tl}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
source $GRAFT_REPO_ROOT/tools/run_limited.sh
run_limited 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/${TAG}_tl -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 4 --no-cpu-baseline --no-e2e --no-kernel-events "$@" > $OUT/${TAG}_tl.log 2>&1
python - /tmp/${TAG}_tl $OUT/${TAG}_timeline.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last aabb_partial launch starts the last step
starts = [i for i, r in enumerate(rows) if "aabb_partial" in r[2]]
first = starts[-1]
t0 = rows[first][0]
prev_end = t0
out = []
for s, e, name in rows[first:]:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
    out.append(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {short}")
    prev_end = e
open(sys.argv[2], "w").write("  start_us   dur_us  gap_us  kernel\n" + "\n".join(out) + "\n")
print(open(sys.argv[2]).read())
PY
