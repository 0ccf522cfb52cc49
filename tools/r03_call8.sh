# round-3 call 8: chain pass variants: one workgroup per 512 points (ships), persistent with prefetch (3 per CU), paired deal without persistence
mkdir -p gpurun_out
T=r03h
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
EXTRA="" run main A=1
for k in 3 100; do run persist$k PCV_HIP_LIBRARY=exp PCV_SPEC_PERSIST=$k; done
EXTRA="" run main2 A=1
run persist100b PCV_HIP_LIBRARY=exp PCV_SPEC_PERSIST=100
cd /tmp && export TMPDIR=/tmp
for k in 0 3 100; do
  PCV_HIP_LIBRARY=exp PCV_SPEC_PERSIST=$k rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d /tmp/sq_$k -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-e2e --no-cpu-baseline --no-parity --no-kernel-events > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py --stats /tmp/sq_$k/r_results.db --sq /tmp/sq_$k/r_results.db -o $GRAFT_REPO_ROOT/gpurun_out/${T}_sq_persist$k > /dev/null 2>&1
  grep -E "kernel,|spec_encode" $GRAFT_REPO_ROOT/gpurun_out/${T}_sq_persist${k}_sq.csv
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03h_ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
