# round-3 call 41: first sort pass without its own pass over the keys (histogram from the rank counts per sort workgroup, map
# applied inside the downsweep) against upsweep_map
mkdir -p gpurun_out
T=r03J
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_gputest.log
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
run main A=1
run pass2counts PCV_HIP_LIBRARY=exp PCV_SORT_ROWS2=0
run main2 A=1
run pass2counts2 PCV_HIP_LIBRARY=exp PCV_SORT_ROWS2=0
timeout 400 python bench.py --no-e2e --no-cpu-baseline > gpurun_out/${T}_bench_parity.json 2> gpurun_out/${T}_bench_parity.err; echo "parity bench rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03J_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get('parity') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'), 'parity', p.get('ok'), p.get('mismatching_nodes'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()}, {k: round(v, 3) for k, v in d['stage_ms'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -3 gpurun_out/${T}_ab_main.err
