# round-3 call 20: buckets for the codes of Float32-coded levels (side array behind `wide`) against one entry per input index
mkdir -p gpurun_out
T=r03u
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${T}_gputest.log
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
EXTRA="" run main A=1
run exp_buckets PCV_HIP_LIBRARY=exp
run nobuckets PCV_HIP_LIBRARY=exp PCV_WIDE_BUCKETS=0
EXTRA="" run main2 A=1
run nobuckets2 PCV_HIP_LIBRARY=exp PCV_WIDE_BUCKETS=0
EXTRA="--force-sharded --shard-mode octants" run sharded_oct A=1
timeout 400 python bench.py --no-e2e --no-cpu-baseline > gpurun_out/${T}_bench_parity.json 2> gpurun_out/${T}_bench_parity.err; echo "parity bench rc=$?"
timeout 600 python bench.py --ecef --points 500000000 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --verify > gpurun_out/${T}_config5_ecef_500M.json 2> gpurun_out/${T}_config5.err; echo "config5 rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03u_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get('parity') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'), 'parity', p.get('ok'), p.get('mismatching_nodes'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()}, {k: round(v, 3) for k, v in d['stage_ms'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
