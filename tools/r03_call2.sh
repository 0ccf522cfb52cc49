# round-3 call 2: first-candidate codes (no kept-code patch), tiles of 8 192 in the record downsweep
mkdir -p gpurun_out
T=r03b
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/${T}_gputest.log
timeout 400 python bench.py --no-e2e > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; echo "default bench rc=$?"
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
EXTRA="" run main A=1
run rec0 PCV_REC_VARIANT=0
run rec4 PCV_REC_VARIANT=4
EXTRA="--kernel-events all" run ev_all A=1
EXTRA="--force-sharded --shard-mode octants" run sharded_oct A=1
EXTRA="--force-sharded --shard-mode buckets --verify" run sharded_buckets_verify A=1
EXTRA="" run main2 A=1
timeout 600 python bench.py --ecef --points 500000000 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --verify > gpurun_out/${T}_config5_ecef_500M.json 2> gpurun_out/${T}_config5.err; echo "config5 rc=$?"
python - <<'PY'
import json, glob
for f in ['gpurun_out/r03b_bench_default.json', 'gpurun_out/r03b_config5_ecef_500M.json'] + sorted(glob.glob('gpurun_out/r03b_ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get('parity') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'), 'parity', p.get('ok'), p.get('mismatching_nodes'), d.get('build_info'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -3 gpurun_out/${T}_bench_default.err
