# round-3 call 5: GPU suite (device PLY decode, batch top-node copies), default bench incl. the end-to-end legs, sharded path at world 1
mkdir -p gpurun_out
T=r03e
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/${T}_gputest.log
timeout 600 python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; echo "default bench rc=$?"
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
EXTRA="--force-sharded --shard-mode octants" run sharded_oct A=1
EXTRA="--force-sharded --shard-mode buckets" run sharded_buckets A=1
python - <<'PY'
import json, glob
for f in ['gpurun_out/r03e_bench_default.json'] + sorted(glob.glob('gpurun_out/r03e_ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get('parity') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'), 'parity', p.get('ok'), p.get('mismatching_nodes'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()})
        if d.get('end_to_end'): print('  e2e', d['end_to_end'])
        if d.get('exchange'): print('  exchange', d['exchange']['ms'], d.get('exchange_per_rank'))
        if d.get('roofline'): print('  roofline', {k: d['roofline'].get(k) for k in ('bound', 'frac', 'profile_matches_build', 'build_hash', 'source')})
        if d.get('encode_sort'): print('  encode_sort', d['encode_sort'])
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -3 gpurun_out/${T}_bench_default.err
