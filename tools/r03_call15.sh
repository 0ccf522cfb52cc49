# round-3 call 15: partition_scatter with plane-major copies; sharded path at world size 1
mkdir -p gpurun_out
T=r03o
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${T}_gputest.log
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
EXTRA="--force-sharded --shard-mode octants" run sharded_oct A=1
EXTRA="--force-sharded --shard-mode buckets --verify" run sharded_buckets_verify A=1
EXTRA="--force-sharded --shard-mode octants" run sharded_oct2 A=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03o_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get('parity') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'), 'parity', p.get('ok'), p.get('mismatching_nodes'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()}, d['exchange']['ms'])
    except Exception as e:
        print(f, 'ERR', e)
PY
