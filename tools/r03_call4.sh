# round-3 call 4: GPU suite, record downsweep with two-phase staging (tiles of 16 384), query bench with full parity
mkdir -p gpurun_out
T=r03d
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/${T}_gputest.log
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
EXTRA="" run main A=1
run rec7 PCV_REC_VARIANT=7
run rec4 PCV_REC_VARIANT=4
EXTRA="" run main2 A=1
run rec7b PCV_REC_VARIANT=7
timeout 900 python bench.py --query > gpurun_out/${T}_query.json 2> gpurun_out/${T}_query.err; echo "query rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03d_ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
try:
    d = json.loads(open('gpurun_out/r03d_query.json').read().strip().splitlines()[-1])
    print('query', d['value'], d['cull_nodes'], d['visible_nodes'], d['query_points'], d['parity'], d['cpu_baseline'])
except Exception as e:
    print('query ERR', e)
PY
tail -3 gpurun_out/${T}_query.err
