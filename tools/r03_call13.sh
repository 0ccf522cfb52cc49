# round-3 call 13: node split with one launch per level (the last workgroup appends the level) against two launches
mkdir -p gpurun_out
T=r03o
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${T}_gputest.log
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
EXTRA="" run main A=1
run twolaunch PCV_HIP_LIBRARY=exp PCV_SPLIT_FUSED=0
EXTRA="" run main2 A=1
run twolaunch2 PCV_HIP_LIBRARY=exp PCV_SPLIT_FUSED=0
EXTRA="--force-sharded --shard-mode octants" run sharded_oct A=1
bash tools/step_timeline.sh ${T} --no-parity > /dev/null 2>&1; echo "timeline rc=$?"
timeout 400 python bench.py --no-e2e --no-cpu-baseline > gpurun_out/${T}_bench_parity.json 2> gpurun_out/${T}_bench_parity.err; echo "parity bench rc=$?"
timeout 400 python bench.py --force-sharded --config3 --points 20000000 --verify --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/${T}_config3_world1_20M.json 2> gpurun_out/${T}_config3_world1.err; echo "config3 world-1 rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03o_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get('parity') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'), 'parity', p.get('ok'), p.get('mismatching_nodes'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()}, {k: round(v, 3) for k, v in d['stage_ms'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
awk 'NR>3 && ($3+0 > 3.0 || $2+0 > 50) {print}' gpurun_out/${T}_timeline.txt
tail -3 gpurun_out/${T}_config3_world1.err
