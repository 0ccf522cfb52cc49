# round-3 call 10: adaptive sample levels; GPU suite; 1 B points on one GPU with parity (8-bit digits, global rank map); timeline
mkdir -p gpurun_out
T=r03j
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${T}_gputest.log
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
EXTRA="" run main A=1
run lev14 PCV_HIP_LIBRARY=exp PCV_SAMPLE_LEVELS=14
run lev11 PCV_HIP_LIBRARY=exp PCV_SAMPLE_LEVELS=11
EXTRA="" run main2 A=1
bash tools/step_timeline.sh ${T} --no-parity > /dev/null 2>&1; echo "timeline rc=$?"
timeout 900 python bench.py --points 1000000000 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --verify > gpurun_out/${T}_parity_1B_single_gpu.json 2> gpurun_out/${T}_1B.err; echo "1B rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03j_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get('parity') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'), 'parity', p.get('ok'), p.get('mismatching_nodes'), p.get('nodes'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()}, {k: round(v, 3) for k, v in d['stage_ms'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -2 gpurun_out/${T}_1B.err
