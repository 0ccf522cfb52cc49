mkdir -p gpurun_out
T=r03F
timeout 600 python -m pytest tests/test_gpu_single_chain.py tests/test_gpu_build.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${T}_gputest.log
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
run main A=1
run unfused PCV_HIP_LIBRARY=exp PCV_SETTLE_IN_SORT=0
run main2 A=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03F_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()}, {k: round(v, 3) for k, v in d['stage_ms'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
