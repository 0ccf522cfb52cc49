# round-3 call 45: counting-pass-free record sort for predicted trees of up to 16 384 nodes (250 M points take it now)
mkdir -p gpurun_out
T=r03M
timeout 600 python -m pytest tests/test_gpu_single_chain.py tests/test_gpu_build.py tests/test_gpu_fuzz.py tests/test_gpu_sharded.py -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_gputest.log | head -1
timeout 600 python bench.py --points 250000000 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --verify > gpurun_out/${T}_parity_250M.json 2> gpurun_out/${T}_parity_250M.err; echo "250M rc=$?"
PCV_HIP_LIBRARY=exp PCV_SORT_ROWS=0 timeout 300 python bench.py --points 250000000 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-parity --digest > gpurun_out/${T}_250M_counting.json 2> /dev/null; echo "250M counting rc=$?"
timeout 300 python bench.py --points 250000000 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-parity --digest > gpurun_out/${T}_250M_rows.json 2> /dev/null; echo "250M rows rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-parity --digest > gpurun_out/${T}_100M.json 2> /dev/null; echo "100M rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03M_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get('parity') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('tree_digest'), 'parity', p.get('ok'), p.get('mismatching_nodes'), {k: round(v, 3) for k, v in d['stage_ms'].items()}, list((d.get('kernel_ms_per_step') or {}).keys()))
    except Exception as e:
        print(f, 'ERR', e)
PY
