# round-3 call 9: chain continuation inside the leaf-wise settle kernel; sample strides 64 / 128 now that no payload patch depends on the band
mkdir -p gpurun_out
T=r03i
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${T}_gputest.log
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
EXTRA="" run main A=1
run sepcont PCV_HIP_LIBRARY=exp PCV_CONT_IN_SETTLE=0
run stride64 PCV_HIP_LIBRARY=exp PCV_SPEC_STRIDE=64
run stride128 PCV_HIP_LIBRARY=exp PCV_SPEC_STRIDE=128
run stride16 PCV_HIP_LIBRARY=exp PCV_SPEC_STRIDE=16
EXTRA="" run main2 A=1
run sepcont2 PCV_HIP_LIBRARY=exp PCV_CONT_IN_SETTLE=0
timeout 400 python bench.py --no-e2e --no-cpu-baseline > gpurun_out/${T}_bench_parity.json 2> gpurun_out/${T}_bench_parity.err; echo "parity bench rc=$?"
timeout 600 python bench.py --ecef --points 500000000 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --verify > gpurun_out/${T}_config5_ecef_500M.json 2> gpurun_out/${T}_config5.err; echo "config5 rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03i_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get('parity') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'), 'parity', p.get('ok'), p.get('mismatching_nodes'), (d.get('build_info') or {}).get('continued_points'), (d.get('build_info') or {}).get('predicted_nodes'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()}, {k: round(v, 3) for k, v in d['stage_ms'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
