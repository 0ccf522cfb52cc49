mkdir -p gpurun_out
T=r03L
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_gputest.log | head -1
B="python bench.py --steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
for i in 1 2 3; do
  run pieces$i PCV_HIP_LIBRARY=exp
  run counts$i PCV_HIP_LIBRARY=exp PCV_SORT_ROWS2=0
done
timeout 400 python bench.py --no-e2e --no-cpu-baseline > gpurun_out/${T}_bench_parity.json 2> gpurun_out/${T}_bench_parity.err; echo "parity bench rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03L_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d['kernel_ms_per_step']; p = d.get('parity') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('tree_digest'), 'parity', p.get('ok'), p.get('mismatching_nodes'), 'sort_records', round(d['stage_ms']['sort_records'], 3), 'downsweep', round(k.get('downsweep_rec_kernel', 0), 3), 'upsweep', round(k.get('upsweep_kernel<u32>', 0), 3))
    except Exception as e:
        print(f, 'ERR', e)
PY
