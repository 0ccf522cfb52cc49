mkdir -p gpurun_out
T=r03K
B="python bench.py --steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
for i in 1 2 3 4; do
  run pieces$i PCV_HIP_LIBRARY=exp
  run counts$i PCV_HIP_LIBRARY=exp PCV_SORT_ROWS2=0
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03K_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d['kernel_ms_per_step']
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'sort_records', round(d['stage_ms']['sort_records'], 3), 'downsweep', round(k.get('downsweep_rec_kernel', 0), 3), 'upsweep', round(k.get('upsweep_kernel<u32>', 0), 3))
    except Exception as e:
        print(f, 'ERR', e)
PY
