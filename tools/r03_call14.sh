# round-3 call 14: what the sparse wide-code gather costs `settle` (PCV_WIDE_MASK folds the gathers onto 16 KB: wrong bytes,
# timing only) and XCD-aware settle items (PCV_SETTLE_XCD)
mkdir -p gpurun_out
T=r03p
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
EXTRA="" run main A=1
run exp_plain PCV_HIP_LIBRARY=exp
run widemask PCV_HIP_LIBRARY=exp PCV_WIDE_MASK=1023
run xcd PCV_HIP_LIBRARY=exp PCV_SETTLE_XCD=1
run xcd_widemask PCV_HIP_LIBRARY=exp PCV_SETTLE_XCD=1 PCV_WIDE_MASK=1023
EXTRA="" run main2 A=1
run widemask2 PCV_HIP_LIBRARY=exp PCV_WIDE_MASK=1023
run xcd2 PCV_HIP_LIBRARY=exp PCV_SETTLE_XCD=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03p_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
