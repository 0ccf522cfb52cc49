mkdir -p gpurun_out
T=r03G
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
run fused1024 A=1
run unfused PCV_HIP_LIBRARY=exp PCV_SETTLE_IN_SORT=0
run fused512x8 PCV_HIP_LIBRARY=exp PCV_FUSE_GEOM=512
run fused512x16 PCV_HIP_LIBRARY=exp PCV_FUSE_GEOM=516
run unfused2 PCV_HIP_LIBRARY=exp PCV_SETTLE_IN_SORT=0
run fused512x8b PCV_HIP_LIBRARY=exp PCV_FUSE_GEOM=512
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03G_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
