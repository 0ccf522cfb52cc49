# round-3 call 40: scheduler strategies for pcv_encode.hip (chain pass, settle, climb): same arithmetic, other instruction order
mkdir -p gpurun_out
T=r03H
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
run main A=1
run maxilp PCV_HIP_LIBRARY=point_cloud_viewer_amd/libpcv_hip_maxilp.so
run memclause PCV_HIP_LIBRARY=point_cloud_viewer_amd/libpcv_hip_memclause.so
run nopost PCV_HIP_LIBRARY=point_cloud_viewer_amd/libpcv_hip_nopost.so
run main2 A=1
run maxilp2 PCV_HIP_LIBRARY=point_cloud_viewer_amd/libpcv_hip_maxilp.so
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03H_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
