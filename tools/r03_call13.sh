# round-3 call 13: chain loop variant (level constants one level ahead, octant bits as booleans); warm timeline
mkdir -p gpurun_out
T=r03m
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
EXTRA="" run main A=1
run chainv2 PCV_HIP_LIBRARY=$PWD/point_cloud_viewer_amd/libpcv_hip_chainv2.so
EXTRA="" run main2 A=1
run chainv2b PCV_HIP_LIBRARY=$PWD/point_cloud_viewer_amd/libpcv_hip_chainv2.so
EXTRA="" run main3 A=1
run chainv2c PCV_HIP_LIBRARY=$PWD/point_cloud_viewer_amd/libpcv_hip_chainv2.so
bash tools/step_timeline.sh ${T} --no-parity > /dev/null 2>&1; echo "timeline rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03m_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
awk 'NR>3 && ($3+0 > 3.0 || $2+0 > 50) {print}' gpurun_out/${T}_timeline.txt
