# round-3 call 18: what `settle` spends on its stores and on the final re-encode (timing variants, wrong output)
mkdir -p gpurun_out
T=r03s
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity"
run() { name=$1; shift; env "$@" timeout 200 $B > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
run main A=1
run nostores PCV_HIP_LIBRARY=point_cloud_viewer_amd/libpcv_hip_nostores.so
run noreenc PCV_HIP_LIBRARY=point_cloud_viewer_amd/libpcv_hip_noreenc.so
run main2 A=1
run nostores2 PCV_HIP_LIBRARY=point_cloud_viewer_amd/libpcv_hip_nostores.so
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03s_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'],
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
