# round-3 call 19: settle tile 512 (ships) against 1024 / 2048 slots per workgroup (more record loads in flight per lane)
mkdir -p gpurun_out
T=r03t
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
run main A=1
run tile1024 PCV_HIP_LIBRARY=point_cloud_viewer_amd/libpcv_hip_tile1024.so
run tile2048 PCV_HIP_LIBRARY=point_cloud_viewer_amd/libpcv_hip_tile2048.so
run main2 A=1
run tile1024b PCV_HIP_LIBRARY=point_cloud_viewer_amd/libpcv_hip_tile1024.so
run tile2048b PCV_HIP_LIBRARY=point_cloud_viewer_amd/libpcv_hip_tile2048.so
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03t_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('tree_digest'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
