# timing-only A/B in one gpurun call: RUNS="name[:ENV=VAL[,ENV2=VAL2]|lib] ..." each run = one short bench; prints the kernel table
mkdir -p gpurun_out
T=${TAG:-abq}
B="python bench.py --steps ${STEPS:-10} --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
i=0
for v in $RUNS; do
  i=$((i+1)); name=${v%%:*}; arg=${v#*:}
  if [ "$name" = "$v" ]; then arg=""; fi
  if [ -f "$PWD/point_cloud_viewer_amd/libpcv_hip_$arg.so" ]; then
    PCV_HIP_LIBRARY=$PWD/point_cloud_viewer_amd/libpcv_hip_$arg.so timeout 200 $B > gpurun_out/${T}_$i.json 2> gpurun_out/${T}_$i.err
  elif [ -n "$arg" ]; then
    env PCV_HIP_LIBRARY=exp ${arg//,/ } timeout 200 $B > gpurun_out/${T}_$i.json 2> gpurun_out/${T}_$i.err
  else
    timeout 200 $B > gpurun_out/${T}_$i.json 2> gpurun_out/${T}_$i.err
  fi
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${T}_$i.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('tree_digest') or '')[:8], {k.replace('_kernel',''):round(v,3) for k,v in (d.get('kernel_ms_per_step') or {}).items()})
except Exception as e: print('$name', 'ERR', e)
PY
done
