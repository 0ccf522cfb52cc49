#!/bin/bash
# timing-only A/B of library variants in one gpurun call: RUNS="name[:lib-tag | :ENV=VAL,ENV2=VAL2] ..." (lib-tag: libpcv_hip_<tag>.so from
# tools/build_variants.sh; ENV: switches of libpcv_hip_exp.so); config 2 alone (--no-legs), kernel table per run; REPS alternations
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
B="python bench.py --steps ${STEPS:-10} --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest --no-legs"
for rep in $(seq 1 ${REPS:-2}); do
for v in $RUNS; do
  name=${v%%:*}; arg=${v#*:}; [ "$name" = "$v" ] && arg=""
  out=gpurun_out/abl_${name}_$rep.json
  if [ -f "$PWD/point_cloud_viewer_amd/libpcv_hip_$arg.so" ]; then PCV_HIP_LIBRARY=$PWD/point_cloud_viewer_amd/libpcv_hip_$arg.so timeout 200 $B > $out 2> $out.err
  elif [ -n "$arg" ]; then env PCV_HIP_LIBRARY=exp ${arg//,/ } timeout 200 $B > $out 2> $out.err
  else timeout 200 $B > $out 2> $out.err; fi
  python - <<PY
import json
try:
    d=json.loads(open('$out').read().strip().splitlines()[-1])
    print('$name', $rep, d['value'], d['ms_per_step'], (d.get('tree_digest') or '')[:8], {k.replace('_kernel',''):round(v,3) for k,v in (d.get('kernel_ms_per_step') or {}).items()})
except Exception as e: print('$name', 'ERR', e)
PY
done; done
