# one gpurun call: GPU tests, the bench with full-size parity, then timing-only A/B runs.
# VARIANTS: space-separated "name" (library built by tools/build_variants.sh) or "name:ENV=VALUE" (environment switch)
mkdir -p gpurun_out
T=${TAG:-ab}
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
timeout 300 python bench.py --steps 10 --warmup 3 --no-e2e --verify > gpurun_out/${T}_bench_main.json 2> gpurun_out/${T}_bench_main.err; echo "rc=$?"
NAMES=""
for v in $VARIANTS; do
  name=${v%%:*}; NAMES="$NAMES $name"
  if [ "$name" != "$v" ]; then
    env PCV_HIP_LIBRARY=exp "${v#*:}" timeout 200 $B > gpurun_out/${T}_bench_$name.json 2> gpurun_out/${T}_bench_$name.err
  else
    PCV_HIP_LIBRARY=$PWD/point_cloud_viewer_amd/libpcv_hip_$name.so timeout 200 $B > gpurun_out/${T}_bench_$name.json 2> gpurun_out/${T}_bench_$name.err
  fi
  echo "$name rc=$?"
done
timeout 200 $B > gpurun_out/${T}_bench_main2.json 2> gpurun_out/${T}_bench_main2.err; echo "rc=$?"
tail -4 gpurun_out/${T}_gputest.log
for f in main $NAMES main2; do python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${T}_bench_$f.json').read().strip().splitlines()[-1])
    p=d.get('parity') or {}
    print('$f', d['value'], d['ms_per_step'], 'parity_ok=%s mism=%s' % (p.get('ok'), p.get('mismatching_nodes')), {k:round(v,3) for k,v in (d.get('kernel_ms_per_step') or {}).items()})
except Exception as e: print('$f', 'ERR', e)
PY
done
