# round-3 call 1: GPU suite, the default bench line (with oracle parity), A/B of the event levels, the record-downsweep
# geometries and the sample stride, then the config-3 dress rehearsal (small, then 1 B points)
mkdir -p gpurun_out
T=r03a
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_gputest.log
timeout 400 python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; echo "default bench rc=$?"
B="python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --digest"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > gpurun_out/${T}_ab_$name.json 2> gpurun_out/${T}_ab_$name.err; echo "$name rc=$?"; }
EXTRA="--kernel-events none" run ev_none A=1
EXTRA="--kernel-events all" run ev_all A=1
EXTRA="" run main A=1
for v in 1 2 3 4 5 6; do run rec$v PCV_REC_VARIANT=$v; done
run stride16 PCV_SPEC_STRIDE=16
run main2 A=1
python - <<'PY'
import json, glob
for f in ['gpurun_out/r03a_bench_default.json'] + sorted(glob.glob('gpurun_out/r03a_ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get('parity') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'digest', d.get('tree_digest'), 'parity', p.get('ok'), p.get('mismatching_nodes'),
              {k.replace('_kernel', ''): round(v, 3) for k, v in (d.get('kernel_ms_per_step') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 900 python bench.py --virtual-ranks 8 --points 2000000 --shard-mode both --verify --steps 2 --warmup 1 > gpurun_out/${T}_config3_virtual8_small.json 2> gpurun_out/${T}_config3_virtual8_small.err; echo "virtual small rc=$?"
tail -c 600 gpurun_out/${T}_config3_virtual8_small.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r03a_config3_virtual8_small.json').read().strip().splitlines()[-1])
    print('small parity', d['parity'], {m: (r['vs_single_gpu_build']['ok'], r.get('vs_oracle', {}).get('ok'), r['ms_per_step_slowest_rank']) for m, r in d['shard_modes'].items()})
    ok = d['parity']['ok']
except Exception as e:
    print('small ERR', e); ok = False
open('gpurun_out/r03a_small_ok', 'w').write('1' if ok else '0')
PY
if [ "$(cat gpurun_out/r03a_small_ok)" = "1" ]; then
  timeout 1500 python bench.py --virtual-ranks 8 --shard-mode both --verify --steps 2 --warmup 1 > gpurun_out/${T}_config3_virtual8_1B.json 2> gpurun_out/${T}_config3_virtual8_1B.err; echo "virtual 1B rc=$?"
  tail -c 600 gpurun_out/${T}_config3_virtual8_1B.err
  python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r03a_config3_virtual8_1B.json').read().strip().splitlines()[-1])
    print('1B parity', d['parity'], d['single_gpu_build']['vs_oracle'], {m: (r['vs_single_gpu_build']['ok'], r.get('vs_oracle', {}).get('ok'), r['ms_per_step_slowest_rank'], r['exchange_rank0']['imbalance_max_over_mean']) for m, r in d['shard_modes'].items()})
except Exception as e:
    print('1B ERR', e)
PY
fi
