# the records that depend on the library build, after a change that leaves the big-cloud paths alone: GPU suite, profile
# passes + default bench line (tied to the build), timeline, sharded path at world size 1, config-3 rehearsal (8 x 125 M)
mkdir -p gpurun_out
T=r03z
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_gputest.log | head -1
bash tools/profile_bench.sh ${T}_prof > gpurun_out/${T}_prof.log 2>&1; echo "profile rc=$?"
for k in traffic valu; do cp gpurun_out/${T}_prof_bench_$k.json profiles/r03_bench_100M_$k.json; done
timeout 600 python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; echo "default bench rc=$?"
bash tools/step_timeline.sh ${T} --no-parity > /dev/null 2>&1; echo "timeline rc=$?"
timeout 300 python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --force-sharded --shard-mode octants > gpurun_out/${T}_sharded_world1.json 2> /dev/null; echo "sharded rc=$?"
timeout 1500 python bench.py --virtual-ranks 8 --shard-mode both --verify --steps 2 --warmup 1 > gpurun_out/${T}_config3_virtual8_1B.json 2> gpurun_out/${T}_config3.err; echo "virtual 1B rc=$?"
python - <<'PY'
import json, glob
for f in ['gpurun_out/r03z_bench_default.json', 'gpurun_out/r03z_config3_virtual8_1B.json', 'gpurun_out/r03z_sharded_world1.json']:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get('parity') or {}
        print(f.split('/')[-1], d.get('value'), d.get('ms_per_step'), 'parity', p.get('ok'), p.get('mismatching_nodes'), (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('profile_matches_build'))
    except Exception as e:
        print(f, 'ERR', str(e)[:100])
PY
