# round-3 records of the final code: GPU suite, default bench line (oracle parity inside), rocprofv3 passes of the build and of
# the query path, config 4 with full parity, config 5 at 500 M, 1 B on one GPU, config-3 dress rehearsal (8 virtual ranks)
mkdir -p gpurun_out
T=r03z
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?"
bash tools/profile_bench.sh ${T}_prof > gpurun_out/${T}_prof.log 2>&1; echo "profile rc=$?"
mkdir -p profiles_new; for k in traffic valu; do cp gpurun_out/${T}_prof_bench_$k.json profiles/r03_bench_100M_$k.json; done
timeout 600 python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; echo "default bench rc=$?"
bash tools/step_timeline.sh ${T} --no-parity > /dev/null 2>&1; echo "timeline rc=$?"
timeout 900 python bench.py --query > gpurun_out/${T}_query.json 2> gpurun_out/${T}_query.err; echo "query rc=$?"
bash tools/profile_query.sh ${T} > gpurun_out/${T}_query_prof.log 2>&1; echo "query profile rc=$?"
timeout 600 python bench.py --ecef --points 500000000 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --verify > gpurun_out/${T}_config5_ecef_500M.json 2> gpurun_out/${T}_config5.err; echo "config5 rc=$?"
timeout 900 python bench.py --points 1000000000 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --verify > gpurun_out/${T}_parity_1B_single_gpu.json 2> gpurun_out/${T}_1B.err; echo "1B rc=$?"
timeout 1500 python bench.py --virtual-ranks 8 --shard-mode both --verify --steps 2 --warmup 1 > gpurun_out/${T}_config3_virtual8_1B.json 2> gpurun_out/${T}_config3.err; echo "virtual 1B rc=$?"
timeout 300 python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-parity --force-sharded --shard-mode octants > gpurun_out/${T}_sharded_world1.json 2> /dev/null; echo "sharded rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03z_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get('parity') or {}
        print(f.split('/')[-1], d.get('value'), d.get('ms_per_step'), 'parity', p.get('ok'), p.get('mismatching_nodes'), p.get('nodes'),
              (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('profile_matches_build'))
    except Exception as e:
        print(f, 'ERR', str(e)[:100])
PY
