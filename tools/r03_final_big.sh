# config 5 at 500 M and 1 B points on one GPU with oracle parity, on the final build
mkdir -p gpurun_out
T=r03z
timeout 600 python bench.py --ecef --points 500000000 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --verify > gpurun_out/${T}_config5_ecef_500M.json 2> gpurun_out/${T}_config5.err; echo "config5 rc=$?"
timeout 900 python bench.py --points 1000000000 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --verify > gpurun_out/${T}_parity_1B_single_gpu.json 2> gpurun_out/${T}_1B.err; echo "1B rc=$?"
python - <<'PY'
import json
for f in ['gpurun_out/r03z_config5_ecef_500M.json', 'gpurun_out/r03z_parity_1B_single_gpu.json']:
    d = json.loads(open(f).read().strip().splitlines()[-1]); p = d.get('parity') or {}
    print(f.split('/')[-1], d.get('value'), d.get('ms_per_step'), 'parity', p.get('ok'), p.get('mismatching_nodes'), p.get('nodes'))
PY
