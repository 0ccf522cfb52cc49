# round-3 call 35: parity at sizes between the recorded ones (250 M: predicted tree beyond the LDS rank map; 40 M; 10 M)
mkdir -p gpurun_out
T=r03C
for P in 250000000 40000000 10000000; do
  timeout 900 python bench.py --points $P --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --verify > gpurun_out/${T}_parity_$P.json 2> gpurun_out/${T}_parity_$P.err; echo "$P rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03C_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d.get('parity') or {}
        b = d.get('build_info') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'parity', p.get('ok'), p.get('mismatching_nodes'), p.get('nodes'), {k: b.get(k) for k in ('single_chain', 'predicted_nodes', 'record_bytes', 'continued_points', 'replayed_points')})
    except Exception as e:
        print(f, 'ERR', e)
PY
