# round-3 call 6: creator / writer threads for the node files, pread with 15 host threads: GPU suite + the end-to-end legs
mkdir -p gpurun_out
T=r03f
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${T}_gputest.log
timeout 600 python bench.py --no-parity --no-cpu-baseline > gpurun_out/${T}_bench_e2e.json 2> gpurun_out/${T}_bench_e2e.err; echo "bench rc=$?"
for w in 6 14 30; do for c in 1 2 4; do
  PCV_HIP_LIBRARY=exp PCV_WRITER_THREADS=$w PCV_CREATOR_THREADS=$c timeout 300 python bench.py --no-parity --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/${T}_e2e_w${w}_c${c}.json 2> /dev/null
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03f_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        e = d['end_to_end']
        print(f.split('/')[-1], d['value'], 'arrays->files', e['Mpoints_per_s_incl_files'], 'write ms', e['d2h_overlapped_with_file_writes_tmpfs_ms'], 'h2d+build', e['h2d_plus_build_ms'],
              '| ply', e['from_ply_file']['Mpoints_per_s_incl_files'], e['from_ply_file']['read_upload_decode_build_ms'], e['from_ply_file']['d2h_overlapped_with_file_writes_tmpfs_ms'])
    except Exception as ex:
        print(f, 'ERR', ex)
PY
