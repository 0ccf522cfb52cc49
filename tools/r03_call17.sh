mkdir -p gpurun_out
for th in 15 31 47 15 31; do
  PCV_HIP_LIBRARY=exp PCV_H2D_TRACE=1 PCV_H2D_THREADS=$th python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/r03r_e2e_t$th.json 2> gpurun_out/r03r_e2e_t$th.err
  echo "threads $th"; grep "pcv h2d" gpurun_out/r03r_e2e_t$th.err | tail -6
  python -c "
import json; d=json.loads(open('gpurun_out/r03r_e2e_t$th.json').read().strip().splitlines()[-1]); e=d['end_to_end']; print(e['h2d_plus_build_ms'], e['d2h_overlapped_with_file_writes_tmpfs_ms'], e['Mpoints_per_s_incl_files'], e['from_ply_file']['read_upload_decode_build_ms'], e['from_ply_file']['d2h_overlapped_with_file_writes_tmpfs_ms'], e['from_ply_file']['Mpoints_per_s_incl_files'])"
done
