cd $GRAFT_REPO_ROOT
for cfg in "1024 4" "2048 4" "2048 6" "1536 6" "2048 8"; do set -- $cfg
  touch point_cloud_viewer_amd/csrc/pcv_sort.hip
  make -s -C point_cloud_viewer_amd/csrc EXTRA="-DPCV_SORT_GROUPS=$1 -DPCV_KEYS_WAVES=$2" 2>&1 | grep -E "error|spill" | head -3
  echo "cfg groups=$1 waves=$2"
  timeout 300 python bench.py --no-cpu-baseline --no-e2e 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(d['value'], d['stage_ms']['sort_keys'], d['stage_ms']['sort_records'], k['upsweep_kernel<u32>'], k['downsweep_kernel<u32>'], k['downsweep_rec_kernel'])"
done
