#!/bin/bash
# LDS / wait counters of the two record downsweeps
PMC_BENCH_ARGS="--no-legs" bash tools/pmc_pass.sh r06i \
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
 "SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
 "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" > gpurun_out/r06i_pmc.log 2>&1
tail -5 gpurun_out/r06i_pmc.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06i_pmc.json'))
for k,v in d['kernels'].items():
    if any(s in k for s in ('downsweep_rec12','downsweep_settle','chain_pass','rank_hist','promote_')):
        print(k[:60], {a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items()})
PY
