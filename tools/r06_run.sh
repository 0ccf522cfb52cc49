#!/bin/bash
# Round-6 GPU-box recipe, ONE gpurun call: stages picked by name, in the order given.
#   usage: gpurun -- 'bash tools/r06_run.sh TAG stage [stage ...]'
#   tests     pytest -m gpu                               probes    tools/f64_rate.sh + scatter_probe runs
#   bench     the default line (config 2 + legs 4 / 5)     quick     the default line without the legs (--no-legs)
#   ab        timing-only A/B runs, RUNS="name[:ENV=VAL|libtag] ..." (tools/ab_quick.sh)
#   profile   rocprofv3 passes of the build (tools/profile_bench.sh) -> profiles-ready JSON / CSV in gpurun_out/
#   qprofile  rocprofv3 passes of bench.py --query (tools/profile_query.sh)
#   smoke     __graft_entry__.smoke()
#   pmc       SQ / TA / TCP counter passes over the build (tools/pmc_pass.sh) -> gpurun_out/TAG_pmc.json
TAG=${1:-r06}; shift
mkdir -p gpurun_out
for stage in "$@"; do
  t0=$(date +%s)
  case $stage in
    tests) timeout 900 python -m pytest tests -m gpu -q --tb=line > gpurun_out/${TAG}_gputest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/${TAG}_gputest.log;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.log;;
    probes) bash tools/f64_rate.sh ${TAG} > gpurun_out/${TAG}_f64_rate.log 2>&1; echo "f64_rate rc=$?"; tail -25 gpurun_out/${TAG}_f64_rate.log
            timeout 300 tools/scatter_probe.bin runs 100000000 > gpurun_out/${TAG}_scatter_runs.jsonl 2> gpurun_out/${TAG}_scatter_runs.err; echo "scatter rc=$?"; cat gpurun_out/${TAG}_scatter_runs.jsonl;;
    bench) timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/${TAG}_bench_default.err
           cp bench_detail.json gpurun_out/${TAG}_bench_default_detail.json; tail -c 4200 gpurun_out/${TAG}_bench_default.json | tail -1 | wc -c
           python tools/bench_brief.py gpurun_out/${TAG}_bench_default_detail.json;;
    quick) timeout 600 python bench.py --no-legs > gpurun_out/${TAG}_bench_quick.json 2> gpurun_out/${TAG}_bench_quick.err; echo "quick rc=$?"; tail -3 gpurun_out/${TAG}_bench_quick.err
           cp bench_detail.json gpurun_out/${TAG}_bench_quick_detail.json
           python tools/bench_brief.py gpurun_out/${TAG}_bench_quick_detail.json;;
    ab) TAG=${TAG}_ab bash tools/ab_quick.sh;;
    intensity) timeout 600 python bench.py --only-intensity --intensity-points ${IPOINTS:-100000000} --steps 10 > gpurun_out/${TAG}_intensity.json 2> gpurun_out/${TAG}_intensity.err; echo "intensity rc=$?"; tail -2 gpurun_out/${TAG}_intensity.err
           python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_intensity.json').read().strip().splitlines()[-1])
print('intensity', d['points'], 'colour', d['color_only']['ms_per_step'], '+intensity', d['color_and_intensity']['ms_per_step'], 'cost', d['intensity_cost'], 'parity', d['parity']['ok'], d['parity']['mismatching_nodes'])
print(d['color_only']['kernel_ms_per_step']); print(d['color_and_intensity']['kernel_ms_per_step'])";;
    profile) bash tools/profile_bench.sh ${TAG}_prof > gpurun_out/${TAG}_prof.log 2>&1; echo "profile rc=$?"; tail -5 gpurun_out/${TAG}_prof.log;;
    qprofile) bash tools/profile_query.sh ${TAG} > gpurun_out/${TAG}_qprof.log 2>&1; echo "qprofile rc=$?"; tail -12 gpurun_out/${TAG}_qprof.log;;
    pmc) bash tools/pmc_pass.sh ${TAG} "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU" \
           "SQ_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
           "TA_BUSY_avr TA_FLAT_WAVEFRONTS_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" > gpurun_out/${TAG}_pmc.log 2>&1; echo "pmc rc=$?"; tail -14 gpurun_out/${TAG}_pmc.log;;
    *) echo "unknown stage $stage";;
  esac
  echo "== $stage took $(( $(date +%s) - t0 )) s"
done
