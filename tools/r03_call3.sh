# round-3 call 3: GPU suite (new tests), kernel timeline of one step, rocprofv3 stats / PMC passes of the current code
mkdir -p gpurun_out
T=r03c
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/${T}_gputest.log
bash tools/step_timeline.sh ${T} --no-parity > /dev/null 2>&1; echo "timeline rc=$?"
cat gpurun_out/${T}_timeline.txt
bash tools/profile_bench.sh ${T}_prof > gpurun_out/${T}_prof.log 2>&1; echo "profile rc=$?"
tail -30 gpurun_out/${T}_prof.log
