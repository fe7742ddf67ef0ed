#!/usr/bin/env bash
# Round 3's evidence in two gpurun calls (each within the 1 200 s limit; the GPU tests run as their own call):
#   gpurun --timeout 1150 -- 'bash tools/round_check_r3.sh step r03_v35'      bench line, rocprof stats, PMC traffic, SQ counters, driver-style bench
#   gpurun --timeout 1150 -- 'bash tools/round_check_r3.sh learner r03_v35'   learner benches, rocprof of a replay, training round, RCCL one-rank step
PART=${1:?step|learner}; TAG=${2:?tag}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
if [ "$PART" = step ]; then
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1 && bash tools/sq_counters.sh $TAG > gpurun_out/${TAG}_sq.log 2>&1
  tail -2 gpurun_out/${TAG}_profile.log
  python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_driver_style.json
  python -c "import json; d=json.load(open('gpurun_out/$TAG/bench_driver_style.json')); print('driver-style bench:', d['value'], d['ms_per_step'], d['timed_regions'], d['region_ms'], d['roofline']['frac'])"
  BGAMD_MFMA_DELTA=1 python bench.py --quick 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_mfma_delta_quick.json
else
  python tools/td_bench.py 512 4096 16384 32768 65536 > gpurun_out/$TAG/td_bench.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/td_stats -- python3 tools/td_bench.py 65536 > gpurun_out/$TAG/td_rocprof.log 2>&1
  grep "^n=" gpurun_out/$TAG/td_bench.txt | cut -c1-120
  python tools/train_breakdown.py 65536 0 4096 2048 s8192 s4096 s2048 s1024 > gpurun_out/$TAG/train_breakdown.txt 2>&1; grep -v amdgpu gpurun_out/$TAG/train_breakdown.txt | cut -c1-200
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stream_stats -- python3 tools/train_breakdown.py 65536 s2048 > gpurun_out/$TAG/stream_rocprof.log 2>&1
  python bench.py --training-round --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_with_training_round.json
  find gpurun_out/$TAG -type f \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" \) -delete
fi
du -sh gpurun_out/$TAG
