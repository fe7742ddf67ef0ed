#!/usr/bin/env bash
# Round 4's evidence, one gpurun call per part (each within the 1 200 s limit):
#   gpurun --timeout 1150 -- 'bash tools/round_check_r4.sh tests r04_v40'     pytest -m gpu (durations), then -m gpu_experimental on the experimental build
#   gpurun --timeout 1150 -- 'bash tools/round_check_r4.sh step r04_v40'      bench line, rocprof stats, PMC traffic, SQ counters, driver-style bench
#   gpurun --timeout 1150 -- 'bash tools/round_check_r4.sh learner r04_v40'   training-round A/B (sequential / pipelined / continuous), rocprof of both halves
#                                                                             alone and side by side, the in-library collective, learner benches
PART=${1:?tests|step|learner}; TAG=${2:?tag}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
if [ "$PART" = tests ]; then
  timeout -k 10 900 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/$TAG/tests_gpu.txt 2>&1; echo "gpu rc=$?"; tail -3 gpurun_out/$TAG/tests_gpu.txt
  timeout -k 10 250 python -m pytest tests -q -m gpu_experimental > gpurun_out/$TAG/tests_gpu_experimental.txt 2>&1; echo "experimental rc=$?"; tail -3 gpurun_out/$TAG/tests_gpu_experimental.txt
elif [ "$PART" = step ]; then
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1 && bash tools/sq_counters.sh $TAG > gpurun_out/${TAG}_sq.log 2>&1
  tail -2 gpurun_out/${TAG}_profile.log
  python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_driver_style.json
  python -c "import json; d=json.load(open('gpurun_out/$TAG/bench_driver_style.json')); print('driver-style bench:', d['value'], d['ms_per_step'], d['timed_regions'], d['region_ms'], d['roofline']['frac'], d['roofline']['kernel'])"
else
  python tools/train_pipeline.py > gpurun_out/$TAG/train_pipeline.txt 2>&1; grep -v amdgpu gpurun_out/$TAG/train_pipeline.txt | tail -9
  export BGAMD_NO_BUILD=1
  for m in cont cont_pipe; do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/pipe_$m -- python3 tools/train_pipeline.py --modes $m --rounds 4 --warm 2 > gpurun_out/$TAG/pipe_${m}_rocprof.log 2>&1
  done
  unset BGAMD_NO_BUILD
  python tools/pipeline_slowdown.py gpurun_out/$TAG/pipe_cont gpurun_out/$TAG/pipe_cont_pipe > gpurun_out/$TAG/pipeline_kernel_slowdown.txt; cat gpurun_out/$TAG/pipeline_kernel_slowdown.txt
  python tools/train_pipeline.py --modes cont,cont_pipe --delay 1 > gpurun_out/$TAG/train_pipeline_delay1.txt 2>&1; grep lanes gpurun_out/$TAG/train_pipeline_delay1.txt
  python tools/lanes_study.py > gpurun_out/$TAG/lanes_study.txt 2>&1; grep lanes gpurun_out/$TAG/lanes_study.txt | cut -c1-250
  export BGAMD_NO_BUILD=1
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG/lanes_32768_trace -- python3 tools/lanes_study.py --only 32768,f32,nofork > gpurun_out/$TAG/lanes_32768_trace.log 2>&1
  python tools/lanes_study.py --timeline gpurun_out/$TAG/lanes_32768_trace > gpurun_out/$TAG/lanes_32768_timeline.txt 2>&1; cat gpurun_out/$TAG/lanes_32768_timeline.txt
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG/lanes_32768_trace_bf16 -- python3 tools/lanes_study.py --only 32768,bf16,nofork > gpurun_out/$TAG/lanes_32768_trace_bf16.log 2>&1
  python tools/lanes_study.py --timeline gpurun_out/$TAG/lanes_32768_trace_bf16 > gpurun_out/$TAG/lanes_32768_timeline_bf16.txt 2>&1; cat gpurun_out/$TAG/lanes_32768_timeline_bf16.txt
  unset BGAMD_NO_BUILD
  python tools/train_dist_step.py 16384 256 1024 2048 > gpurun_out/$TAG/train_dist_step.txt 2>&1; grep "slots," gpurun_out/$TAG/train_dist_step.txt | cut -c1-400
  python tools/td_bench.py 512 4096 16384 32768 65536 > gpurun_out/$TAG/td_bench.txt 2>&1
  python tools/train_breakdown.py 65536 0 s4096 s2048 s1024 > gpurun_out/$TAG/train_breakdown.txt 2>&1; grep -v amdgpu gpurun_out/$TAG/train_breakdown.txt | cut -c1-200
  python bench.py --training-round --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_with_training_round.json
  python -c "import json; print(json.dumps(json.load(open('gpurun_out/$TAG/bench_with_training_round.json'))['training_round'], indent=1))"
  find gpurun_out/$TAG -type f \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" \) -delete
fi
du -sh gpurun_out/$TAG
