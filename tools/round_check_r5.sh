#!/usr/bin/env bash
# Round 5's evidence, one gpurun call per part (each within the 1 200 s limit):
#   gpurun --timeout 1150 -- 'bash tools/round_check_r5.sh tests r05_v60'     pytest -m gpu (durations), then -m gpu_experimental on the experimental build
#   gpurun --timeout 1150 -- 'bash tools/round_check_r5.sh step r05_v60'      smoke, bench line (with the training round), rocprof stats + per-mode trace summary
#                                                                             + 65 536-lane timeline, PMC traffic, SQ counters, driver-style bench, 2-rank gloo bench
#   gpurun --timeout 1150 -- 'bash tools/round_check_r5.sh more r05_v60'      lanes study + 32 768-lane timelines, soak, training loop four ways (+ delayed), learner benches
#   gpurun --timeout 1150 -- 'bash tools/round_check_r5.sh quality r05_v60'   the quality gate (tools/quality_r04.sh) and the minutes-long soak of the training loop
PART=${1:?tests|step|more|quality}; TAG=${2:?tag}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
if [ "$PART" = tests ]; then
  timeout -k 10 900 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/$TAG/tests_gpu.txt 2>&1; echo "gpu rc=$?"; tail -3 gpurun_out/$TAG/tests_gpu.txt
  timeout -k 10 400 python -m pytest tests -q -m gpu_experimental > gpurun_out/$TAG/tests_gpu_experimental.txt 2>&1; echo "experimental rc=$?"; tail -3 gpurun_out/$TAG/tests_gpu_experimental.txt
elif [ "$PART" = more ]; then
  python tools/lanes_study.py > gpurun_out/$TAG/lanes_study.txt 2>&1; grep " lanes " gpurun_out/$TAG/lanes_study.txt | grep nofork | cut -c1-200
  export BGAMD_NO_BUILD=1
  for m in f32 bf16; do
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG/lanes_32768_trace_$m -- python3 tools/lanes_study.py --only 32768,$m,nofork > gpurun_out/$TAG/lanes_32768_trace_$m.log 2>&1
    python tools/lanes_study.py --timeline gpurun_out/$TAG/lanes_32768_trace_$m > gpurun_out/$TAG/lanes_32768_timeline_$m.txt 2>&1; cat gpurun_out/$TAG/lanes_32768_timeline_$m.txt
  done
  unset BGAMD_NO_BUILD
  python tools/soak.py > gpurun_out/$TAG/soak.txt 2>&1; grep -v amdgpu gpurun_out/$TAG/soak.txt | tail -2 | cut -c1-300
  python tools/train_pipeline.py > gpurun_out/$TAG/train_pipeline.txt 2>&1; grep -v amdgpu gpurun_out/$TAG/train_pipeline.txt | tail -9 | cut -c1-300
  python tools/train_pipeline.py --modes cont,cont_pipe --delay 1 > gpurun_out/$TAG/train_pipeline_delay1.txt 2>&1; grep lanes gpurun_out/$TAG/train_pipeline_delay1.txt | cut -c1-300
  python tools/td_bench.py 512 4096 16384 32768 65536 > gpurun_out/$TAG/td_bench.txt 2>&1; grep "^n=" gpurun_out/$TAG/td_bench.txt | cut -c1-160
  python tools/train_breakdown.py 65536 0 s4096 s2048 s1024 > gpurun_out/$TAG/train_breakdown.txt 2>&1; grep -v amdgpu gpurun_out/$TAG/train_breakdown.txt | cut -c1-200
  find gpurun_out/$TAG -type f \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" \) -delete
elif [ "$PART" = quality ]; then
  bash tools/quality_r04.sh > gpurun_out/$TAG/quality_r04_recipe.txt 2>&1; grep "===\|tdgammonNEW100k" gpurun_out/$TAG/quality_r04_recipe.txt | cut -c1-200
  bash tools/soak_training.sh $TAG
else
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1 && bash tools/sq_counters.sh $TAG > gpurun_out/${TAG}_sq.log 2>&1
  tail -2 gpurun_out/${TAG}_profile.log
  python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_driver_style.json
  python -c "import json; d=json.load(open('gpurun_out/$TAG/bench_driver_style.json')); print('driver-style bench:', d['value'], d['ms_per_step'], d['timed_regions'], d['region_ms'], d['roofline']['frac'], d['roofline']['kernel'], d['training_round'].get('continuous_window_84_steps_replay_beside_the_next_window'))"
  python bench.py --gpus 2 --dist-backend gloo --games 32768 --steps 20 --warmup 5 2>gpurun_out/$TAG/bench_2ranks_gloo.err | tail -1 > gpurun_out/$TAG/bench_2ranks_gloo.json
  python -c "import json; d=json.load(open('gpurun_out/$TAG/bench_2ranks_gloo.json')); print('2 ranks (gloo, one GPU):', d['value'], d['n_gpus'], d['ranks_seen'], d['per_rank_ms_per_step'])"
  find gpurun_out/$TAG -type f \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" \) -delete
fi
du -sh gpurun_out/$TAG
