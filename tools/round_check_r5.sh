#!/usr/bin/env bash
# Round 5's evidence, one gpurun call per part (each within the 1 200 s limit):
#   gpurun --timeout 1150 -- 'bash tools/round_check_r5.sh tests r05_v60'     pytest -m gpu (durations), then -m gpu_experimental on the experimental build
#   gpurun --timeout 1150 -- 'bash tools/round_check_r5.sh step r05_v60'      smoke, bench line (with the training round), rocprof stats + per-mode trace summary
#                                                                             + 65 536-lane timeline, PMC traffic, SQ counters, driver-style bench, 2-rank gloo bench
PART=${1:?tests|step}; TAG=${2:?tag}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
if [ "$PART" = tests ]; then
  timeout -k 10 900 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/$TAG/tests_gpu.txt 2>&1; echo "gpu rc=$?"; tail -3 gpurun_out/$TAG/tests_gpu.txt
  timeout -k 10 400 python -m pytest tests -q -m gpu_experimental > gpurun_out/$TAG/tests_gpu_experimental.txt 2>&1; echo "experimental rc=$?"; tail -3 gpurun_out/$TAG/tests_gpu_experimental.txt
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
