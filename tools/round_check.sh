#!/usr/bin/env bash
# Everything the round's evidence comes from, in one gpurun call:  gpurun --timeout 1150 -- 'bash tools/round_check.sh r02_v26'
TAG=${1:?tag}
mkdir -p gpurun_out/$TAG
python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.log 2>&1; tail -3 gpurun_out/$TAG/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1 && bash tools/sq_counters.sh $TAG > gpurun_out/${TAG}_sq.log 2>&1
tail -1 gpurun_out/${TAG}_profile.log
python tools/td_bench.py 512 4096 16384 32768 65536 > gpurun_out/$TAG/td_bench.txt 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/td_stats -- python3 tools/td_bench.py 65536 > gpurun_out/$TAG/td_rocprof.log 2>&1
grep "^n=" gpurun_out/$TAG/td_bench.txt | cut -c1-120
python tools/train_breakdown.py 65536 0 4096 2048 s8192 s4096 s2048 s1024 > gpurun_out/$TAG/train_breakdown.txt 2>&1; grep -v amdgpu gpurun_out/$TAG/train_breakdown.txt | cut -c1-200
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_driver_style.json
python bench.py --training-round --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/$TAG/bench_with_training_round.json
python -c "import json; d=json.load(open('gpurun_out/$TAG/bench_driver_style.json')); print('driver-style bench:', d['value'], d['ms_per_step'], d['timed_regions'], d['region_ms'], d['roofline']['frac'])"
