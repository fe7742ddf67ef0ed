#!/usr/bin/env bash
# Soak of the recommended training loop (round 5): ~1 500 pipelined windows of continuous self-play at 65 536 lanes (~98 M games, ~8 G turns replayed) in one process --
# the ring log wraps ~120 times, the learner's thread and stream run beside the env for minutes -- then the arena.  Watches: no games dropped, no error flag, device memory flat.
#   gpurun --timeout 1100 -- 'bash tools/soak_training.sh r05_v62'
TAG=${1:?tag}
mkdir -p gpurun_out/$TAG
( while true; do rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -1; sleep 20; done ) > gpurun_out/$TAG/soak_training_vram.txt 2>&1 &
MON=$!
timeout -k 10 900 python3 examples/selfplay_train.py --arena 4096 --games 65536 --rounds ${2:-1500} --max-plies 400 --slots 2048 --scale-games 96 --continuous --classic-rounds 3 --pipeline-rounds --schedule-div 16 > gpurun_out/$TAG/soak_training.txt 2>&1
echo "rc=$?"
kill $MON
grep -v amdgpu gpurun_out/$TAG/soak_training.txt | awk 'NR<=3 || NR%100==0' | cut -c1-220
grep -v amdgpu gpurun_out/$TAG/soak_training.txt | tail -4 | cut -c1-300
awk 'NR==1 || NR%4==0' gpurun_out/$TAG/soak_training_vram.txt | head -20
