#!/usr/bin/env bash
# Round 3: the quality runs behind DESIGN §7 on the round-3 learner build (pipelined mid-size trace pass, float4 reduce): ~1.05 M games of
# epsilon-greedy self-play + streamed TD(lambda) from the reference's random init, then 8 192 games against the reference's 100k checkpoint.
set -e
run() { echo "=== $*"; python3 examples/selfplay_train.py --arena 4096 --games 65536 --rounds 16 --max-plies 400 "$@" 2>&1 | grep -v amdgpu.ids | tail -3; }
run --slots 2048 --scale-games 96
run --slots 1024 --scale-games 48
run --slots 2048 --scale-games 96 --scale-warmup 12
