#!/usr/bin/env bash
# streamed replay (k slots replay the round's games one after another) against sub-rounds of the same size: 1.05 M games from a random net,
# head to head against the reference's 100k-episode checkpoint
set -e
run() { echo "=== $*"; python3 examples/selfplay_train.py --arena 4096 --games 65536 --rounds 16 --max-plies 400 "$@" 2>&1 | grep -v amdgpu.ids | tail -3; }
run --sub-round 2048 --scale-games 96
run --slots 1024 --scale-games 48
run --slots 2048 --scale-games 96
run --slots 4096 --scale-games 192
run --slots 8192 --scale-games 384
run --slots 4096 --scale-games 192 --precision bf16
