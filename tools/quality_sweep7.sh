#!/usr/bin/env bash
# does a warm-up of the step (first round 12 games' worth, doubling per round) keep the larger streamed configurations stable?
set -e
run() { echo "=== $*"; python3 examples/selfplay_train.py --arena 4096 --games 65536 --rounds 16 --max-plies 400 "$@" 2>&1 | grep -v amdgpu.ids | tail -3; }
run --slots 4096 --scale-games 192 --scale-warmup 12
run --slots 8192 --scale-games 384 --scale-warmup 12
run --slots 4096 --scale-games 96 --scale-warmup 12
run --slots 2048 --scale-games 96 --scale-warmup 12
run --slots 16384 --scale-games 768 --scale-warmup 12
