#!/usr/bin/env bash
# streamed replay: the summed update of a step is (slots x scale / slots) = `--scale-games` reference games at EVERY step (no idle tail as in
# a lock-step sub-round), so the step size that is stable is smaller than for sub-rounds of the same size: sweep it
set -e
run() { echo "=== $*"; python3 examples/selfplay_train.py --arena 4096 --games 65536 --rounds 16 --max-plies 400 "$@" 2>&1 | grep -v amdgpu.ids | tail -3; }
run --slots 4096 --scale-games 96
run --slots 4096 --scale-games 48
run --slots 8192 --scale-games 96
run --slots 8192 --scale-games 192
run --slots 2048 --scale-games 48
run --slots 16384 --scale-games 96
