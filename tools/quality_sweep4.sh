#!/usr/bin/env bash
# throughput of the recommended training configurations on the r02_v26 learner (quality as in sweeps 1-3)
set -e
run() { echo "=== $*"; python3 examples/selfplay_train.py --arena 4096 "$@" 2>&1 | grep -v amdgpu.ids | tail -3; }
run --games 512 --rounds 2048
run --games 65536 --rounds 16 --max-plies 400 --sub-round 512
run --games 65536 --rounds 16 --max-plies 400 --sub-round 2048 --scale-games 96
run --games 65536 --rounds 16 --max-plies 400 --sub-round 4096 --scale-games 192
run --games 65536 --rounds 16 --max-plies 400 --sub-round 4096 --scale-games 192 --precision bf16
