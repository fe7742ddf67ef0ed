#!/usr/bin/env bash
set -e
run() { echo "=== $*"; python3 examples/selfplay_train.py --arena 4096 --games 65536 --rounds 16 --max-plies 400 "$@" 2>&1 | grep -v amdgpu.ids | tail -3; }
run --sub-round 8192 --scale-games 384
run --sub-round 16384 --scale-games 768
run --scale-games 3072
run --sub-round 2048 --scale-games 96
run --sub-round 4096 --scale-games 192
run --sub-round 4096 --scale-games 192 --precision bf16
