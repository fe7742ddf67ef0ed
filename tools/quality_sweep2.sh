#!/usr/bin/env bash
# second sweep: can larger sub-rounds keep the quality with a larger step (linear scaling of the summed update)?
set -e
run() { echo "=== $*"; python3 examples/selfplay_train.py --arena 4096 --games 65536 --rounds 16 --max-plies 400 "$@" 2>&1 | grep -v amdgpu.ids | tail -2; }
run --sub-round 4096 --scale-games 96
run --sub-round 4096 --scale-games 192
run --sub-round 2048 --scale-games 48
run --sub-round 2048 --scale-games 96
run --sub-round 1024 --scale-games 24
run --sub-round 1024 --scale-games 48
run --sub-round 256 --scale-games 24
