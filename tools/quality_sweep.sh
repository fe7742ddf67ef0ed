#!/usr/bin/env bash
# Training quality vs round size: ~1 M games of epsilon-greedy self-play + TD(lambda) from the reference's
# random init, evaluated head to head against the reference's own 100k-episode checkpoint (tdgammonNEW100k) on 8 192 games.
#   gpurun --timeout 1100 -- 'bash tools/quality_sweep.sh > gpurun_out/quality.log 2>&1'
set -e
run() { echo "=== $*"; python3 examples/selfplay_train.py --arena 4096 "$@" 2>&1 | grep -v amdgpu.ids | tail -4; }
run --games 512   --rounds 2048
run --games 4096  --rounds 256
run --games 4096  --rounds 256 --sub-round 512
run --games 65536 --rounds 16  --max-plies 400
run --games 65536 --rounds 16  --max-plies 400 --sub-round 512
run --games 65536 --rounds 16  --max-plies 400 --sub-round 4096
