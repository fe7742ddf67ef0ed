#!/usr/bin/env bash
# Round 3: why do LONGER runs of the §7 recipe get worse (54.0 % after 1.05 M games, 50.7 % after 4.2 M, 40.4 % after 16.8 M)?  Variants at 4.2 M games.
set -e
run() { echo "=== $*"; SECONDS=0; python3 examples/selfplay_train.py --arena 4096 --games 65536 --max-plies 400 --slots 2048 --rounds 64 "$@" 2>&1 | grep -v amdgpu.ids | tail -3; echo "$SECONDS s wall"; }
run --scale-games 96 --eps 0.1
run --scale-games 96 --eps-end 0.05
run --scale-games 48
run --scale-games 24
run --scale-games 96 --lam 0.0
run --scale-games 96 --max-plies 800
