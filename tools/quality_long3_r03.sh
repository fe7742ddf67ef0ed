#!/usr/bin/env bash
# Round 3: longer runs with the reference's alpha / lambda schedule stretched (evaluated at games_done // D): is the decline of quality_long_r03.sh the schedule's floors?
set -e
run() { echo "=== $*"; SECONDS=0; python3 examples/selfplay_train.py --arena 4096 --games 65536 --max-plies 400 --slots 2048 --scale-games 96 "$@" 2>&1 | grep -v amdgpu.ids | tail -3; echo "$SECONDS s wall"; }
run --rounds 64 --schedule-div 4
run --rounds 64 --schedule-div 16
run --rounds 64 --schedule-div 64
run --rounds 256 --schedule-div 16
run --rounds 256 --schedule-div 64
