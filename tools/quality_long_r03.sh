#!/usr/bin/env bash
# Round 3: what the throughput buys -- LONGER runs of the recipe of DESIGN §7 (65 536-game rounds streamed through 2 048 slots at 96 / 2 048,
# eps 0.1 -> 0 linearly, the reference's alpha / lambda schedule by episode count), then 8 192 games against the reference's 100k-episode checkpoint.
set -e
run() { echo "=== $*"; SECONDS=0; python3 examples/selfplay_train.py --arena 4096 --games 65536 --max-plies 400 --slots 2048 --scale-games 96 "$@" 2>&1 | grep -v amdgpu.ids | tail -4; echo "$SECONDS s wall"; }
run --rounds 16
run --rounds 64
run --rounds 256
