#!/usr/bin/env bash
# Round 4: the quality gate of the delayed update (bgamd_td_set_delay(1): every update applied one step late, a training step = one launch) on the
# recipe of tools/quality_r04.sh: the first 3 rounds classic AND exact, then windows with the delayed update; three seeds
run() { echo "=== $*"; python3 examples/selfplay_train.py --arena 4096 --games 65536 --max-plies 400 --slots 2048 --scale-games 96 "$@" 2>&1 | grep -v amdgpu.ids | grep "games,\|round 16\|tdgammon"; }
run --rounds 16 --classic-rounds 3 --update-delay 1
run --rounds 16 --continuous --classic-rounds 3 --pipeline-rounds --update-delay 1
run --rounds 16 --continuous --classic-rounds 3 --pipeline-rounds --update-delay 1 --seed 2
run --rounds 16 --continuous --classic-rounds 3 --pipeline-rounds --update-delay 1 --seed 3
run --rounds 16 --continuous --classic-rounds 3 --update-delay 1
