#!/usr/bin/env bash
# Round 4: the quality gate of the continuous / pipelined training loop: ~1.05-1.1 M games of epsilon-greedy self-play +
# streamed TD(lambda) through 2 048 slots from the reference's random init, then 8 192 games against the reference's 100k checkpoint.
#   (a) round 3's loop: one game per lane and round, replay after the round
#   (b) the first 3 rounds classic, then continuous self-play with a replay after every window of 84 steps
#   (c) the same with the replay of window w - 1 beside the play of window w (policy one window staler)
# (continuous windows FROM the random init are fragile: profiles/r04_training_quality.txt, tools/quality_r04_diag.sh)
set -e
run() { echo "=== $*"; python3 examples/selfplay_train.py --arena 4096 --games 65536 --max-plies 400 --slots 2048 --scale-games 96 "$@" 2>&1 | grep -v amdgpu.ids | tail -4; }
run --rounds 16
run --rounds 16 --continuous --classic-rounds 3
run --rounds 16 --continuous --classic-rounds 3 --pipeline-rounds
