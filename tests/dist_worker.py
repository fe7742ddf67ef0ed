#!/usr/bin/env python3
"""One rank of the 2-rank shared-GPU rehearsal (tests/test_gpu_round2.py::test_two_rank_rehearsal_on_one_gpu).

Started as a FRESH process (nothing GPU-related is inherited), joins a gloo group on 127.0.0.1, and runs its shard of
 (1) the config-3 path: `lanes` games of a world x lanes env, `steps` greedy steps  -> states / turns of its lanes
 (2) the config-4 path: one self-play round on its shard + the split / all-reduce / apply TD(lambda) replay
     (ONE all-reduce of the 25 601-float update per training step)                    -> its replica's weights
and writes both to <out>/rank<r>.npz."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the HIP runtime starts: backgammon_env/__init__.py says why
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "backgammon-engine_amd")):
    sys.path.insert(0, p)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    lanes, steps, train_lanes = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import backgammon_env as bg
    from backgammon_env.learner import DeviceTDLambdaLearner, play_round
    from backgammon_env.shard import aggregate, shard_for_rank

    w = np.fromfile(os.path.join(ROOT, "tests", "golden", "tdgammonNEW100k.f32"), dtype=np.float32)
    off, stride = shard_for_rank(rank, world, lanes)
    env = bg.VecGame(lanes, seed=4242, lane_offset=off, lane_stride=stride)
    env.load_weights(w)
    env.run_greedy(steps)
    st, tn = env.states().cpu().numpy(), env.turns().cpu().numpy()
    s = env.stats()
    tot, _ = aggregate({k: s[k] for k in ("steps", "games_finished", "p1_wins")}, 0.0)

    off2, stride2 = shard_for_rank(rank, world, train_lanes)
    tenv = bg.VecGame(train_lanes, seed=77, lane_offset=off2, lane_stride=stride2)
    tenv.load_weights(w)
    L = DeviceTDLambdaLearner(w, max_games=train_lanes, alpha=0.1, lam=0.9)
    sq = cnt = 0
    r0 = {}
    for rnd in range(3):                                 # three rounds: each plays with the weights the one before left
        rows, lengths, p1_won = play_round(tenv, max_plies=400, epsilon=0.05)
        # round 0: the whole shard lock-step; round 1: two sub-rounds per rank (every rank runs the same number);
        # round 2: streamed through train_lanes / 4 slots per rank (the ranks' step counts differ: the shorter one joins every collective)
        sub, slots = (0, 0) if rnd == 0 else (train_lanes // 2, 0) if rnd == 1 else (0, train_lanes // 4)
        a, b = L.replay_rows(rows, lengths, p1_won, group=dist.group.WORLD, sub_round=sub, slots=slots,
                             batch_scale=24.0 / (world * (sub or slots or train_lanes)))
        sq, cnt = sq + a, cnt + b
        if rnd == 0:                                     # the lock-step round's log and result: the parent replays both shards' logs on ONE learner
            r0 = dict(r0_rows=rows.cpu().numpy(), r0_lengths=lengths.cpu().numpy(), r0_p1_won=p1_won.cpu().numpy(),
                      r0_theta=L.theta.cpu().numpy())
        tenv.load_weights(L.theta.cpu().numpy())
    np.savez(os.path.join(out, f"rank{rank}.npz"), states=st, turns=tn, theta=L.theta.cpu().numpy(),
             totals=np.array([tot["steps"], tot["games_finished"], tot["p1_wins"]], dtype=np.int64),
             learner=np.array([sq, cnt], dtype=np.float64), lengths=lengths.cpu().numpy(), **r0)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
