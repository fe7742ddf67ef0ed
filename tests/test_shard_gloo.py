"""The N>1 path on CPU: world_size-2 gloo process group.  Sharding is pure index arithmetic plus one
SUM/MAX reduction of counters and time; the compute stand-in here is the oracle (tests may use it),
keyed by the same global game ids the HIP env uses, so shard results must equal the unsharded run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, games_per_rank, steps, out_dir):
    for p in (ROOT, os.path.join(ROOT, "backgammon-engine_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from backgammon_env.shard import aggregate, shard_for_rank
    from oracle import oracle as O
    off, stride = shard_for_rank(rank, world, games_per_rank)
    fin = cand = 0
    last = np.zeros((games_per_rank, 29), dtype=np.int32)
    for lane in range(games_per_rank):
        snap, f, c, _ = O.lane_run(7, off + lane, stride, steps, 0)
        fin += f
        cand += c
        last[lane] = snap[-1, :29]
    tot, tmax = aggregate({"steps": games_per_rank * steps, "games_finished": fin, "candidates_raw": cand},
                          elapsed_s=1.0 + rank)
    np.save(os.path.join(out_dir, f"last{rank}.npy"), last)
    if rank == 0:
        np.save(os.path.join(out_dir, "tot.npy"), np.array([tot["steps"], tot["games_finished"], tot["candidates_raw"], tmax]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "backgammon-engine_amd"))
    from backgammon_env.shard import global_game_id, shard_for_rank
    from oracle import oracle as O
    world, per, steps = 2, 24, 70
    mp.spawn(_worker, args=(world, _free_port(), per, steps, str(tmp_path)), nprocs=world, join=True)
    tot = np.load(tmp_path / "tot.npy")
    got = np.concatenate([np.load(tmp_path / f"last{r}.npy") for r in range(world)])
    fin = cand = 0
    for lane in range(world * per):
        snap, f, c, _ = O.lane_run(7, lane, world * per, steps, 0)
        assert (snap[-1, :29] == got[lane]).all(), lane
        fin += f
        cand += c
    assert tot[0] == world * per * steps and tot[1] == fin and tot[2] == cand
    assert tot[3] == 2.0                                   # MAX over ranks of the elapsed time
    assert shard_for_rank(1, 2, 65536) == (65536, 131072)
    assert global_game_id(1, 2, 65536, lane=5, episode=3) == 65536 + 5 + 3 * 131072
    with pytest.raises(ValueError):
        shard_for_rank(2, 2, 8)


def _learner_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "backgammon-engine_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from backgammon_env.learner import TDLambdaLearner
    d = np.load(os.path.join(out_dir, "in.npz"))
    G = d["X"].shape[1]
    sl = slice(rank * G // world, (rank + 1) * G // world)
    L = TDLambdaLearner(d["w"], alpha=0.1, lam=0.9)
    Tr = int(d["lengths"][sl].max())                # ranks hold logs of different depth
    L.replay(torch.from_numpy(d["X"][:Tr, sl]), d["lengths"][sl], d["won"][sl])
    np.save(os.path.join(out_dir, f"theta{rank}.npy"), L.theta.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_learner_one_allreduce_per_step_equals_single_process(tmp_path, weights):
    """Configs 4/5: each rank replays ITS shard's games; the per-step update is all-reduced (SUM), so every rank
    ends with the weights a single process would get from all games -- and the ranks never diverge."""
    sys.path.insert(0, os.path.join(ROOT, "backgammon-engine_amd"))
    from backgammon_env.learner import TDLambdaLearner
    from oracle import oracle as O
    G, T = 6, 30
    lanes = [O.lane_run(11, lane, G, T, 0)[0] for lane in range(G)]
    X = np.stack([[O.encode(l[t:t + 1, :28], int(l[t, 28]))[0] for l in lanes] for t in range(T)]).astype(np.float32)
    lengths = np.array([30, 22, 28, 17, 25, 19])
    won = np.array([1, 0, 0, 1, 1, 0])
    np.savez(tmp_path / "in.npz", X=X, lengths=lengths, won=won, w=weights)
    single = TDLambdaLearner(weights, alpha=0.1, lam=0.9)
    single.replay(torch.from_numpy(X), lengths, won)
    mp.spawn(_learner_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    t0, t1 = np.load(tmp_path / "theta0.npy"), np.load(tmp_path / "theta1.npy")
    assert np.array_equal(t0, t1)                                   # identical on every rank
    assert np.abs(t0 - single.theta.numpy()).max() < 1e-6           # == the unsharded replay
    assert np.abs(t0 - weights).max() > 1e-4
