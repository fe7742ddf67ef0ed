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
    from backgammon_env.shard import aggregate, census, shard_for_rank
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
    seen, per_rank = census(1.0 + rank)
    np.save(os.path.join(out_dir, f"last{rank}.npy"), last)
    np.save(os.path.join(out_dir, f"census{rank}.npy"), np.array([seen] + per_rank))
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
    for r in range(world):                                 # ranks_seen / per-rank times of the bench line: the same on every rank
        assert list(np.load(tmp_path / f"census{r}.npy")) == [2.0, 1.0, 2.0]
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


# ---- bench.py started plainly with --gpus N: the parent launches its own ranks ------------------------------------

_STUB = """
import json, os, sys, time
r, out, mode = int(os.environ["RANK"]), sys.argv[1], sys.argv[2]
json.dump({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")} | {"argv": sys.argv[3:], "pid": os.getpid()},
          open(os.path.join(out, "rank%d.json" % r), "w"))
print("warning from rank %d" % r, file=sys.stderr)
if mode == "ok":
    print(json.dumps({"metric": "stub", "n_gpus": int(os.environ["WORLD_SIZE"]), "from_rank": r}))
elif mode == "fail1":
    if r == 1:
        time.sleep(0.3); sys.exit(3)
    time.sleep(120)
elif mode == "hang":
    time.sleep(120)
"""


def _bench():
    import importlib
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_bench_launcher_starts_its_own_ranks_relays_one_line_and_the_exit_code(tmp_path):
    """launch_ranks (what `python bench.py --gpus N` does without WORLD_SIZE): N children with the rendezvous variables of
    torch.distributed.run on 127.0.0.1, rank 0's stdout relayed alone, a failing rank's code returned and the others stopped, a watchdog."""
    import io
    import json
    import time
    bench = _bench()
    assert bench.torch is None and bench.dist is None          # importing bench.py (what the launching parent does) pulls in neither torch nor HIP
    stub = tmp_path / "stub.py"
    stub.write_text(_STUB)

    def run(mode, n=2, timeout_s=60.0, extra=()):
        out, err = io.StringIO(), io.StringIO()
        t0 = time.monotonic()
        rc = bench.launch_ranks(n, [], child_cmd=[sys.executable, str(stub), str(tmp_path), mode, *extra], timeout_s=timeout_s, out=out, err=err)
        return rc, out.getvalue(), err.getvalue(), time.monotonic() - t0

    rc, out, err, _ = run("ok", n=3, extra=("--steps", "7"))
    assert rc == 0
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "stub", "n_gpus": 3, "from_rank": 0}        # ONE line: rank 0's
    assert "[rank 1] " in err and "[rank 2] " in err and '"from_rank": 2' in err                           # the others' output goes to stderr, tagged
    envs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["LOCAL_WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and envs[0]["argv"] == ["--steps", "7"]
    assert len({e["pid"] for e in envs} | {os.getpid()}) == 4                                              # children, not an exec of this process

    rc, out, err, dt = run("fail1")
    assert rc == 3 and dt < 30 and "rank 1 exited with 3" in err and out.strip() == ""                     # the sleeping rank 0 was stopped, not waited for
    pid0 = json.load(open(tmp_path / "rank0.json"))["pid"]
    with pytest.raises(ProcessLookupError):
        os.kill(pid0, 0)

    rc, out, err, dt = run("hang", timeout_s=1.0)
    assert rc == 124 and dt < 30 and "watchdog" in err


def test_bench_main_routes_gpus_n_to_the_launcher(monkeypatch):
    bench = _bench()
    seen = {}

    def fake(n, argv, timeout_s=0, **kw):
        seen.update(n=n, argv=list(argv), timeout_s=timeout_s)
        return 7
    monkeypatch.setattr(bench, "launch_ranks", fake)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--dist-backend", "gloo", "--steps", "20", "--launch-timeout", "99"])
    assert e.value.code == 7 and seen == {"n": 2, "argv": ["--gpus", "2", "--dist-backend", "gloo", "--steps", "20", "--launch-timeout", "99"], "timeout_s": 99.0}
    monkeypatch.setenv("WORLD_SIZE", "3")                            # an outside launcher with another world size: refused, loudly
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"])
    assert "WORLD_SIZE=3" in str(e.value.code)
    a = bench.parse_args([])
    assert a.gpus == 1 and a.training_round and not bench.parse_args(["--no-training-round"]).training_round
