"""Round-5 GPU tests (all through the C ABI):
  * bench.py started PLAINLY with --gpus 2 launches its own two ranks (gloo rehearsal on the one GPU): one line, ranks_seen 2, the
    training round of configs 4/5 with its per-step all-reduce, replicas bit-identical;
  * the launch structure the bench times, EVERY lane against the oracle: steps k and k + 1 INSIDE run_greedy (roots + root pass + apply
    inside boundary_kernel<true>) at 65 536 lanes f32 and 32 768 lanes bf16;
  * continuous self-play at 65 536 lanes: every game of the ring log is the game play_round(episode=j) plays;
  * reads past the rows a step produced return zeros (four arenas), evaluate_incremental's rows are listed linearly, the ring
    log refuses steps that would break its game table, weights outside the f16 hi + lo range are refused at load."""
import ctypes
import json
import multiprocessing as mp
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from test_gpu_parity import _np

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def bg():
    import backgammon_env
    return backgammon_env


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


# ---- bench.py --gpus N, started plainly ---------------------------------------------------------------------

def test_bench_started_plainly_with_two_ranks():
    """`python bench.py --gpus 2 --dist-backend gloo --games 32768 --steps 20 --warmup 5`, no launcher, no WORLD_SIZE: the parent starts the
    two ranks itself (before touching the GPU), relays rank 0's ONE line and the ranks' exit code.  The line says n_gpus 2, ranks_seen 2,
    counts both ranks' env steps, carries per-rank times and the training round (per-step all-reduce over both ranks; replicas identical)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--games", "32768",
                          "--steps", "20", "--warmup", "5", "--training-round-timeout", "600"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert d["launched_by"] == "bench.py launch_ranks" and d["dist_backend"] == "gloo"
    pr = d["per_rank_ms_per_step"]
    assert len(pr["by_rank"]) == 2 and pr["min"] <= pr["max"] and abs(pr["max"] - d["ms_per_step"]) < 1e-3
    # value = both ranks' env steps (every lane live: auto-reset) over the slowest rank's region
    assert abs(d["value"] - 2 * 32768 * 20 / (d["ms_per_step"] * 20 * 1e-3)) < 1e-3 * d["value"]
    assert d["value"] > 5e7 and "cpu_baseline" not in d and "roofline" in d
    tr = d["training_round"]
    # (ranks sharing a GPU rehearse the training round on 4 096 games each, once through: the route, not a measurement)
    assert "error" not in tr and tr["ranks"] == 2 and tr["games"] == 8192 and tr["replicas_identical"] is True and "rehearsal" in tr
    assert "all_reduce" in tr["collective"] and tr["turns"] > 4e5
    for k in ("lockstep_whole_round", "streamed_2048_slots", "continuous_window_84_steps", "continuous_window_84_steps_replay_beside_the_next_window"):
        assert tr[k]["round_turns_per_s"] > 1e4, k
    assert tr["lockstep_whole_round"]["weights_checksum"] != tr["streamed_2048_slots"]["weights_checksum"]
    print("bench.py --gpus 2 (gloo, two ranks on one GPU), started plainly: %.1f M env steps/s, per-rank ms/step %s; training round: %s; %.0f s"
          % (d["value"] / 1e6, pr["by_rank"], {k: v["round_turns_per_s"] for k, v in tr.items() if isinstance(v, dict)}, time.time() - t0))


def test_bench_rccl_route_rehearsed_on_one_rank():
    """BENCH_FORCE_DIST=1: the bench on ONE rank through everything the multi-GPU run does -- an RCCL process group (backend nccl), the census,
    the barriers around the timed region, and a training round whose every training step is all-reduced by the LIBRARY on the learner's own
    RCCL communicator (bgamd_td_replay_allreduce), lock-step, streamed, and from the learner's side stream and host thread beside an env at
    play.  A world of one has nothing to move; what is exercised is every call more ranks make."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--games", "16384", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    tr = d["training_round"]
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and "error" not in tr, tr
    assert tr["collective"].startswith("in-library ncclAllReduce") and tr["collective_note"] is None and tr["replicas_identical"] is True
    for k in ("lockstep_whole_round", "streamed_2048_slots", "continuous_window_84_steps", "continuous_window_84_steps_replay_beside_the_next_window"):
        assert tr[k]["round_turns_per_s"] > 1e6, k
    print("bench.py, RCCL route on one rank: %.1f M env steps/s; training round through the in-library all-reduce: %s"
          % (d["value"] / 1e6, {k: v["round_turns_per_s"] for k, v in tr.items() if isinstance(v, dict)}))


# ---- the launch structure the bench times, every lane against the oracle --------------------------------------

@pytest.mark.parametrize("n,mode,tol,ks", [(65536, "f32", 1e-5, (13, 39)), (32768, "bf16", 5e-3, (24,))])
def test_every_lane_of_steps_inside_a_run_vs_oracle(bg, golden_dir, weights, n, mode, tol, ks):
    """test_every_lane_of_65536_vs_oracle checks steps issued one by one (roots_kernel + root_hidden_resident_kernel + apply_kernel).  The
    bench times run_greedy(K): between two steps of a run ONE launch (boundary_kernel<true>) applies step t and produces the roots and the
    value net's root pass of step t + 1.  Here three envs of the same seed run k - 1, k and k + 1 steps in ONE call each: step k of the
    second has its roots and root pass from the fused launch, and in the third env step k is APPLIED by the fused launch too (its step
    k + 1 starts from it).  Every live lane: the state after k steps is an oracle afterstate of the state after k - 1 steps under the
    step's dice, value-optimal within the bound; likewise k + 1 from k."""
    import full_lane_worker as W
    prec = {"f32": bg.F32, "bf16": bg.BF16}[mode]
    workers = min(16, os.cpu_count() or 1)
    wpath = os.path.join(golden_dir, "tdgammonNEW100k.f32")
    with mp.get_context("spawn").Pool(workers, initializer=W.init, initargs=(wpath,)) as pool:
        for k in ks:
            snaps = []
            for steps in (k - 1, k, k + 1):
                env = bg.VecGame(n, seed=777 + n)
                env.load_weights(weights)
                env.run_greedy(steps, auto_reset=False, precision=prec)
                if mode == "f32" and steps > 1:                 # (the dense modes have no root pass; they take the fused boundary launch all the same)
                    assert env.kernel_choice()["root"] == "inside boundary_kernel<true>"
                snaps.append((_np(env.states()), _np(env.turns()), _np(env.dice()), (_np(env.flags()) & 4) == 0))
                assert env.stats()["error_flags"] == 0
                del env
            for a, b in ((0, 1), (1, 2)):
                pre, pt, _, live = snaps[a]
                post, _, dice, _ = snaps[b]
                idx = np.nonzero(live)[0]
                assert len(idx) > 0.95 * n
                res = pool.map(W.check, [(pre[c], pt[c], dice[c], post[c], tol) for c in np.array_split(idx, workers * 8) if len(c)])
                bad = [r for r in res if r[0] != "OK"]
                assert not bad, bad[0]
                print("%s, %d lanes, step %d of run_greedy(%d): %d live lanes, %d oracle afterstates; max |value - best| %.3g; %d lanes not the "
                      "fp64 first best index" % (mode, n, k + a, k + a, len(idx), sum(r[4] for r in res), max(r[1] for r in res),
                                                 sum(r[2] for r in res)), flush=True)


# ---- continuous self-play at the headline size --------------------------------------------------------------------------------------

def test_continuous_selfplay_ring_log_at_65536_lanes(bg, weights):
    """test_continuous_selfplay_ring_log_holds_the_games_play_round_plays at BASELINE's size: 65 536 lanes, two windows of 84 steps into a
    ring of 256 slots, fixed weights, epsilon 0.05.  Every game that ended -- the lanes' first games and the second games of the lanes
    whose first was short -- is, turn for turn, length and winner, the game play_round(episode=0 / 1) logs for that lane (compared on the
    device)."""
    from backgammon_env.learner import ContinuousSelfPlay, play_round
    n, R = 65536, 256
    env = bg.VecGame(n, seed=99)
    env.load_weights(weights)
    sp = ContinuousSelfPlay(env, ring_steps=R)
    tabs = []
    for _ in range(2):
        sp.play(84, epsilon=0.05)
        tabs.append(sp.finished())
        assert sp.dropped == 0
    lane, start, length, won = [torch.cat([t[i] for t in tabs]) for i in range(4)]
    G = int(lane.numel())
    assert G > 0.8 * n and env.stats()["error_flags"] == 0
    # which game of its lane each one is: the games of a lane end in table order
    order = torch.argsort(lane.long(), stable=True)
    ls = lane[order].long()
    first_of_lane = torch.ones(G, dtype=torch.bool, device=lane.device)
    first_of_lane[1:] = ls[1:] != ls[:-1]
    pos = torch.arange(G, device=lane.device)
    seg_start = torch.cummax(torch.where(first_of_lane, pos, torch.zeros_like(pos)), 0).values
    epi = torch.empty(G, dtype=torch.long, device=lane.device)
    epi[order] = pos - seg_start
    assert int(epi.max().item()) >= 1                                # some lanes finished two games
    env2 = bg.VecGame(n, seed=99)
    env2.load_weights(weights)
    checked = 0
    for e in range(int(epi.max().item()) + 1):
        rows, lengths, p1 = play_round(env2, max_plies=600, epsilon=0.05, episode=e)
        sel = torch.nonzero(epi == e).flatten()
        ln, st, lg, wn = lane[sel].long(), start[sel].long(), length[sel].long(), won[sel]
        assert torch.equal(lengths[ln].long(), lg) and torch.equal(p1[ln], wn), e
        T = int(lg.max().item())
        k = torch.arange(T, device=lane.device)[:, None]
        ring = sp.rows[(st[None, :] + k) % R, ln[None, :].expand(T, len(sel))]           # [T, games, 8]
        flat = rows[:T][:, ln] if rows.shape[0] >= T else None
        assert flat is not None
        mask = (k < lg[None, :])[:, :, None]
        assert torch.equal(torch.where(mask, ring, torch.zeros_like(ring)), torch.where(mask, flat, torch.zeros_like(flat))), e
        checked += len(sel)
    assert checked == G
    print(f"continuous self-play at {n} lanes: {G} games in 168 steps ({int((epi > 0).sum().item())} second games) equal play_round's, turn for turn")
    sp.close()


# ---- ADVICE r4 --------------------------------------------------------------------------------------------------------------------

def test_rows_past_the_steps_rows_read_as_zeros_and_incremental_rows_are_linear(bg, O, weights):
    """bgamd_env_unique_rows_read lists the four arenas one after the other; a C-ABI caller asking for more rows than the step produced
    must get zeros, not reads beyond the arenas (the last arena is clamped like the others), and after bgamd_evaluate_incremental --
    which writes its rows linearly -- the list is those rows, not a remap through the last step's arena counters."""
    from backgammon_env import _capi
    lib = _capi.load()
    n = 2048
    env = bg.VecGame(n, seed=3)
    env.load_weights(weights)
    env.run_greedy(9)
    env.step_greedy()
    info, st, val = env.unique_rows()
    u = int(info.shape[0])
    cap = n * 64                                                       # the env's default arena is larger than this; far past the rows produced
    assert u < cap
    stx = torch.full((cap, 28), -7, dtype=torch.int32, device="cuda")
    vx = torch.full((cap,), -7.0, dtype=torch.float32, device="cuda")
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.bgamd_env_unique_rows_read(env._h, 0, cap, ctypes.c_void_p(stx.data_ptr()), ctypes.c_void_p(vx.data_ptr()), s)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(stx[:u], st) and torch.equal(vx[:u], val)
    assert int(stx[u:].abs().sum().item()) == 0 and float(vx[u:].abs().sum().item()) == 0.0
    # evaluate_incremental after a step: rows 0 .. m-1 of the list are the rows handed in
    pre, pt = _np(env.states()), _np(env.turns())
    roots, turns, cands, ridx = [], [], [], []
    for lane in range(40):
        sO = O.State.from28(pre[lane], pt[lane])
        _, _, c = O.evaluate_turn_sequences(sO, int(pt[lane]), 3, 1)
        if len(c):
            ridx += [len(roots)] * len(c)
            roots.append(pre[lane]); turns.append(pt[lane]); cands.append(c)
    cands = np.concatenate(cands).astype(np.int32)
    v = env.evaluate_incremental(np.array(roots, dtype=np.int32), np.array(turns, dtype=np.int32), cands, np.array(ridx, dtype=np.int32))
    m = len(cands)
    st2 = torch.empty((m, 28), dtype=torch.int32, device="cuda")
    rc = lib.bgamd_env_unique_rows_read(env._h, 0, m, ctypes.c_void_p(st2.data_ptr()), None, s)
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(_np(st2), cands)         # (its values go to the caller's buffer, not to the env's: only the rows are listed)
    assert v.shape[0] == m and float(v.min()) > 0.0 and float(v.max()) < 1.0


def test_ring_log_refuses_steps_that_break_its_game_table(bg, weights):
    """The ring log's contract is enforced: with a ring set, a greedy step without auto-reset or for one player only is BGAMD_E_INVALID
    (a game is the contiguous slots ending at its end record: such steps would make the game table point at other games' rows)."""
    env = bg.VecGame(256, seed=1)
    env.load_weights(weights)
    env.record_ring(64)
    env.reset(episode=0)
    env.run_greedy(4)                                                  # fine
    with pytest.raises(bg.BgamdError):
        env.run_greedy(2, auto_reset=False)
    with pytest.raises(bg.BgamdError):
        env.step_greedy(only_player=0)
    assert env.trajectory_step() == 4
    env.record_ring(None)
    env.run_greedy(2, auto_reset=False)                                # no ring: allowed again


def test_weights_outside_the_f16_split_are_refused_at_load(bg, weights):
    """A table the root pass's f16 hi + lo planes cannot hold is refused loudly and nothing of the slot is overwritten: the env goes on
    playing with the weights it had (the same games as an env that never saw the bad table)."""
    a, b = bg.VecGame(512, seed=8), bg.VecGame(512, seed=8)
    a.load_weights(weights)
    b.load_weights(weights)
    bad = weights.copy()
    bad[1234] = 1.0e5
    with pytest.raises(bg.BgamdError, match="65504"):
        b.load_weights(bad)
    nan = weights.copy()
    nan[25600] = np.nan
    with pytest.raises(bg.BgamdError):
        b.load_weights(nan)
    a.run_greedy(30)
    b.run_greedy(30)
    assert torch.equal(a.states(), b.states()) and torch.equal(a.turns(), b.turns())

