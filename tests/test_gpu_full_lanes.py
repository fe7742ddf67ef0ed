"""BASELINE.json's full size, every lane: all 65 536 lanes of a greedy step against the oracle, not a sample of them."""
import multiprocessing as mp
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bg():
    import backgammon_env
    return backgammon_env


def test_every_lane_of_65536_vs_oracle(bg, golden_dir, weights):
    """What test_greedy_65536_sampled_lanes_vs_oracle checks on 255 lanes, on ALL lanes that are live, at three phases of the games
    (plies ~12, ~38, ~64 of greedy play, auto-reset on in between): the state every lane is in after step_greedy is one of the oracle's
    afterstates of its pre-move state and dice (the reference's evaluateTurnSequences restated in C), and the fp64 restatement of the
    reference model puts its value within 1e-5 of the arg-max (PLAYER1) / arg-min (PLAYER2).  The oracle work runs in spawned worker
    processes (tests/full_lane_worker.py: numpy + oracle, no GPU)."""
    import full_lane_worker as W
    n = 65536
    env = bg.VecGame(n, seed=20240603)
    env.load_weights(weights)
    workers = min(16, os.cpu_count() or 1)
    wpath = os.path.join(golden_dir, "tdgammonNEW100k.f32")
    env.run_greedy(12)
    with mp.get_context("spawn").Pool(workers, initializer=W.init, initargs=(wpath,)) as pool:
        for step in range(3):
            pre, pt = env.states().cpu().numpy(), env.turns().cpu().numpy()
            live = (env.flags().cpu().numpy() & 4) == 0
            env.step_greedy(auto_reset=False)
            post, dice = env.states().cpu().numpy(), env.dice().cpu().numpy()
            idx = np.nonzero(live)[0]
            assert len(idx) > 0.9 * n
            t0 = time.time()
            res = pool.map(W.check, [(pre[c], pt[c], dice[c], post[c]) for c in np.array_split(idx, workers * 8) if len(c)])
            bad = [r for r in res if r[0] != "OK"]
            assert not bad, bad[0]
            print("step %d: %d live lanes checked, %d oracle afterstates, %d stuck lanes; max |value of the applied state - best| = %.3g; "
                  "%d lanes (%.4f %%) took a state other than the fp64 oracle's first best index (all within 1e-5); %.0f s on %d workers"
                  % (step, len(idx), sum(r[4] for r in res), sum(r[3] for r in res), max(r[1] for r in res), sum(r[2] for r in res),
                     100.0 * sum(r[2] for r in res) / len(idx), time.time() - t0, workers), flush=True)
            env.run_greedy(25)
    assert env.stats()["error_flags"] == 0


def test_every_lane_of_65536_random_policy_vs_oracle(bg, golden_dir):
    """Config 2 at BASELINE's size: ALL 65 536 lanes of the random-policy env, every step of 128 (auto-reset on: most lanes finish a game and start the next), against the oracle's
    whole-env run of the same lane (test_random_trajectories_vs_oracle_4096 at 16 x the lanes): every state, turn and step flag equal,
    and the env's counters equal the sums over the lanes."""
    import full_lane_worker as W
    n, steps, seed = 65536, 128, 424242
    env = bg.VecGame(n, seed=seed)
    snaps = np.zeros((steps, n, 29), dtype=np.int8)
    flags = np.zeros((steps, n), dtype=np.int8)
    for t in range(steps):
        env.step_random()
        snaps[t, :, :28] = env.states().cpu().numpy()
        snaps[t, :, 28] = env.turns().cpu().numpy()
        flags[t] = (env.flags().cpu().numpy() >> 4) & 3
    workers = min(16, os.cpu_count() or 1)
    wpath = os.path.join(golden_dir, "tdgammonNEW100k.f32")
    t0 = time.time()
    with mp.get_context("spawn").Pool(workers, initializer=W.init, initargs=(wpath,)) as pool:
        chunks = np.array_split(np.arange(n), workers * 8)
        res = pool.map(W.run_random, [(seed, c, n, steps, snaps[:, c], flags[:, c].astype(np.int32)) for c in chunks])
    bad = [r for r in res if r[0] != "OK"]
    assert not bad, bad[0]
    s = env.stats()
    assert s["games_finished"] == sum(r[1] for r in res) and s["candidates_raw"] == sum(r[2] for r in res) and s["steps"] == n * steps
    print("random policy: %d lanes x %d steps state-for-state equal to the oracle's runs (%d games finished, %d raw candidates); %.0f s on %d workers"
          % (n, steps, s["games_finished"], s["candidates_raw"], time.time() - t0, workers))


def test_every_lane_of_32768_bf16_selfplay_vs_oracle(bg, golden_dir, weights):
    """Config 5's per-GPU share, every lane: 32 768 lanes stepping with the bf16 value net (the speed mode outside the 1e-5 bound).  Whatever
    the precision of the values, the MOVES must be the reference's: the state every live lane is in after the step is one of the oracle's
    afterstates of its pre-move state and dice; and the fp64 value of the state bf16 picked is within 5e-3 of the best (measured below)."""
    import full_lane_worker as W
    n = 32768
    env = bg.VecGame(n, seed=5150)
    env.load_weights(weights)
    workers = min(16, os.cpu_count() or 1)
    wpath = os.path.join(golden_dir, "tdgammonNEW100k.f32")
    env.run_greedy(10, precision=bg.BF16)
    with mp.get_context("spawn").Pool(workers, initializer=W.init, initargs=(wpath,)) as pool:
        for step in range(2):
            pre, pt = env.states().cpu().numpy(), env.turns().cpu().numpy()
            live = (env.flags().cpu().numpy() & 4) == 0
            env.step_greedy(auto_reset=False, precision=bg.BF16)
            post, dice = env.states().cpu().numpy(), env.dice().cpu().numpy()
            idx = np.nonzero(live)[0]
            res = pool.map(W.check, [(pre[c], pt[c], dice[c], post[c], 5e-3) for c in np.array_split(idx, workers * 8) if len(c)])
            bad = [r for r in res if r[0] != "OK"]
            assert not bad, bad[0]
            print("bf16 step %d: %d live lanes, every applied state a legal afterstate; max fp64 value gap to the best %.3g; %d lanes (%.2f %%) "
                  "not the fp64 oracle's first best index" % (step, len(idx), max(r[1] for r in res), sum(r[2] for r in res),
                                                              100.0 * sum(r[2] for r in res) / len(idx)), flush=True)
            env.run_greedy(30, precision=bg.BF16)
    assert env.stats()["error_flags"] == 0


def test_every_lane_of_65536_epsilon_greedy_vs_oracle(bg, golden_dir, weights):
    """Config 4's self-play policy at its per-GPU size, every lane: step_greedy(epsilon = 0.1) on 65 536 lanes -- a lane whose TURN-stream
    draw says "explore" (model.py:205-206) plays exactly reference-order candidate k = (x2 C) >> 32 of the oracle's enumeration, every
    other lane plays an afterstate whose fp64 value is within 1e-5 of the best."""
    import full_lane_worker as W
    n, seed, eps = 65536, 97531, 0.1
    env = bg.VecGame(n, seed=seed)
    env.load_weights(weights)
    env.run_greedy(20, epsilon=eps)
    pre, pt = env.states().cpu().numpy(), env.turns().cpu().numpy()
    ply, epi = [x.cpu().numpy() for x in env.progress()]
    live = (env.flags().cpu().numpy() & 4) == 0
    env.step_greedy(epsilon=eps, auto_reset=False)
    post, chosen = env.states().cpu().numpy(), env.last_choice()["chosen"].cpu().numpy()
    idx = np.nonzero(live)[0]
    workers = min(16, os.cpu_count() or 1)
    wpath = os.path.join(golden_dir, "tdgammonNEW100k.f32")
    with mp.get_context("spawn").Pool(workers, initializer=W.init, initargs=(wpath,)) as pool:
        res = pool.map(W.check_eps, [(seed, n, eps, c, pre[c], pt[c], ply[c], epi[c], post[c], chosen[c]) for c in np.array_split(idx, workers * 8)])
    bad = [r for r in res if r[0] != "OK"]
    assert not bad, bad[0]
    n_explore = sum(r[1] for r in res)
    print("epsilon-greedy %.2f: %d live lanes, %d explored (candidate k of the reference-order list, exactly), the others within %.3g of the best"
          % (eps, len(idx), n_explore, max(r[2] for r in res)))
    assert 0.08 * len(idx) < n_explore < 0.12 * len(idx)
