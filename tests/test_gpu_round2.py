"""Round-2 GPU parity tests (all through the C ABI):
  * the headline kernels -- the root pass (root_hidden_resident_kernel / boundary_kernel<true>) + eval_rows_delta_kernel -- compared ROW BY ROW with the reference
    model's own outputs (fixture G7, written by the unmodified reference) and with the fp64 oracle;
  * the 65 536-lane configuration (second-stream root pass, fused step boundaries) against the oracle on sampled lanes;
  * the per-GPU shares of the training configs: a 65 536-game TD(lambda) round (config 4) and a 32 768-lane round with
    bf16 self-play and fp32 traces (config 5);
  * the 2-rank layout rehearsed by two fresh processes that share the one GPU (gloo);
  * full games through the scalar drop-in surface in the shape of train.py:103-121 against fixture G3.
Integer work is bit-exact; value-net outputs within 1e-5 of the reference (north_star)."""
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

from test_gpu_parity import START, _check_greedy_step, _np

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


def _state_from_features(x):
    """198 encoder features (model.py:111-144) -> (state28 int32, turn): thermometer counts back to checker counts."""
    s = np.zeros(28, dtype=np.int32)
    for i in range(24):
        for side, sgn in ((0, 1), (1, -1)):
            f = x[8 * i + 4 * side: 8 * i + 4 * side + 4]
            n = int(round(float(f[0] + f[1] + f[2] + 2.0 * f[3])))
            if n:
                assert s[i] == 0
                s[i] = sgn * n
    s[24], s[25] = int(round(2 * float(x[194]))), int(round(2 * float(x[195])))
    s[26], s[27] = int(round(15 * float(x[196]))), int(round(15 * float(x[197])))
    assert abs(s[:24][s[:24] > 0].sum() + s[24] + s[26] - 15) == 0 and abs(-s[:24][s[:24] < 0].sum() + s[25] + s[27] - 15) == 0
    return s, (0 if x[192] == 1.0 else 1)


def _rows_by_state(states):
    return {tuple(int(v) for v in s): i for i, s in enumerate(states)}


# ---- the headline value-net path, row by row ---------------------------------------------------------------------

def test_delta_kernel_rows_vs_reference_values(bg, O, golden_dir, weights):
    """Fixture G7: 124 turns of the reference's greedy games, every distinct afterstate with the value the reference
    model's forward pass gave it.  One greedy step in BGAMD_F32 on those turns: the rows handed to the value net are
    EXACTLY the reference's distinct afterstates and EVERY per-row output of eval_rows_delta_kernel (root term from
    root_hidden_resident_kernel: W1 as f16 hi + lo since round 4) is within 1e-5 of the reference (fp32 and fp64)."""
    g = np.load(os.path.join(golden_dir, "g7_candidate_values.npz"))
    roots, off = g["roots"], g["off"]
    R = len(roots)
    env = bg.VecGame(R, arena_rows=1 << 20)
    env.load_weights(weights)
    env.set_states(roots[:, :28], roots[:, 28])
    env.set_dice(roots[:, 29:31])
    env.step_greedy(roll=False, auto_reset=False, precision=bg.F32)
    info, st, val = [_np(x) for x in env.unique_rows()]
    assert env.stats()["error_flags"] == 0
    e32 = e64 = 0.0
    n_rows = 0
    for k in range(R):
        m = info[:, 0] == k
        ref = _rows_by_state(g["states"][off[k]:off[k + 1]])
        got = {tuple(int(v) for v in s) for s in st[m]}
        assert got == set(ref), k                          # the staged rows ARE the reference's distinct afterstates
        for s, v in zip(st[m], val[m]):
            j = off[k] + ref[tuple(int(x) for x in s)]
            e32 = max(e32, abs(float(v) - float(g["v32"][j])))
            e64 = max(e64, abs(float(v) - float(g["v64"][j])))
            n_rows += 1
    print("eval_rows_delta_kernel, %d rows of %d turns: max |gpu - reference fp32| = %.3g, max |gpu - reference fp64| = %.3g"
          % (n_rows, R, e32, e64))
    assert n_rows >= int(off[-1]) and e32 < 1e-5 and e64 < 1e-5
    # the same rows through the stateless incremental operator: the same kernels, bit for bit
    ridx = np.repeat(np.arange(R), np.diff(off)).astype(np.int32)
    v2 = _np(env.evaluate_incremental(roots[:, :28], roots[:, 28], g["states"].astype(np.int32), ridx))
    assert np.abs(v2 - g["v32"]).max() < 1e-5 and np.abs(v2 - g["v64"]).max() < 1e-5
    by_key = {}
    for s, v, gi in zip(st, val, info[:, 0]):
        by_key[(int(gi),) + tuple(int(x) for x in s)] = v
    same = [by_key[(int(r),) + tuple(int(x) for x in s)] == v for s, v, r in zip(g["states"], v2, ridx)]
    assert all(same)
    # root == row (empty delta list): the root pass alone against fixture G5's value rows
    g5 = np.load(os.path.join(golden_dir, "g5_values.npz"))
    big = bg.VecGame(len(g5["turn"]), arena_rows=1 << 20)
    big.load_weights(weights)
    v3 = _np(big.evaluate_incremental(g5["states"].astype(np.int32), g5["turn"], g5["states"].astype(np.int32),
                                      np.arange(len(g5["turn"]), dtype=np.int32)))
    print("root pass alone (root_hidden_resident_kernel): max |gpu - reference fp32| = %.3g" % np.abs(v3 - g5["v32"]).max())
    assert np.abs(v3 - g5["v32"]).max() < 1e-5 and np.abs(v3 - g5["v64"]).max() < 1e-5


def test_delta_kernel_rows_midgame_4096_vs_oracle(bg, O, weights):
    """4 096 mid-game lanes: EVERY row value of the incremental kernel against the oracle's fp64 forward of that row."""
    n = 4096
    env = bg.VecGame(n, seed=808)
    env.load_weights(weights)
    env.run_greedy(35)                                       # de-phased mid-game positions (some already bearing off)
    pt = _np(env.turns())
    env.step_greedy(auto_reset=False, precision=bg.F32)
    info, st, val = [_np(x) for x in env.unique_rows()]
    assert len(st) > 10 * n and env.stats()["error_flags"] == 0
    mover = pt[info[:, 0]]
    assert ((info[:, 1] >> 31) == mover).all()               # rows carry the MOVER's turn bit (model.py:209)
    worst = 0.0
    for tb in (0, 1):
        m = mover == tb
        v64 = O.forward_f64(weights, O.encode(st[m], tb))
        worst = max(worst, float(np.abs(val[m] - v64).max()))
    print("eval_rows_delta_kernel, %d mid-game rows: max |gpu - oracle fp64| = %.3g" % (len(st), worst))
    assert worst < 1e-5


def test_delta_list_worst_case_and_overflow_flag(bg, O, weights):
    """The longest (feature, delta) list a legal turn can produce is 13 entries: four single checkers each leave their
    point and hit a blot (4 origins + 4 landing points of the mover, 4 hit points + the bar counter of the opponent).
    That turn is evaluated exactly; a row that is NOT an afterstate of its root (more than 16 changed features) raises
    BGAMD_E_DELTA instead of being truncated silently."""
    board = [0] * 24
    for p in (1, 3, 5, 7):
        board[p - 1] = 1                                     # PLAYER1 singles on 1, 3, 5, 7
        board[p] = -1                                        # PLAYER2 blots on 2, 4, 6, 8
    board[19] = 11                                           # the other 11 PLAYER1 checkers
    board[23] = -11                                          # the other 11 PLAYER2 checkers
    root = np.array(board + [0, 0, 0, 0], dtype=np.int32)
    env = bg.VecGame(64, arena_rows=1 << 18)
    env.load_weights(weights)
    env.set_states(np.tile(root, (64, 1)), np.zeros(64, dtype=np.int32))
    env.set_dice(np.tile(np.array([[1, 1]], dtype=np.int32), (64, 1)))
    env.step_greedy(roll=False, auto_reset=False, precision=bg.F32)
    info, st, val = [_np(x) for x in env.unique_rows()]
    assert env.stats()["error_flags"] == 0
    m = info[:, 0] == 0
    after = board.copy()
    for p in (1, 3, 5, 7):
        after[p - 1] = 0
        after[p] = 1
    want = np.array(after + [0, 4, 0, 0], dtype=np.int32)     # four PLAYER2 checkers on the bar
    hit = [i for i in np.nonzero(m)[0] if (st[i] == want).all()]
    assert hit, "the four-hit turn is among the candidates"
    x0, x1 = O.encode(root[None], 0)[0], O.encode(want[None], 0)[0]
    assert int((x0 != x1).sum()) == 13                        # the bound, stated on the encoder itself
    cand = O.evaluate_turn_sequences(O.State.from28(root, 0), 0, 1, 1)[2]
    assert {tuple(int(v) for v in s) for s in st[m]} == {tuple(int(v) for v in s) for s in cand}
    v64 = O.forward_f64(weights, O.encode(st[m], 0))
    assert np.abs(val[m] - v64).max() < 1e-5
    # every legal turn stays inside the bound: the largest list over all rows of a late, contact-heavy sample
    # (ksteps_executed counts the list entries) -- and the flag itself, on rows that are no afterstates of their root
    far = np.array([0, 0, 0, 0, 0, 5, 0, 3, 0, 0, 0, -5, 5, 0, 0, 0, -3, 0, -5, 0, 0, 0, 0, 2] + [0, 0, 0, 0], dtype=np.int32)
    x2 = O.encode(far[None], 0)[0]
    assert int((x0 != x2).sum()) > 16
    with pytest.raises(bg.BgamdError, match="incremental value net"):
        env.evaluate_incremental(root[None], [0], far[None], [0])
        env.stats()
    env.reset_stats()
    # a root index outside the given roots is an invalid state, not a wild read
    with pytest.raises(bg.BgamdError):
        env.evaluate_incremental(root[None], [0], want[None], [5])
        env.stats()


# ---- the headline configuration: 65 536 lanes, second-stream root pass, fused boundaries -------------------------------

def _greedy_65536_sampled_lanes(bg, O, weights, twin_switches=()):
    """At 65 536 lanes run_greedy fuses apply(t) + roots(t+1) + (round 4) the value net's root pass of step t + 1 into one launch, everything
    on the caller's stream.  Sampled lanes are checked against the oracle after ONE step_greedy and after run_greedy(8) -- whose 8th step is
    observable through a twin env that takes 7 steps of a run and then ONE separate step; the twin must stay bit-identical throughout.
    twin_switches (experimental build, tests/test_gpu_experimental.py): the twin with rounds 1-3's launch structure (the root pass a launch
    of its own, forked onto the env's second stream: BGAMD_ROOT_IN_BOUNDARY=0 + BGAMD_OVERLAP=1)."""
    n = 65536
    a = bg.VecGame(n, seed=777)
    for k in twin_switches:
        os.environ[k.split("=")[0]] = k.split("=")[1]
    try:
        b = bg.VecGame(n, seed=777)
    finally:
        for k in twin_switches:
            del os.environ[k.split("=")[0]]
    a.load_weights(weights); b.load_weights(weights)
    a.run_greedy(30); b.run_greedy(30)
    lanes = list(range(5, n, 257))                            # 255 lanes
    for rnd in range(2):
        pre, pt = _np(a.states()), _np(a.turns())
        assert np.array_equal(pre, _np(b.states())) and np.array_equal(pt, _np(b.turns()))
        frozen = (_np(a.flags()) & 4) != 0                    # finished without auto-reset earlier in this test: skipped
        a.step_greedy(auto_reset=False); b.step_greedy(auto_reset=False)
        post, dice = _np(a.states()), _np(a.dice())
        assert np.array_equal(post, _np(b.states()))
        _check_greedy_step(O, weights, pre, pt, dice, post, [l for l in lanes if not frozen[l]])
        # run_greedy(8): a takes the fused route, b takes 7 fused steps and then ONE separate step whose
        # pre-state is observable -- the 8th step of a is checked against the oracle from b's pre-state
        a.run_greedy(8, auto_reset=False)
        b.run_greedy(7, auto_reset=False)
        pre, pt = _np(b.states()), _np(b.turns())
        frozen = (_np(b.flags()) & 4) != 0
        b.step_greedy(auto_reset=False)
        post, dice = _np(a.states()), _np(a.dice())
        assert np.array_equal(post, _np(b.states())) and np.array_equal(_np(a.turns()), _np(b.turns()))
        _check_greedy_step(O, weights, pre, pt, dice, post, [l for l in lanes if not frozen[l]])
        a.run_greedy(20); b.run_greedy(20)                    # on to later game phases (auto-reset on)
    sa, sb = a.stats(), b.stats()
    assert sa["error_flags"] == 0 and sb["error_flags"] == 0
    assert all(sa[k] == sb[k] for k in ("steps", "games_finished", "p1_wins", "rows_evaluated"))
    return a, b


def test_greedy_65536_sampled_lanes_vs_oracle(bg, O, weights):
    _greedy_65536_sampled_lanes(bg, O, weights)


def _stream_modes_run_and_graph_capture(bg, weights, mode):
    """The fork / join of the root pass (BGAMD_OVERLAP=1 forces it on a small env) under run_greedy(k > 1) and inside a
    captured HIP graph: bit-identical to single-stream stepping."""
    n = 3000
    os.environ[mode] = "1"
    try:
        a, c = bg.VecGame(n, seed=321), bg.VecGame(n, seed=321)
    finally:
        del os.environ[mode]
    os.environ["BGAMD_NO_OVERLAP"] = "1"
    try:
        b = bg.VecGame(n, seed=321)
    finally:
        del os.environ["BGAMD_NO_OVERLAP"]
    for e in (a, b, c):
        e.load_weights(weights)
    for k in (1, 3, 8, 16):
        a.run_greedy(k, epsilon=0.1)
        for _ in range(k):
            b.step_greedy(epsilon=0.1)
        assert np.array_equal(_np(a.states()), _np(b.states())) and np.array_equal(_np(a.turns()), _np(b.turns())), k
    # graph capture of the forked step
    c.run_greedy(28, epsilon=0.1)
    assert np.array_equal(_np(c.states()), _np(b.states()))
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            c.run_greedy(4, epsilon=0.1)
    for _ in range(3):
        g.replay()
    b.run_greedy(12, epsilon=0.1)
    torch.cuda.synchronize()
    assert np.array_equal(_np(c.states()), _np(b.states())) and np.array_equal(_np(c.turns()), _np(b.turns()))
    assert a.stats()["error_flags"] == 0 and c.stats()["error_flags"] == 0


# ---- training configs: per-GPU shares ---------------------------------------------------------------------------------

def test_rounds_play_fresh_games(bg, weights):
    """Every round of play_round is a new episode of every lane: new opening rolls, new dice, new exploration draws
    (play_game rolls fresh dice for every game, train.py:64-121); an explicit episode replays that round exactly."""
    from backgammon_env.learner import play_round
    env = bg.VecGame(512, seed=5)
    env.load_weights(weights)
    r0 = play_round(env, max_plies=300, epsilon=0.05)
    r1 = play_round(env, max_plies=300, epsilon=0.05)
    assert not torch.equal(r0[1], r1[1])                      # game lengths differ lane by lane
    k = min(r0[0].shape[0], r1[0].shape[0], 3)
    assert not torch.equal(r0[0][1:k], r1[0][1:k])            # ... because the games do (ply 0 is always the start position)
    _, epi = env.progress()
    assert (_np(epi) == 1).all()
    again = play_round(env, max_plies=300, epsilon=0.05, episode=0)
    assert torch.equal(again[1], r0[1]) and torch.equal(again[2], r0[2])
    T = again[0].shape[0]
    assert torch.equal(again[0], r0[0][:T])
    assert env.stats()["error_flags"] == 0


def _torch_fp64_replay(bg, env, weights, rows, lengths, p1_won, alpha, lam, scale):
    """The closed form of learner.TDLambdaLearner in float64 on the GPU (PyTorch): the checker for whole rounds."""
    from backgammon_env.learner import TDLambdaLearner
    L = TDLambdaLearner(weights, device="cuda", alpha=alpha, lam=lam, dtype=torch.float64)
    X = env.encode_rows(rows)                                  # [T, n, 198] fp32, cast per step inside replay
    sq, cnt = L.replay(X, lengths, p1_won, batch_scale=scale)
    th = L.theta.cpu().numpy()
    del X, L
    torch.cuda.empty_cache()
    return th, sq, cnt


def test_config4_share_td_round_65536(bg, O, weights):
    """Config 4's per-GPU share: ONE round of 65 536 concurrent self-play games (epsilon-greedy, frozen weights,
    train.py:527-547) logged as 32-byte rows, then the lock-step TD(lambda) replay of all of them on the HIP learner
    (6.7 GB of fp32 traces).  Checked: every game finishes, checkers conserved in the log, (game, step) count, two
    replays bit-identical, the update equals the float64 closed form over the WHOLE round (PyTorch on the GPU), and a
    200-game subset equals the oracle's numpy restatement of the reference learner."""
    from backgammon_env.learner import DeviceTDLambdaLearner, play_round
    n = 65536
    env = bg.VecGame(n, seed=31337)
    env.load_weights(weights)
    t0 = time.time()
    rows, lengths, p1_won = play_round(env, max_plies=320, epsilon=0.05)
    torch.cuda.synchronize()
    t_play = time.time() - t0
    ln = _np(lengths)
    st = env.stats()
    assert st["error_flags"] == 0 and st["games_finished"] >= (ln > 0).sum() and (ln > 0).mean() > 0.995
    # the log holds legal positions: checker conservation on a sample of rows, first row = start position
    X0 = _np(env.encode_rows(rows[0, :64]))
    assert (X0[:, :192] == O.encode(np.array([START + [0, 0, 0, 0]], dtype=np.int32), 0)[0, :192]).all()
    scale = 24.0 / n
    out = []
    for rep in range(2):
        Ld = DeviceTDLambdaLearner(weights, max_games=n, alpha=0.1, lam=0.9)
        t1 = time.time()
        sq_d, cnt_d = Ld.replay_rows(rows, lengths, p1_won, batch_scale=scale)
        t_rep = time.time() - t1
        out.append(_np(Ld.theta))
        assert cnt_d == int(ln.sum())
        del Ld
        torch.cuda.empty_cache()
    assert np.array_equal(out[0], out[1])                      # deterministic run to run
    print("config-4 share: %d games, %d turns; self-play %.2f s, replay %.2f s (%.1f M turn-updates/s)"
          % (int((ln > 0).sum()), int(ln.sum()), t_play, t_rep, ln.sum() / t_rep / 1e6))
    th64, sq64, cnt64 = _torch_fp64_replay(bg, env, weights, rows, lengths, p1_won, 0.1, 0.9, scale)
    moved = np.abs(th64 - weights).max()
    d = np.abs(out[0] - th64).max()
    print("whole round vs float64 closed form: max |dtheta| = %.3g, max |device - fp64| = %.3g, td loss %.5f" % (moved, d, sq64 / cnt64))
    assert cnt64 == cnt_d and moved > 1e-3 and d < 1e-4 * max(1.0, moved)
    assert abs(sq_d - sq64) < 1e-3 * max(1.0, sq64)
    # 200-game subset: device learner == the oracle's restatement of the reference learner (float64, CPU)
    sub = np.nonzero(ln > 0)[0][:: max(1, int((ln > 0).sum()) // 200)][:200]
    ls = torch.zeros_like(lengths)
    ls[torch.as_tensor(sub, device=lengths.device)] = lengths[torch.as_tensor(sub, device=lengths.device)]
    Ls = DeviceTDLambdaLearner(weights, max_games=256, alpha=0.1, lam=0.9)
    sq_s, cnt_s = Ls.replay_rows(rows, ls, p1_won, batch_scale=0.25)
    Tm = int(ln[sub].max())
    Xs = env.encode_rows(rows[:Tm, torch.as_tensor(sub, device=rows.device)].contiguous()).cpu().double().numpy()
    th_o, sq_o, cnt_o = O.td_lambda_lockstep(weights, Xs, ln[sub], _np(p1_won)[sub], 0.1, 0.9, batch_scale=0.25)
    moved_s = np.abs(th_o - weights).max()
    assert cnt_s == cnt_o == int(ln[sub].sum())
    assert np.abs(_np(Ls.theta) - th_o).max() < 2e-5 * max(1.0, moved_s), (np.abs(_np(Ls.theta) - th_o).max(), moved_s)


def test_config5_share_bf16_selfplay_fp32_traces_32768(bg, O, weights):
    """Config 5's per-GPU share: 32 768 concurrent games played with the bf16 MFMA value net (speed mode), logged,
    and replayed with fp32 weights and fp32 traces.  The bf16 games are legal games (every logged transition is a
    candidate of the oracle on sampled lanes) that mostly pick the fp32 move; the learner equals the float64 closed
    form on the same log; sub-rounds apply one after another."""
    from backgammon_env.learner import DeviceTDLambdaLearner, play_round
    n = 32768
    env = bg.VecGame(n, seed=2718)
    env.load_weights(weights)
    rows, lengths, p1_won = play_round(env, max_plies=320, epsilon=0.0, precision=bg.BF16)
    ln = _np(lengths)
    assert env.stats()["error_flags"] == 0 and (ln > 0).mean() > 0.995
    # sampled lanes: each logged transition s_t -> s_{t+1} is one of the oracle's afterstates of s_t (with the turn flipped)
    S = _np(env.encode_rows(rows[:12, :4096:97].contiguous()))      # [12, 43, 198]
    legal = 0
    lanes = list(range(0, 4096, 97))
    rolls = [(a, b) for a in range(1, 7) for b in range(a, 7)]
    for j, lane in enumerate(lanes):
        for t in range(10):
            if t + 1 >= ln[lane]:
                break
            (s0, tb), (s1, tb1) = _state_from_features(S[t, j]), _state_from_features(S[t + 1, j])
            assert tb1 == 1 - tb                               # the turn flipped
            ok = False
            for d1, d2 in rolls:                               # the dice are not logged: some roll must explain the move
                cand = O.evaluate_turn_sequences(O.State.from28(s0, tb), tb, d1, d2)[2]
                if (len(cand) == 0 and (s1 == s0).all()) or (len(cand) and (cand == s1[None, :]).all(axis=1).any()):
                    ok = True
                    break
            assert ok, (lane, t, s0.tolist(), s1.tolist())
            legal += 1
    assert legal > 300
    # agreement of the bf16 choices with fp32 on the same positions (one step from a common mid-game state)
    a, b = bg.VecGame(8192, seed=99), bg.VecGame(8192, seed=99)
    a.load_weights(weights); b.load_weights(weights)
    a.run_greedy(25); b.run_greedy(25)
    a.step_greedy(precision=bg.F32, auto_reset=False); b.step_greedy(precision=bg.BF16, auto_reset=False)
    agree = (_np(a.states()) == _np(b.states())).all(axis=1).mean()
    print("bf16 vs fp32 move agreement at 8 192 mid-game lanes: %.4f" % agree)
    assert agree > 0.93
    scale = 24.0 / n
    Ld = DeviceTDLambdaLearner(weights, max_games=n, alpha=0.1, lam=0.9)
    sq_d, cnt_d = Ld.replay_rows(rows, lengths, p1_won, batch_scale=scale)
    th = _np(Ld.theta)
    assert cnt_d == int(ln.sum()) and np.isfinite(th).all()
    th64, sq64, cnt64 = _torch_fp64_replay(bg, env, weights, rows, lengths, p1_won, 0.1, 0.9, scale)
    moved = np.abs(th64 - weights).max()
    assert cnt64 == cnt_d and moved > 1e-3 and np.abs(th - th64).max() < 1e-4 * max(1.0, moved)
    # sub-rounds: 64 sub-rounds of 512 games, each from the weights the one before left
    Ld.set_weights(weights)
    sq_s, cnt_s = Ld.replay_rows(rows, lengths, p1_won, batch_scale=24.0 / 512, sub_round=512)
    th_s = _np(Ld.theta)
    assert cnt_s == cnt_d and np.isfinite(th_s).all() and np.abs(th_s - weights).max() > 1e-3
    assert not np.array_equal(th_s, th)
    # ... and with one sub-round per game the replay IS the reference's order of updates (a 24-game round, fp64 check)
    few = np.nonzero(ln > 0)[0][:24]
    lf = torch.zeros_like(lengths)
    idx = torch.as_tensor(few, device=lengths.device)
    lf[idx] = lengths[idx]
    Ld.set_weights(weights)
    Ld.replay_rows(rows, lf, p1_won, sub_round=1)
    th_seq = _np(Ld.theta)
    from backgammon_env.learner import TDLambdaLearner
    Lc = TDLambdaLearner(weights, device="cpu", alpha=0.1, lam=0.9, dtype=torch.float64)
    order = few[np.argsort(-ln[few], kind="stable")]          # the learner's order: by decreasing length (stable)
    for lane in order:
        one = torch.zeros_like(lengths).cpu()
        one[lane] = int(ln[lane])
        Xl = env.encode_rows(rows[:int(ln[lane]), lane:lane + 1].contiguous()).cpu()
        Lc.replay(Xl, one[lane:lane + 1], p1_won[lane:lane + 1].cpu())
    moved = np.abs(Lc.theta.numpy() - weights).max()
    assert np.abs(th_seq - Lc.theta.numpy()).max() < 5e-5 * max(1.0, moved), (np.abs(th_seq - Lc.theta.numpy()).max(), moved)


# ---- two ranks on one GPU (gloo): configs 3 and 4 as the multi-GPU bench / training loop lay them out ---------------------

def test_two_rank_rehearsal_on_one_gpu(bg, weights, tmp_path):
    """Two FRESH processes (spawned before they touch the GPU) share the one GPU and a gloo group:
      * 2 x 32 768 lanes play, lane for lane, the games of ONE 65 536-lane env (50 greedy steps);
      * the split / all-reduce / apply learner over two shards leaves bit-identical weight replicas that moved."""
    lanes, steps, train_lanes = 32768, 50, 512
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env_vars = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), "2", str(port), str(tmp_path),
                               str(lanes), str(steps), str(train_lanes)], env=env_vars, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    whole = bg.VecGame(2 * lanes, seed=4242)                   # meanwhile, the one-env reference run in this process
    whole.load_weights(weights)
    whole.run_greedy(steps)
    st, tn, sw = _np(whole.states()), _np(whole.turns()), whole.stats()
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            raise
        outs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(x[-3000:] for x in outs)
    r = [np.load(os.path.join(tmp_path, f"rank{k}.npz")) for k in range(2)]
    for k in range(2):
        assert np.array_equal(r[k]["states"], st[k * lanes:(k + 1) * lanes]), k
        assert np.array_equal(r[k]["turns"], tn[k * lanes:(k + 1) * lanes]), k
        assert list(r[k]["totals"]) == [sw["steps"], sw["games_finished"], sw["p1_wins"]]       # all-reduced counters
    assert np.array_equal(r[0]["theta"], r[1]["theta"])        # replicas identical after three rounds of all-reduced updates (lock-step, sub-rounds, streamed)
    assert np.abs(r[0]["theta"] - weights).max() > 1e-4
    assert not np.array_equal(r[0]["lengths"], r[1]["lengths"])   # the two shards played different games
    assert r[0]["learner"][1] > 0 and r[1]["learner"][1] > 0
    # sharded == unsharded for the DEVICE learner on the lock-step route: both shards' round-0 logs replayed by ONE learner
    # (1 024 games, the same per-game step) give the weights the two all-reducing replicas ended round 0 with, to fp32
    # rounding -- the sum over the games is associated differently, nothing else
    from backgammon_env.learner import DeviceTDLambdaLearner
    assert np.array_equal(r[0]["r0_theta"], r[1]["r0_theta"])
    T = max(r[0]["r0_rows"].shape[0], r[1]["r0_rows"].shape[0])
    rows_all = np.zeros((T, 2 * train_lanes, 8), dtype=np.int32)
    for k in range(2):
        rows_all[:r[k]["r0_rows"].shape[0], k * train_lanes:(k + 1) * train_lanes] = r[k]["r0_rows"]
    lengths_all = np.concatenate([r[0]["r0_lengths"], r[1]["r0_lengths"]])
    p1_all = np.concatenate([r[0]["r0_p1_won"], r[1]["r0_p1_won"]])
    one = DeviceTDLambdaLearner(weights, max_games=2 * train_lanes, alpha=0.1, lam=0.9)
    one.replay_rows(torch.from_numpy(rows_all).cuda(), torch.from_numpy(lengths_all).cuda(), torch.from_numpy(p1_all).cuda(),
                    batch_scale=24.0 / (2 * train_lanes))
    th_one = _np(one.theta)
    moved = np.abs(th_one - weights).max()
    gap = np.abs(th_one - r[0]["r0_theta"]).max()
    print("device learner, lock-step round: 2 all-reducing shards vs one learner over both logs: max |dtheta| %.3g (weights moved %.3g)" % (gap, moved))
    assert moved > 1e-4 and gap < 2e-5 * max(1.0, moved) and gap < 0.02 * moved


# ---- the scalar drop-in surface: whole games in the shape of the reference's loop ------------------------------------------

def test_pybind_module_full_games_vs_g3():
    """The compiled pybind11 module (the reference's own binding technology over the same C ABI) plays fixture G3's first 24
    games turn for turn, answers cppsrc/tests.cpp's known answers, clones and reports the reference's error strings -- in a
    process of its own (tests/pybind_driver.py: one extension module per name and process)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pybind_driver.py"), "games", "24"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    tag, turns, dt = r.stdout.split()[-3:]
    assert tag == "OK" and int(turns) == 2116
    print("scalar surface (config 1, pybind11 module): %s turns of 24 games in %s s = %.0f env steps/s" % (turns, dt, int(turns) / float(dt)))


def test_scalar_surface_full_games_vs_g3(bg, golden_dir, surface="python_package"):
    """24 complete games through the Python package's scalar Game exactly as train.py:103-121 / benchmark.py:54-61 drive the reference module:
    setDice -> evaluateTurnSequences -> pick by index -> tryMove x len -> is_game_over -> setTurn, compared turn by
    turn with fixture G3 (played by the unmodified reference on the same injected dice and choices).  Reports the
    config-1 throughput of this surface."""
    g = np.load(os.path.join(golden_dir, "g3_random_trajectories.npz"))
    rows = g["rows"]
    p1, p2 = bg.Player("White", bg.PlayerType.PLAYER1), bg.Player("Black", bg.PlayerType.PLAYER2)
    turns = 0
    t0 = time.time()
    for lane in range(24):
        r = rows[rows[:, 0] == lane]
        game = bg.Game(0)
        game.setPlayers(p1, p2)
        game.setTurn(int(r[0, 30]))
        for t in range(len(r)):
            turn, d1, d2, C, k = int(r[t, 30]), int(r[t, 31]), int(r[t, 32]), int(r[t, 33]), int(r[t, 34])
            assert game.getTurn() == turn
            assert list(game.getGameBoard()) == list(r[t, 2:26])
            assert [game.getJailedCount(0), game.getJailedCount(1), game.getBornOffCount(0), game.getBornOffCount(1)] == list(r[t, 26:30])
            game.setDice(d1, d2)
            assert list(game.get_last_dice()) == [d1, d2]
            seqs, states = game.evaluateTurnSequences(turn, d1, d2)
            assert len(seqs) == C and states.shape == (C, 28)
            if C:
                pl = game.getPlayers(turn)
                for o, d in seqs[k]:
                    ok, msg = game.tryMove(pl, abs(o - d), o, d)
                    assert ok and msg == ""
                assert list(game.getGameBoard()) + [game.getJailedCount(0), game.getJailedCount(1), game.getBornOffCount(0),
                                                    game.getBornOffCount(1)] == list(states[k])
            over, winner = game.is_game_over()
            assert int(over) == r[t, 35] and (not over or winner == r[t, 36])
            turns += 1
            if over:
                assert t == len(r) - 1
                break
            game.setTurn(1 - turn)
    dt = time.time() - t0
    print("scalar surface (config 1, %s): %d turns of 24 games in %.2f s = %.0f env steps/s" % (surface, turns, dt, turns / dt))
    # clone(): independent copy, cheap (pooled one-lane envs), last_dice reset to [1, 1] (game.cpp:68-77)
    game = bg.Game(1)
    game.setDice(3, 4)
    t0 = time.time()
    clones = [game.clone() for _ in range(50)]
    dt_first = time.time() - t0
    del clones
    created = []
    real_init = bg.VecGame.__init__

    def counting_init(self, *a, **k):
        created.append(1)
        real_init(self, *a, **k)

    bg.VecGame.__init__ = counting_init
    try:
        t0 = time.time()
        for _ in range(200):
            c = game.clone()
        dt_pool = (time.time() - t0) / 200
    finally:
        bg.VecGame.__init__ = real_init
    assert not created, "clone() allocated %d new envs while the pool held idle ones" % len(created)
    assert list(c.getGameBoard()) == list(game.getGameBoard()) and c.getTurn() == 1 and list(c.get_last_dice()) == [1, 1]
    assert c.tryMove(p2, 1, 6, 5)[0] and list(game.getGameBoard()) == START and list(c.getGameBoard()) != START
    assert c.tryMove(p2, 5, 6, 12) == (False, "Cannot move in that direction.") and c.getPieces().numJailed(0) == 0
    print("Game.clone(): %.2f ms each while the pool fills, %.3f ms from the pool" % (1e3 * dt_first / 50, 1e3 * dt_pool))


@pytest.mark.parametrize("variant", ["default", "direct", "matrix_pipe", "direct_and_wide", "matrix_pipe_and_wide"])
def test_streamed_replay_matches_host_closed_form(bg, weights, variant):
    """(variant direct_unfused -- BGAMD_TD_FUSED=0, experimental build -- runs in tests/test_gpu_experimental.py)"""
    _streamed_replay_variant(bg, weights, variant)


_STREAM_CLOSED_FORM = {}


def _streamed_replay_variant(bg, weights, variant):
    """bgamd_td_begin_stream: k slots replay the round's games one after another.  Against the float64 host closed form of the
    same schedule (ragged lengths, lanes that are not replayed, a one-turn game), for 1, 7 and 64 slots, on the small-round
    kernels and on the large-round ones (the two matrix-pipe forward kernels, the whole-row trace workgroups: forced down to this size); a slot
    per game equals the lock-step replay; the (game, step) count is the round's turns; deterministic run to run."""
    from backgammon_env.learner import DeviceTDLambdaLearner, TDLambdaLearner, play_round
    n = 160
    env = bg.VecGame(n, seed=321)
    env.load_weights(weights)
    rows, lengths, p1_won = play_round(env, max_plies=400, epsilon=0.1)
    lengths = lengths.clone()
    lengths[::9] = 0
    lengths[5] = 1
    turns = int(lengths.sum().item())
    Xr = env.encode_rows(rows).cpu().double()
    if "matrix_pipe" in variant:
        os.environ["BGAMD_TD_MFMA_MIN"] = "1"
    if "direct" in variant:
        os.environ["BGAMD_TD_DIRECT_MIN"] = "1"
    if "unfused" in variant:
        os.environ["BGAMD_TD_FUSED"] = "0"
    if "wide" in variant:
        os.environ["BGAMD_TD_WIDE_MIN"] = "1"
    try:
        mk = lambda: DeviceTDLambdaLearner(weights, max_games=n, alpha=0.1, lam=0.8)
        learners = [mk() for _ in range(3)]
    finally:
        os.environ.pop("BGAMD_TD_MFMA_MIN", None); os.environ.pop("BGAMD_TD_WIDE_MIN", None); os.environ.pop("BGAMD_TD_DIRECT_MIN", None); os.environ.pop("BGAMD_TD_FUSED", None)
    for slots in (1, 7, 64):
        # the float64 host closed form depends on the round and the slot count only, not on the kernel variant: computed once per session
        # (the digest of the log is part of the key: another round would be another entry)
        key = (slots, hash(_np(rows).tobytes()), hash(_np(lengths).tobytes()))
        if key not in _STREAM_CLOSED_FORM:
            Lc = TDLambdaLearner(weights, device="cpu", alpha=0.1, lam=0.8, dtype=torch.float64)
            sq_c, cnt_c = Lc.replay_stream(Xr, lengths.cpu(), p1_won.cpu(), slots=slots, batch_scale=0.3)
            _STREAM_CLOSED_FORM[key] = (Lc, sq_c, cnt_c)
        Lc, sq_c, cnt_c = _STREAM_CLOSED_FORM[key]
        out = []
        for Ld in learners[:2]:
            Ld.set_weights(weights)
            sq_d, cnt_d = Ld.replay_rows(rows, lengths, p1_won, batch_scale=0.3, slots=slots)
            out.append(_np(Ld.theta))
            assert cnt_d == cnt_c == turns and abs(sq_d - sq_c) < 1e-3 * max(1.0, sq_c)
        assert np.array_equal(out[0], out[1])
        moved = np.abs(Lc.theta.numpy() - weights).max()
        d = np.abs(out[0] - Lc.theta.numpy()).max()
        print("%s, %d slots: weights moved by %.3g, device - host fp64 = %.3g" % (variant, slots, moved, d))
        assert moved > 1e-3 and d < 2e-5 * max(1.0, moved), (slots, d, moved)
        Ld = learners[2]                                        # the distributed route on one rank
        Ld.set_weights(weights)
        sq_s, cnt_s = Ld.replay_rows(rows, lengths, p1_won, batch_scale=0.3, slots=slots, split_apply=True)
        assert cnt_s == turns and np.array_equal(_np(Ld.theta), out[0])
    a, b = learners[0], learners[1]
    a.set_weights(weights); b.set_weights(weights)
    a.replay_rows(rows, lengths, p1_won, batch_scale=0.3, slots=n)          # a slot per game ...
    b.replay_rows(rows, lengths, p1_won, batch_scale=0.3)                   # ... is the lock-step replay
    assert np.abs(_np(a.theta) - _np(b.theta)).max() < 2e-6


def test_streamed_replay_at_scale_matches_host_closed_form(bg, weights):
    """The streamed replay on the kernels a large round runs on (matrix-pipe forward pass from 512 running slots, whole-row trace
    workgroups from 8 192): 12 288 games through 3 072 and through 8 192 slots against the float64 closed form of the same
    schedule (PyTorch on the GPU)."""
    from backgammon_env.learner import DeviceTDLambdaLearner, TDLambdaLearner, play_round
    n = 12288
    env = bg.VecGame(n, seed=99)
    env.load_weights(weights)
    rows, lengths, p1_won = play_round(env, max_plies=300, epsilon=0.1)
    lengths = lengths.clone()
    lengths[::13] = 0
    turns = int(lengths.sum().item())
    X = env.encode_rows(rows)                                   # float32 [T, n, 198], exact
    Ld = DeviceTDLambdaLearner(weights, max_games=n, alpha=0.1, lam=0.75)
    for slots in (3072, 8192):
        Lc = TDLambdaLearner(weights, device="cuda", alpha=0.1, lam=0.75, dtype=torch.float64)
        sq_c, cnt_c = Lc.replay_stream(X, lengths, p1_won, slots=slots, batch_scale=24.0 / slots)
        Ld.set_weights(weights)
        sq_d, cnt_d = Ld.replay_rows(rows, lengths, p1_won, batch_scale=24.0 / slots, slots=slots)
        th_c, th_d = _np(Lc.theta), _np(Ld.theta).astype(np.float64)
        moved, d = np.abs(th_c - weights).max(), np.abs(th_d - th_c).max()
        print("%d games through %d slots: weights moved by %.3g, device - host fp64 = %.3g, TD error sums %.6f / %.6f"
              % (int((lengths > 0).sum()), slots, moved, d, sq_d, sq_c))
        assert cnt_d == cnt_c == turns and abs(sq_d - sq_c) < 1e-4 * sq_c
        assert moved > 1e-3 and d < 2e-5 * max(1.0, moved)


def test_lazily_scaled_traces_equal_the_ordinary_pass(bg, weights):
    """The stored trace is e / c with one scale c = Π λ per replay, so columns whose feature is zero at a step are read but
    not written.  BGAMD_TD_LAZY=0 runs e <- λ e + ∇ on every column at every step (round 1's arithmetic to the bit).  Same
    log: the weights agree to fp32 rounding, for a λ whose scale is folded back in every ~100 steps and for one that needs
    it every ~9 steps; fewer columns are written than read."""
    from backgammon_env.learner import DeviceTDLambdaLearner, play_round
    n = 2048
    env = bg.VecGame(n, seed=77)
    env.load_weights(weights)
    rows, lengths, p1_won = play_round(env, max_plies=400, epsilon=0.1)
    for lam in (0.75, 0.04):
        out = {}
        for mode in ("lazy", "ordinary"):
            if mode == "ordinary":
                os.environ["BGAMD_TD_LAZY"] = "0"
            try:
                L = DeviceTDLambdaLearner(weights, max_games=n, alpha=0.1, lam=lam)
            finally:
                os.environ.pop("BGAMD_TD_LAZY", None)
            sq, cnt = L.replay_rows(rows, lengths, p1_won, batch_scale=0.02)
            out[mode] = (_np(L.theta).astype(np.float64), sq, cnt, L.active_columns(), L.written_columns())
        moved = np.abs(out["ordinary"][0] - weights).max()
        err = np.abs(out["lazy"][0] - out["ordinary"][0]).max()
        print("lambda %.2f: max |theta_lazy - theta_ordinary| = %.3g (weights moved by up to %.3g); columns read %d, written %d (ordinary: %d)"
              % (lam, err, moved, out["lazy"][3], out["lazy"][4], out["ordinary"][4]))
        assert moved > 1e-3 and err < 2e-6
        assert abs(out["lazy"][1] - out["ordinary"][1]) <= 1e-5 * out["ordinary"][1] and out["lazy"][2] == out["ordinary"][2]
        assert out["ordinary"][4] == out["ordinary"][3] == out["lazy"][3]
        assert out["lazy"][4] < 0.75 * out["lazy"][3]


def test_sparse_traces_equal_dense_traces_bit_for_bit(bg, weights):
    """The learner touches only the W1 trace columns of features that have been non-zero in a game so far (the others
    are exactly zero).  BGAMD_TD_DENSE=1 makes every column active from the first step -- the dense pass.  Same log,
    same hyper-parameters: identical weights bit for bit, identical TD errors, ~0.4 of the columns touched."""
    from backgammon_env.learner import DeviceTDLambdaLearner, play_round
    n = 3000
    env = bg.VecGame(n, seed=2025)
    env.load_weights(weights)
    rows, lengths, p1_won = play_round(env, max_plies=400, epsilon=0.1)
    lengths = lengths.clone()
    lengths[::11] = 0
    out = {}
    for mode in ("sparse", "dense"):
        if mode == "dense":
            os.environ["BGAMD_TD_DENSE"] = "1"
        try:
            L = DeviceTDLambdaLearner(weights, max_games=n, alpha=0.1, lam=0.8)
        finally:
            os.environ.pop("BGAMD_TD_DENSE", None)
        sq, cnt = L.replay_rows(rows, lengths, p1_won, batch_scale=0.01)
        out[mode] = (_np(L.theta), sq, cnt, L.active_columns())
        sq2, cnt2 = L.replay_rows(rows, lengths, p1_won, batch_scale=0.01, split_apply=True)     # second replay on used buffers
        out[mode + "2"] = (_np(L.theta), sq2, cnt2)
    assert np.array_equal(out["sparse"][0], out["dense"][0]) and out["sparse"][1] == out["dense"][1]
    assert np.array_equal(out["sparse2"][0], out["dense2"][0])
    assert out["dense"][3] == 198 * out["dense"][2]
    frac = out["sparse"][3] / out["dense"][3]
    print("active W1 trace columns: %.3f of the dense pass" % frac)
    assert 0.2 < frac < 0.6 and np.abs(out["sparse"][0] - weights).max() > 1e-4
