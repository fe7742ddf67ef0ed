"""Round-4 GPU tests (all through the C ABI):
  * fixture G9: the reference's MULTI-GAME order of TD(lambda) updates (train.py:536-547: traces reset per game, the schedule
    between the games) against the device learner -- one game per call with `update_learning_params` in between, sub-rounds of one
    game, a streamed replay through one slot;
  * continuous self-play (bgamd_env_set_trajectory_ring): every game of the ring log is the game `play_round(episode=k)` plays,
    turn for turn; the streamed replay over the game table (bgamd_td_begin_stream_games) equals the replay of the same games from a
    per-lane log bit for bit, across the ring's wrap-around;
  * a replay on its own stream and host thread beside an env at play (the pipelined training loop) leaves the same weights and the
    same games as one after the other;
  * the step's one collective issued by the library (bgamd_td_replay_allreduce, an RCCL communicator of one rank) against the local
    route and the Python-driven split route: identical weights;
  * a pooled scalar Game dropped with work queued on a side stream.
Integer work is bit-exact; learner weights against the reference within 2e-6 per game (fixture G6's bar)."""
import gc
import os
import threading

import numpy as np
import pytest

from test_gpu_parity import _np

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def bg():
    import backgammon_env
    return backgammon_env


# ---- fixture G9 -----------------------------------------------------------------------------------------------------------------

def _g9_rows(bg, golden_dir):
    g = np.load(os.path.join(golden_dir, "g9_td_lambda_multi_game.npz"))
    off = g["off"]
    G, T = len(off) - 1, int(np.diff(off).max())
    rows = torch.zeros((T, G, 8), dtype=torch.int32, device="cuda")
    for k in range(G):
        st, turn = g["states"][off[k]:off[k + 1]].astype(np.int32), g["turn"][off[k]:off[k + 1]].astype(np.int32)
        rows[:len(st), k] = bg.pack_rows(st, turn)
    return g, rows, np.diff(off).astype(np.int32), g["winner"] == 0


def test_g9_reference_multi_game_order_device_learner(bg, golden_dir, weights):
    """The reference's round loop over 8 games (fixture G9, written by the unmodified apply_td_updates) through the HIP learner:
    (a) one game per replay with update_learning_params between the games -- after EVERY game within 2e-6 of the reference's weights;
    (b) replay_rows(sub_round=1) == the reference applying the games by decreasing length; (c) replay_rows(slots=1) == the reference
    applying them in the one slot's queue order.  The two orders end 5.7e-2 apart: an order mix-up cannot pass."""
    from backgammon_env.learner import DeviceTDLambdaLearner
    g, rows, lengths, won = _g9_rows(bg, golden_dir)
    G = len(lengths)
    L = DeviceTDLambdaLearner(weights, max_games=16)
    base = int(g["base_episode"])
    worst = 0.0
    for k in range(G):
        L.update_learning_params(base + k + 1)
        assert (L.learning_rate, L.lambda_decay) == tuple(g["alpha_lambda_sched"][k])
        ln = np.zeros(G, dtype=np.int32); ln[k] = lengths[k]
        sq, cnt = L.replay_rows(rows, ln, won)
        assert cnt == lengths[k]
        err = float(np.abs(_np(L.theta) - g["w_sched"][k]).max())
        worst = max(worst, err)
        assert err < 2e-6, (k, err)
    alpha, lam = g["alpha_lambda_fixed"]
    L = DeviceTDLambdaLearner(weights, max_games=16, alpha=alpha, lam=lam)
    sq, cnt = L.replay_rows(rows, lengths, won, sub_round=1)
    e_sorted = float(np.abs(_np(L.theta) - g["w_sorted_final"]).max())
    assert cnt == int(lengths.sum()) and e_sorted < 2e-6, e_sorted
    # ... and half way: the first four games of that order alone
    first4 = g["order_sorted"][:4]
    ln = np.zeros(G, dtype=np.int32); ln[first4] = lengths[first4]
    L = DeviceTDLambdaLearner(weights, max_games=16, alpha=alpha, lam=lam)
    L.replay_rows(rows, ln, won, sub_round=1)
    assert np.abs(_np(L.theta) - g["w_sorted_mid"]).max() < 2e-6
    L = DeviceTDLambdaLearner(weights, max_games=16, alpha=alpha, lam=lam)
    sq, cnt = L.replay_rows(rows, lengths, won, slots=1)
    e_stream = float(np.abs(_np(L.theta) - g["w_stream1_final"]).max())
    assert cnt == int(lengths.sum()) and e_stream < 2e-6, e_stream
    assert np.abs(_np(L.theta) - g["w_sorted_final"]).max() > 1e-3
    print(f"G9: per game with the schedule {worst:.2e}, sub_round=1 {e_sorted:.2e}, slots=1 {e_stream:.2e} from the reference's weights")


# ---- continuous self-play ---------------------------------------------------------------------------------------------------------

def _tables_by_lane(table):
    lane, start, length, won = [_np(x) for x in table]
    out = {}
    for i in range(len(lane)):
        out.setdefault(int(lane[i]), []).append((int(start[i]), int(length[i]), bool(won[i])))
    return out


def test_continuous_selfplay_ring_log_holds_the_games_play_round_plays(bg, weights):
    """512 lanes, epsilon-greedy, fixed weights, eight windows of 70 steps into a ring of 512 slots (the eighth window wraps): game j of a
    lane in the ring log -- its turns, its length, its winner -- is the game play_round(episode=j) logs for that lane, turn for turn.
    The dice, the opening roll and the exploration draws are functions of (lane, episode, ply): lanes out of step change nothing."""
    from backgammon_env.learner import ContinuousSelfPlay, play_round
    n, K, R, W = 512, 70, 512, 8
    env = bg.VecGame(n, seed=77)
    env.load_weights(weights)
    sp = ContinuousSelfPlay(env, ring_steps=R)
    ring_games = {}                                                   # lane -> [(rows [len, 8], won)] in the order they ended
    wrapped = 0
    for w in range(W):
        sp.play(K, epsilon=0.1)
        table = sp.finished()
        assert sp.dropped == 0                                        # (a game of more than 442 turns would be: none among these)
        rows = _np(sp.rows)
        for lane, games in _tables_by_lane(table).items():
            for start, length, won in games:
                idx = (start + np.arange(length)) % R
                wrapped += int(start + length > R)
                ring_games.setdefault(lane, []).append((rows[idx, lane].copy(), won))
    assert env.trajectory_step() == W * K and env.stats()["error_flags"] == 0 and wrapped > 0
    n_games = sum(len(v) for v in ring_games.values())
    assert n_games > 4 * n
    sp.close()
    env2 = bg.VecGame(n, seed=77)
    env2.load_weights(weights)
    checked = 0
    for epi in range(max(len(v) for v in ring_games.values())):
        rows, lengths, won = play_round(env2, max_plies=600, epsilon=0.1, episode=epi)
        rows, lengths, won = _np(rows), _np(lengths), _np(won)
        for lane, games in ring_games.items():
            if epi < len(games):
                r, w = games[epi]
                assert len(r) == lengths[lane] and w == bool(won[lane]), (lane, epi)
                assert np.array_equal(r, rows[:len(r), lane]), (lane, epi)
                checked += 1
    assert checked == n_games
    print(f"continuous self-play: {n_games} games of {n} lanes in {W * K} steps ({wrapped} across the ring's wrap) equal play_round's, turn for turn")


def test_replay_over_the_game_table_equals_the_per_lane_replay(bg, weights):
    """The streamed replay over a ring log + game table (bgamd_td_begin_stream_games) against replay_rows(slots=k) on the SAME games
    copied into a one-game-per-lane log: the same schedule, the same kernels -> the same weights bit for bit, at 1 / 7 / 600 slots
    (VALU forward, and the fused MFMA step), with games that cross the ring's wrap-around."""
    from backgammon_env.learner import ContinuousSelfPlay, DeviceTDLambdaLearner
    n, R = 1024, 512
    env = bg.VecGame(n, seed=5)
    env.load_weights(weights)
    sp = ContinuousSelfPlay(env, ring_steps=R)
    for _ in range(5):                                                # steps 0 .. 399: not replayed here
        sp.play(80, epsilon=0.05)
        sp.finished()
    sp.play(40, epsilon=0.05)
    sp.play(40, epsilon=0.05)                                         # two run_greedy calls, one window: steps 400 .. 479
    t1 = sp.finished()
    sp.play(80, epsilon=0.05)                                         # steps 480 .. 559: wraps at 512
    t2 = sp.finished()
    assert sp.dropped <= 1
    for table in (t1, t2):
        lane, start, length, won = table
        G = int(lane.numel())
        assert G > 200
        T = int(length.max().item())
        crosses = int(((start + length) > R).sum().item())
        # the same games as a per-lane log: column i = game i
        k = torch.arange(T, device="cuda")[:, None]
        idx = (start[None, :].long() + k) % R                         # [T, G]
        flat = sp.rows[idx, lane[None, :].long().expand(T, G)]        # [T, G, 8]
        flat = torch.where((k < length[None, :])[:, :, None], flat, torch.zeros_like(flat)).contiguous()
        for slots in (1, 7, 600):
            A = DeviceTDLambdaLearner(weights, max_games=1024, alpha=0.1, lam=0.8)
            sa, ca = A.replay_games(sp.rows, lane, start, length, won, slots=slots, batch_scale=min(1.0, 24.0 / slots))
            B = DeviceTDLambdaLearner(weights, max_games=1024, alpha=0.1, lam=0.8)
            sb, cb = B.replay_rows(flat, length, won, slots=slots, batch_scale=min(1.0, 24.0 / slots))
            assert ca == cb == int(length.sum().item()) and sa == sb
            assert torch.equal(A.theta, B.theta), slots
            assert (A.theta - torch.as_tensor(weights, device="cuda")).abs().max().item() > 1e-4
        if table is t2:
            assert crosses > 0                                        # some of these games do wrap around the ring
    print(f"replay over the game table == per-lane replay, bit for bit ({int(t1[0].numel())} + {int(t2[0].numel())} games)")


def test_replay_beside_an_env_at_play_changes_nothing(bg, weights):
    """The pipelined training loop: the learner replays window w - 1 on its own stream from its own host thread WHILE the env plays window
    w.  Against the same two things one after the other: the same weights bit for bit, the same games, the same ring."""
    from backgammon_env.learner import ContinuousSelfPlay, DeviceTDLambdaLearner
    n, K, R = 8192, 48, 512

    def run(overlap):
        env = bg.VecGame(n, seed=9)
        env.load_weights(weights)
        sp = ContinuousSelfPlay(env, ring_steps=R)
        L = DeviceTDLambdaLearner(weights, max_games=1024, alpha=0.1, lam=0.7)
        side = torch.cuda.Stream()
        out = {}

        def replay(table):
            torch.cuda.set_device(0)
            with torch.cuda.stream(side):
                out["r"] = L.replay_games(sp.rows, *table, slots=512, batch_scale=24.0 / 512)

        sp.play(K, epsilon=0.1)
        pending = sp.finished(keep_margin=K)
        for w in range(3):
            th = threading.Thread(target=replay, args=(pending,))
            if overlap:
                th.start()
            sp.play(K, epsilon=0.1)
            table = sp.finished(keep_margin=K)
            if not overlap:
                th.start()
            th.join()
            side.synchronize()
            pending = table
        torch.cuda.synchronize()
        assert env.stats()["error_flags"] == 0
        return L.theta.clone(), sp.rows.clone(), sp.end.clone(), env.states().clone(), out["r"]

    a, b = run(False), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and a[4] == b[4]
    assert (a[0] - torch.as_tensor(weights, device="cuda")).abs().max().item() > 1e-4


# ---- the step's collective issued by the library --------------------------------------------------------------------------------

def test_in_library_allreduce_world_of_one_equals_the_local_and_split_routes(bg, weights, monkeypatch):
    """bgamd_td_replay_allreduce with an RCCL communicator of ONE rank (bgamd_td_comm_init: the library resolves librccl at run time):
    [step kernels -> ncclAllReduce in place -> apply kernel] per training step on the learner's stream, against the local route (the
    reduce kernel applies the update) and the Python-driven split route (td_step -> td_apply): identical weights, streamed and lock-step."""
    from backgammon_env.learner import DeviceTDLambdaLearner, play_round
    n = 4096
    env = bg.VecGame(n, seed=21)
    env.load_weights(weights)
    rows, lengths, won = play_round(env, max_plies=400, epsilon=0.05)
    monkeypatch.delenv("BGAMD_FORCE_COLLECTIVE", raising=False)

    def fresh():
        return DeviceTDLambdaLearner(weights, max_games=n, alpha=0.1, lam=0.7)

    for kw in (dict(slots=256, batch_scale=24.0 / 256), dict(slots=1024, batch_scale=24.0 / 1024), dict(sub_round=2048, batch_scale=24.0 / 2048)):
        A = fresh()
        ra = A.replay_rows(rows, lengths, won, **kw)
        B = fresh()
        rb = B.replay_rows(rows, lengths, won, split_apply=True, **kw)
        C = fresh()
        C.init_collective()
        assert C._comm == (0, 1)
        monkeypatch.setenv("BGAMD_FORCE_COLLECTIVE", "1")
        rc = C.replay_rows(rows, lengths, won, **kw)
        monkeypatch.delenv("BGAMD_FORCE_COLLECTIVE")
        assert ra == rb == rc
        assert torch.equal(A.theta, B.theta) and torch.equal(A.theta, C.theta), kw
        assert (A.theta - torch.as_tensor(weights, device="cuda")).abs().max().item() > 1e-4
        del C
        gc.collect()


# ---- pooled scalar games -------------------------------------------------------------------------------------------

def test_pooled_game_dropped_with_work_queued_on_a_side_stream(bg, weights):
    """A Game dropped right after make_move inside `with torch.cuda.stream(s)` -- its step still queued on s, no getter called -- hands its
    one-lane env back to the pool; the next Game() takes that env and resets it on the NULL stream.  The reset must not overtake the
    queued step: the new Game starts from the start position, every time."""
    from backgammon_env.policy import TDLGammonModel
    m = TDLGammonModel()
    m.load_flat(weights)
    start = bg.Game(0).getGameBoard()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(24):
            g = bg.Game(i % 2)
            assert g.getGameBoard() == start and g.getJailedCount(0) == 0 and g.getBornOffCount(1) == 0 and g.getTurn() == i % 2
            g.setDice(3, 1)
            m.make_move(g)                                            # step_greedy on s; no getter afterwards
            del g
            gc.collect()
    s.synchronize()
    g = bg.Game(0)
    assert g.getGameBoard() == start


# ---- the root pass inside the boundary launch ----------------------------------------------------------------------------------------

def _root_pass_inside_the_boundary_launch(bg, weights, monkeypatch, n, force):
    """Inside a run the root pass of step t + 1 runs in the boundary launch of step t (boundary_kernel<true>, bg_root_resident.h): against
    single steps (step_greedy: apply, roots and the stand-alone root pass) and -- force=True, experimental build: BGAMD_ROOT_IN_BOUNDARY=1 / 0
    force either structure at any env size -- against the same run with the root pass as a launch of its own every step (rounds 1-3): the
    same games, values and counters to the last bit, on lane counts that leave partial tiles and partial workgroups, with exploration,
    across run boundaries and with the turn log on."""
    if force:
        monkeypatch.setenv("BGAMD_ROOT_IN_BOUNDARY", "1")    # (the default from 24 576 lanes; below, the pass stays a launch of its own: forced here)
    a = bg.VecGame(n, seed=808)
    if force:
        monkeypatch.setenv("BGAMD_ROOT_IN_BOUNDARY", "0")
    b = bg.VecGame(n, seed=808)
    monkeypatch.delenv("BGAMD_ROOT_IN_BOUNDARY", raising=False)
    c = bg.VecGame(n, seed=808)
    for e in (a, b, c):
        e.load_weights(weights)
    ta, tb = a.record_trajectory(64), b.record_trajectory(64)
    for k in (1, 2, 9, 30):
        a.run_greedy(k, epsilon=0.05); b.run_greedy(k, epsilon=0.05)
        for _ in range(k):
            c.step_greedy(epsilon=0.05)
        for x in (b, c):
            assert torch.equal(a.states(), x.states()) and torch.equal(a.turns(), x.turns()) and torch.equal(a.dice(), x.dice()), k
            la, lx = a.last_choice(), x.last_choice()
            assert torch.equal(la["value"], lx["value"]) and torch.equal(la["seq"], lx["seq"]), k
    assert torch.equal(ta, tb)
    assert a.stats() == b.stats() == c.stats() and a.stats()["error_flags"] == 0
    assert a.kernel_choice()["root"] == "inside boundary_kernel<true>" and not a.kernel_choice()["root_on_second_stream"]
    assert c.kernel_choice()["root"] == "root_hidden_resident_kernel"
    if force:
        assert b.kernel_choice()["root"] == "root_hidden_resident_kernel"


def test_root_pass_inside_the_boundary_launch_is_bit_identical(bg, weights, monkeypatch):
    """33 000 lanes: the default puts the pass inside the boundary launch (from 24 576 lanes); against single steps."""
    _root_pass_inside_the_boundary_launch(bg, weights, monkeypatch, 33000, force=False)


# ---- the expansion below the roots in one launch: against rounds 1-4's two launches (experimental build: tests/test_gpu_experimental.py) ------

def _expansion_in_one_launch_plays_the_same_games(bg, weights, monkeypatch, n):
    """expand_all_kernel (the default): the doubles turns' plies 2 and 3 AND their leaf stage on the first workgroups of a launch whose
    other workgroups are the non-doubles leaf stage -- against doubles_kernel + expand_kernel<LEAF> (BGAMD_EXPAND_MERGED=0, rounds 1-4):
    the rows are the same rows in another order of the arena, so the games, the chosen values and sequences, the turn log and every
    counter are equal to the last bit; the staged rows of a step are the same multiset; with exploration, across run boundaries, at
    two shares of doubles workgroups."""
    monkeypatch.setenv("BGAMD_EXPAND_MERGED", "1")
    a = bg.VecGame(n, seed=4242)
    monkeypatch.setenv("BGAMD_EXPAND_DBL_PCT", "25")
    c = bg.VecGame(n, seed=4242)
    monkeypatch.delenv("BGAMD_EXPAND_DBL_PCT", raising=False)
    monkeypatch.setenv("BGAMD_EXPAND_MERGED", "0")
    b = bg.VecGame(n, seed=4242)
    monkeypatch.delenv("BGAMD_EXPAND_MERGED", raising=False)
    for e in (a, b, c):
        e.load_weights(weights)
    ta, tb = a.record_trajectory(96), b.record_trajectory(96)
    for k in (1, 2, 9, 30, 50):
        for e in (a, b, c):
            e.run_greedy(k, epsilon=0.05)
        for x in (b, c):
            assert torch.equal(a.states(), x.states()) and torch.equal(a.turns(), x.turns()) and torch.equal(a.dice(), x.dice()), k
            la, lx = a.last_choice(), x.last_choice()
            assert torch.equal(la["value"], lx["value"]) and torch.equal(la["seq"], lx["seq"]), k
        # the rows the value net saw in the last step: the same (game, key, afterstate, value) records, in another order of the arena
        rec = []
        for e in (a, b):
            info, st, val = e.unique_rows()
            r = torch.cat([info, st.to(torch.int64), val.view(torch.int32).to(torch.int64).view(-1, 1)], 1)
            order = torch.argsort(info[:, 0] * (1 << 32) + info[:, 1], stable=True)
            rec.append(r[order])
        assert rec[0].shape == rec[1].shape and torch.equal(rec[0], rec[1]), k
    assert torch.equal(ta, tb)
    assert a.stats() == b.stats() == c.stats() and a.stats()["error_flags"] == 0
    assert a.kernel_choice()["expand"] == "expand_all_kernel" and b.kernel_choice()["expand"] == "doubles_kernel + expand_kernel<LEAF>"


def _expansion_in_one_launch_odd_env_sizes(bg, weights, monkeypatch, n):
    """Env sizes around the launch's grid rules (one lane; a wave more or less; workgroup counts that do and do not divide by the four
    list parts; the size from which the root pass moves into the boundary launch): one launch against two, f32 and bf16, 45 steps with
    exploration -- the same games and counters."""
    monkeypatch.delenv("BGAMD_EXPAND_MERGED", raising=False)
    a = bg.VecGame(n, seed=97 + n)
    monkeypatch.setenv("BGAMD_EXPAND_MERGED", "0")
    b = bg.VecGame(n, seed=97 + n)
    monkeypatch.delenv("BGAMD_EXPAND_MERGED", raising=False)
    a.load_weights(weights); b.load_weights(weights)
    for prec in (bg.F32, bg.BF16):
        for k in (1, 14, 30):
            a.run_greedy(k, epsilon=0.05, precision=prec); b.run_greedy(k, epsilon=0.05, precision=prec)
            assert torch.equal(a.states(), b.states()) and torch.equal(a.turns(), b.turns()) and torch.equal(a.dice(), b.dice()), (n, prec, k)
            assert torch.equal(a.last_choice()["value"], b.last_choice()["value"]), (n, prec, k)
    assert a.stats() == b.stats() and a.stats()["error_flags"] == 0


# ---- the delayed update: a training step in one launch -----------------------------------------------------------------------------

def test_delayed_update_replay_matches_the_delayed_closed_form(bg, weights):
    """bgamd_td_set_delay(1): a streamed replay through 512 / 1 024 / 2 048 / 4 096 slots applies the update of step t one step late, the
    reduction of step t - 1 riding on the idle waves of step t's launch (td_step_fused_kernel<., ., DELAY>).  Against the float64 closed form
    of the SAME delayed schedule (TDLambdaLearner.replay_stream(delay=1)) within the exact replay's bound; the exact closed form is measurably
    another trajectory; twice bit-identical; over a game table / ring log as over per-lane rows; a replay that does not qualify (64 slots)
    is the exact one bit for bit."""
    from backgammon_env.learner import DeviceTDLambdaLearner, TDLambdaLearner, play_round
    n = 12288
    env = bg.VecGame(n, seed=99)
    env.load_weights(weights)
    rows, lengths, p1_won = play_round(env, max_plies=400, epsilon=0.1)
    X = env.encode_rows(rows)
    for slots in (2048, 1024, 512, 4096):
        scale = 48.0 / slots
        L = DeviceTDLambdaLearner(weights, max_games=n, alpha=0.1, lam=0.8)
        L.set_delay(1)
        out = []
        for rep in range(2):
            L.set_weights(weights)
            sq, cnt = L.replay_rows(rows, lengths, p1_won, batch_scale=scale, slots=slots)
            out.append((_np(L.theta).copy(), sq, cnt))
        assert np.array_equal(out[0][0], out[1][0]) and out[0][1:] == out[1][1:] and out[0][2] == int(_np(lengths).sum())
        Ld = TDLambdaLearner(weights, device="cuda", alpha=0.1, lam=0.8, dtype=torch.float64)
        Ld.replay_stream(X, lengths, p1_won, slots=slots, batch_scale=scale, delay=1)
        Le = TDLambdaLearner(weights, device="cuda", alpha=0.1, lam=0.8, dtype=torch.float64)
        Le.replay_stream(X, lengths, p1_won, slots=slots, batch_scale=scale, delay=0)
        th_d, th_e = _np(Ld.theta), _np(Le.theta)
        moved = np.abs(th_d - weights).max()
        gap, gap_exact = np.abs(out[0][0] - th_d).max(), np.abs(out[0][0] - th_e).max()
        print("delayed update through %d slots: max |theta - delayed float64 closed form| = %.3g (weights moved %.3g); from the EXACT closed form %.3g"
              % (slots, gap, moved, gap_exact))
        assert moved > 1e-3 and gap < 2e-4 * max(1.0, moved)
        assert gap_exact > 5 * gap                                    # the delay is visible: it is the delayed schedule that was replayed
    # a replay that does not take the one-launch step is the exact replay
    A = DeviceTDLambdaLearner(weights, max_games=n, alpha=0.1, lam=0.8)
    A.set_delay(1)
    A.replay_rows(rows, lengths, p1_won, batch_scale=0.3, slots=64)
    B = DeviceTDLambdaLearner(weights, max_games=n, alpha=0.1, lam=0.8)
    B.replay_rows(rows, lengths, p1_won, batch_scale=0.3, slots=64)
    assert torch.equal(A.theta, B.theta)
