"""TD(λ) learner math (host side, torch): one game through the closed-form lock-step replay equals the
reference's apply_td_updates (fixture G6, produced by the reference learner) -- CPU only."""
import os

import numpy as np
import torch

from oracle import oracle as O


def test_g6_single_game_matches_reference(golden_dir, weights):
    from backgammon_env.learner import TDLambdaLearner
    g = np.load(os.path.join(golden_dir, "g6_td_lambda.npz"))
    st, turn = g["states"].astype(np.int32), g["turn"]
    X = np.stack([O.encode(st[i:i + 1], int(turn[i]))[0] for i in range(len(st))])[:, None, :]   # [T,1,198]
    alpha, lam = g["alpha_lambda"]
    L = TDLambdaLearner(weights, alpha=alpha, lam=lam)
    sq, cnt = L.replay(torch.from_numpy(X), [len(st)], [int(g["winner"][0]) == 0])
    assert cnt == len(st)
    w_after = L.theta.numpy()
    assert np.abs(w_after - g["w_after"]).max() < 2e-6
    assert np.abs(g["w_after"] - weights).max() > 1e-4          # the fixture actually moved the weights
    # the reference reports the squared TD errors of all but the terminal step; ours adds the terminal one (<= 1)
    assert 0.0 <= sq - float(np.sum(g["losses"])) < 1.0


def test_batched_equals_sum_of_single_games_first_step(weights):
    """Lock-step semantics: after the FIRST step the batched update equals the sum of the per-game updates."""
    from backgammon_env.learner import TDLambdaLearner
    rng = np.random.RandomState(0)
    lanes = [O.lane_run(5, lane, 8, 40, 0)[0] for lane in range(3)]
    X = np.stack([[O.encode(l[t:t + 1, :28], int(l[t, 28]))[0] for l in lanes] for t in range(2)])   # [2,3,198]
    Lb = TDLambdaLearner(weights, alpha=0.1, lam=0.9)
    Lb.replay(torch.from_numpy(X[:1]), [5, 5, 5], [1, 0, 1])          # T=1 slice: one step, v_next = 0 branch unused
    acc = np.zeros(25601, dtype=np.float64)
    for k in range(3):
        L1 = TDLambdaLearner(weights, alpha=0.1, lam=0.9)
        L1.replay(torch.from_numpy(X[:1, k:k + 1]), [5], [[1, 0, 1][k]])
        acc += (L1.theta.numpy().astype(np.float64) - weights)
    assert np.abs((Lb.theta.numpy() - weights) - acc).max() < 1e-6


def test_checkpoint_interop_state_dict_roundtrip(tmp_path, weights):
    """the learner writes the reference's 4-tensor state_dict (train.py:513-515) and the mirror
    policy class reads it back bit for bit (no GPU needed for the file format)."""
    from backgammon_env.learner import TDLambdaLearner
    from backgammon_env.policy import TDLGammonModel, flatten_state_dict
    L = TDLambdaLearner(weights)
    path = tmp_path / "ckpt.pth"
    torch.save(L.state_dict(), path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    assert {k: tuple(v.shape) for k, v in sd.items()} == {"fc1.weight": (128, 198), "fc1.bias": (128,),
                                                          "fc2.weight": (1, 128), "fc2.bias": (1,)}
    m = TDLGammonModel()
    m.load_state_dict(sd)
    assert np.array_equal(m._w, weights) and np.array_equal(flatten_state_dict(m.state_dict()), weights)


def test_oracle_td_restatement_matches_reference_fixture_and_host_learner(golden_dir, weights):
    """The numpy float64 restatement in oracle/ (the checker of the device learner) reproduces the reference's own
    update (fixture G6) and agrees with the host-side closed form on a ragged multi-game round."""
    from backgammon_env.learner import TDLambdaLearner
    g = np.load(os.path.join(golden_dir, "g6_td_lambda.npz"))
    st, turn = g["states"].astype(np.int32), g["turn"]
    X = np.stack([O.encode(st[i:i + 1], int(turn[i]))[0] for i in range(len(st))])[:, None, :]
    alpha, lam = g["alpha_lambda"]
    th, sq, cnt = O.td_lambda_lockstep(weights, X, [len(st)], [int(g["winner"][0]) == 0], alpha, lam)
    assert cnt == len(st) and np.abs(th - g["w_after"]).max() < 2e-6
    lanes = [O.lane_run(9, lane, 8, 30 + 3 * lane, 0)[0] for lane in range(5)]
    lengths = [30, 33, 0, 12, 1]
    T = 34
    Xm = np.zeros((T, 5, 198), dtype=np.float32)
    for k, l in enumerate(lanes):
        for t in range(min(T, len(l))):
            Xm[t, k] = O.encode(l[t:t + 1, :28], int(l[t, 28]))[0]
    tho, sqo, cnto = O.td_lambda_lockstep(weights, Xm, lengths, [1, 0, 1, 0, 1], 0.1, 0.9, batch_scale=0.5)
    L = TDLambdaLearner(weights, alpha=0.1, lam=0.9, dtype=torch.float64)
    sql, cntl = L.replay(torch.from_numpy(Xm).double(), lengths, [1, 0, 1, 0, 1], batch_scale=0.5)
    assert cntl == cnto == sum(lengths)
    assert np.abs(L.theta.numpy() - tho).max() < 1e-12 and abs(sql - sqo) < 1e-9


def test_streamed_replay_host_closed_form(weights):
    """The streamed schedule (slots take game after game): with one slot it IS the reference's order -- a round's games
    replayed one after another, traces reset per game, every later game from the weights the earlier ones left
    (train.py:536-547) -- and with a slot per game it is the lock-step replay."""
    import torch
    from backgammon_env.learner import TDLambdaLearner, stream_schedule
    g = torch.Generator().manual_seed(3)
    T, G = 12, 9
    X = (torch.rand(T, G, 198, generator=g) > 0.85).double()
    lengths = torch.tensor([5, 12, 0, 7, 1, 9, 3, 12, 4])
    won = torch.tensor([1, 0, 1, 1, 0, 0, 1, 0, 1], dtype=torch.bool)
    queue, qoff, n_steps, k = stream_schedule(lengths, 1)
    assert k == 1 and n_steps == int(lengths.sum()) and sorted(queue.tolist()) == [0, 1, 3, 4, 5, 6, 7, 8]
    a = TDLambdaLearner(weights, alpha=0.1, lam=0.7, dtype=torch.float64)
    sq_a, cnt_a = a.replay_stream(X, lengths, won, slots=1, batch_scale=0.5)
    b = TDLambdaLearner(weights, alpha=0.1, lam=0.7, dtype=torch.float64)
    sq_b = 0.0
    for lane in queue.tolist():                                  # one game at a time, in the slot's order
        s, _ = b.replay(X[:, lane:lane + 1], lengths[lane:lane + 1], won[lane:lane + 1], batch_scale=0.5)
        sq_b += s
    assert cnt_a == int(lengths.sum()) and abs(sq_a - sq_b) < 1e-12
    assert (a.theta - b.theta).abs().max() < 1e-13 and (a.theta - torch.as_tensor(weights).double()).abs().max() > 1e-4
    c = TDLambdaLearner(weights, alpha=0.1, lam=0.7, dtype=torch.float64)
    c.replay_stream(X, lengths, won, slots=64, batch_scale=0.5)
    d = TDLambdaLearner(weights, alpha=0.1, lam=0.7, dtype=torch.float64)
    d.replay(X, lengths, won, batch_scale=0.5)
    assert (c.theta - d.theta).abs().max() < 1e-13
    # three slots: every game exactly once, slot totals balanced
    queue, qoff, n_steps, k = stream_schedule(lengths, 3)
    tot = [int(lengths[queue[qoff[i]:qoff[i + 1]].long()].sum()) for i in range(3)]
    assert k == 3 and sorted(queue.tolist()) == [0, 1, 3, 4, 5, 6, 7, 8] and n_steps == max(tot) and max(tot) - min(tot) <= 4


def test_stream_schedule_library_equals_host_restatement():
    """bgamd_td_stream_schedule (host code of the library: what DeviceTDLambdaLearner streams by) against the Python restatement
    the host closed form uses: the same queue, offsets and step count on ragged rounds; slot totals balanced to a short game."""
    import ctypes as C
    import numpy as np
    import torch
    from backgammon_env import _capi
    from backgammon_env.learner import stream_schedule
    lib = _capi.load()
    rng = np.random.default_rng(5)
    for n, slots in ((1, 1), (40, 7), (300, 64), (5000, 128), (64, 200)):
        ln = np.maximum(0, (rng.gamma(2.0, 40.0, size=n) + 4).astype(np.int32) * (rng.random(n) > 0.1)).astype(np.int32)
        ln[rng.integers(0, n)] = 1
        queue, qoff = np.zeros(n, dtype=np.int32), np.zeros(slots + 1, dtype=np.int32)
        ng, ns = C.c_int64(), C.c_int64()
        assert lib.bgamd_td_stream_schedule(ln.ctypes.data, n, slots, queue.ctypes.data, qoff.ctypes.data, C.byref(ng), C.byref(ns)) == 0
        q, o, n_steps, k = stream_schedule(torch.from_numpy(ln), slots)
        games = int((ln > 0).sum())
        assert ng.value == games and ns.value == n_steps and k == min(slots, games)
        assert queue[:games].tolist() == q.tolist() and qoff[:k + 1].tolist() == o.tolist() and all(qoff[k:] == games)
        tot = [int(ln[queue[qoff[i]:qoff[i + 1]]].sum()) for i in range(k)]
        assert n_steps == max(tot)
        if games >= 4 * slots:
            assert max(tot) - min(tot) <= int(np.sort(ln[ln > 0])[: max(1, games // 4)].max()) + 1      # within a short game


def _g9(golden_dir):
    g = np.load(os.path.join(golden_dir, "g9_td_lambda_multi_game.npz"))
    off = g["off"]
    G, T = len(off) - 1, int(np.diff(off).max())
    X = np.zeros((T, G, 198), dtype=np.float32)
    for k in range(G):
        st, turn = g["states"][off[k]:off[k + 1]].astype(np.int32), g["turn"][off[k]:off[k + 1]]
        for t in range(len(st)):
            X[t, k] = O.encode(st[t:t + 1], int(turn[t]))[0]
    return g, X, np.diff(off).astype(np.int64), g["winner"] == 0


def test_g9_multi_game_order_matches_reference(golden_dir, weights):
    """Fixture G9 (tests/golden/make_golden_r4.py): the body of the reference's round loop (train.py:536-547) over 8 of G5's games --
    `update_learning_params` between the games (alpha steps 0.1 -> 0.096 at episode 40 000), traces reset per game, every game from the
    weights the one before left -- through the reference's own apply_td_updates.  The host closed form, one game per replay() call with the
    schedule in between, equals it after EVERY game (the bar of fixture G6, per game); the streamed replay through one slot and the
    one-game sub-rounds equal the fixture's runs in those two orders at fixed alpha / lambda."""
    from backgammon_env.learner import TDLambdaLearner
    g, X, lengths, won = _g9(golden_dir)
    Xt = torch.from_numpy(X)
    G = len(lengths)
    L = TDLambdaLearner(weights)
    base = int(g["base_episode"])
    prev = weights
    for k in range(G):
        L.update_learning_params(base + k + 1)
        assert (L.learning_rate, L.lambda_decay) == tuple(g["alpha_lambda_sched"][k])
        ln = np.zeros(G, dtype=np.int64); ln[k] = lengths[k]
        sq, cnt = L.replay(Xt, ln, won)
        assert cnt == lengths[k]
        assert np.abs(L.theta.numpy() - g["w_sched"][k]).max() < 2e-6, k
        assert np.abs(g["w_sched"][k] - prev).max() > 1e-4                       # every game moved the weights
        prev = g["w_sched"][k]
    assert g["alpha_lambda_sched"][0][0] != g["alpha_lambda_sched"][-1][0]        # the schedule did step inside the fixture
    alpha, lam = g["alpha_lambda_fixed"]
    # the order of a streamed replay through ONE slot
    L = TDLambdaLearner(weights, alpha=alpha, lam=lam)
    from backgammon_env.learner import stream_schedule
    queue, _, _, _ = stream_schedule(torch.from_numpy(lengths), 1)
    assert queue.tolist() == g["order_stream1"].tolist()
    L.replay_stream(Xt, lengths, won, slots=1)
    assert np.abs(L.theta.numpy() - g["w_stream1_final"]).max() < 2e-6
    # sub-rounds of one game: decreasing length (what DeviceTDLambdaLearner.replay_rows(sub_round=1) runs)
    L = TDLambdaLearner(weights, alpha=alpha, lam=lam)
    for n_done, k in enumerate(g["order_sorted"].tolist()):
        ln = np.zeros(G, dtype=np.int64); ln[k] = lengths[k]
        L.replay(Xt, ln, won)
        if n_done == 3:
            assert np.abs(L.theta.numpy() - g["w_sorted_mid"]).max() < 2e-6
    assert np.abs(L.theta.numpy() - g["w_sorted_final"]).max() < 2e-6
    # the two orders are different runs: an order mix-up cannot pass
    assert np.abs(g["w_sorted_final"] - g["w_stream1_final"]).max() > 1e-3
