"""Round-3 GPU tests (all through the C ABI):
  * fixture G8: row-level value-net outputs for positions with checkers ON THE BAR (roots with features 194 / 195 set,
    bar-entry moves), written by the unmodified reference while replaying fixture G5's games -- what fixture G7 could
    not reach through the binding's setters; both delta kernels (the VALU one the step uses, the opt-in MFMA one);
  * (the MFMA delta kernel, the register-resident dense f16 x 2 kernel and the LDS-staged root pass moved behind -DBGAMD_EXPERIMENTAL in
    round 4: their tests are tests/test_gpu_experimental.py, marker gpu_experimental);
  * the scalar surface behind one-lane pooled envs: weights follow the MODEL (not a recycled address), set_seed() decides
    the dice of every Game created afterwards, a Game works inside a non-default torch stream.
Integer work is bit-exact; value-net outputs within 1e-5 of the reference (north_star)."""
import gc
import os
import time

import numpy as np
import pytest

from test_gpu_parity import _np
from test_gpu_round2 import _rows_by_state

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def bg():
    import backgammon_env
    return backgammon_env


@pytest.fixture(params=["valu_delta"])
def delta_kernel(request, monkeypatch):
    """The env reads BGAMD_MFMA_DELTA when it is created (the MFMA delta kernel exists in the experimental build only:
    tests/test_gpu_experimental.py runs this test's other half)."""
    monkeypatch.setenv("BGAMD_MFMA_DELTA", "0")
    return request.param


def _fixture_step(bg, weights, g, delta_kernel):
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
    return env, info, st, val, n_rows, e32, e64


def test_bar_positions_rows_vs_reference_values(bg, golden_dir, weights, delta_kernel):
    """Fixture G8: 957 turns of the reference's greedy games whose root has a checker on the bar (584 with the MOVER on the
    bar: every move starts with a bar entry), every distinct afterstate with the value the reference model's forward pass
    gave it.  One greedy step in BGAMD_F32 on those turns: the rows handed to the value net are exactly the reference's
    distinct afterstates, every per-row output is within 1e-5 of the reference (fp32 and fp64), and the move each game
    then makes is value-optimal for the reference's own values."""
    g = np.load(os.path.join(golden_dir, "g8_bar_candidate_values.npz"))
    roots, off = g["roots"], g["off"]
    assert len(roots) >= 100 and ((roots[:, 24] > 0) | (roots[:, 25] > 0)).all()
    mover_on_bar = roots[np.arange(len(roots)), 24 + roots[:, 28]] > 0
    assert mover_on_bar.sum() >= 100
    env, info, st, val, n_rows, e32, e64 = _fixture_step(bg, weights, g, delta_kernel)
    print("%s, %d rows of %d bar turns: max |gpu - reference fp32| = %.3g, max |gpu - reference fp64| = %.3g"
          % (delta_kernel, n_rows, len(roots), e32, e64))
    assert n_rows >= int(off[-1]) and e32 < 1e-5 and e64 < 1e-5
    post = _np(env.states())
    for k in range(len(roots)):
        sl = slice(off[k], off[k + 1])
        j = np.nonzero((g["states"][sl] == post[k][None, :]).all(axis=1))[0]
        assert len(j) == 1, k                              # the applied state is one of the reference's afterstates
        v = g["v64"][sl]
        best = v.max() if roots[k, 28] == 0 else v.min()
        assert abs(v[j[0]] - best) < 1e-5, k


# ---- the scalar surface behind pooled one-lane envs (round-2 advisor findings) ---------------------------------------------

def test_pooled_game_plays_with_the_weights_of_its_model(bg, weights):
    """Envs outlive their Game (the one-lane pool) and keep device weights.  Model after model evaluated on pooled Games --
    the loop that scores checkpoint after checkpoint -- must play with ITS weights: the binding is keyed by a token that is
    never reused, not by id(model), which CPython hands to the next model once the last one is freed."""
    from backgammon_env.policy import TDLGammonModel
    rng = np.random.RandomState(3)
    variants = [weights] + [(weights + rng.normal(0, 0.05, weights.shape)).astype(np.float32) for _ in range(3)]
    seen_ids, choices = set(), []
    for w in variants:
        m = TDLGammonModel()
        m.load_flat(w)
        seen_ids.add(id(m))
        game = bg.Game(0)                                     # the start position; from the pool from the second model on
        game.setDice(3, 1)
        seq = m.make_move(game)
        value = float(game._v.last_choice()["value"][0])
        # what THIS model's weights say about the state the move led to
        post = np.array(game.getGameBoard() + [game.getJailedCount(0), game.getJailedCount(1), game.getBornOffCount(0),
                                               game.getBornOffCount(1)], dtype=np.int32)
        want = float(_np(m.values(post[None, :], 0))[0])
        assert abs(value - want) < 2e-6, (value, want)
        choices.append((tuple(seq), value))
        del game, m
        gc.collect()
    assert len({c[1] for c in choices}) == len(variants)      # four models, four different evaluations
    print("ids seen for 4 successive models:", len(seen_ids))


def test_set_seed_decides_the_dice_of_pooled_games(bg):
    """set_seed(s); create, destroy and create Games: the k-th Game created after set_seed rolls the same dice in two runs,
    whether its env is new or comes from the pool, and they are the dice of game id k - 1 on a fresh env of that seed."""
    def run():
        bg.set_seed(20260101)
        seqs = []
        a = bg.Game(0)
        seqs.append([tuple(a.roll_dice()) for _ in range(6)])
        del a
        gc.collect()                                          # a's env is back in the pool
        b = bg.Game(1)                                        # ... and comes out of it
        c = bg.Game(0)
        seqs.append([tuple(b.roll_dice()) for _ in range(6)])
        seqs.append([tuple(c.roll_dice()) for _ in range(6)])
        del b, c
        gc.collect()
        return seqs
    first = run()
    assert run() == first                                     # two runs: identical, although the pool holds more envs now
    assert len({tuple(s) for s in first}) == 3                # three games, three dice streams
    for k, want in enumerate(first):                          # == the unpooled stream of global game id k
        v = bg.VecGame(1, seed=20260101, lane_offset=k, lane_stride=1 << 40, arena_rows=32768)
        got = []
        for _ in range(6):
            v.roll(advance_ply=True)
            got.append(tuple(int(x) for x in _np(v.dice())[0]))
        assert got == want, k
    bg.set_seed(int.from_bytes(os.urandom(8), "little"))


def test_game_inside_a_side_stream(bg, weights):
    """The host-argument surface (bgamd_game_*) runs on the NULL stream, VecGame calls on torch's current stream, and torch's
    side streams do not synchronise with the NULL stream: a Game created and driven inside `with torch.cuda.stream(s)` must
    still see its own calls in order (reset before set_state, step_greedy before the snapshot that reads its result)."""
    from backgammon_env.policy import TDLGammonModel
    m = TDLGammonModel()
    m.load_flat(weights)
    ref = bg.Game(0)
    ref.setDice(6, 5)
    want_seq = m.make_move(ref)
    want_board = ref.getGameBoard()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(8):                                    # pooled envs, back to back
            g = bg.Game(0)
            assert g.getTurn() == 0 and g.getGameBoard()[0] == 2 and list(g.get_last_dice()) == [1, 1]
            g.setDice(6, 5)
            seq = m.make_move(g)
            assert seq == want_seq and g.getGameBoard() == want_board
            del g
    s.synchronize()


# ---- the learner's mid-sized training step (round 3: forward + pipelined trace pass in one launch) --------------------------------

def test_fused_training_step_equals_the_two_launch_step(bg, weights, monkeypatch):
    """A streamed replay through 256 ... 3 072 slots -- 1 024 and 2 048 are the step sizes the quality study allows -- with the round-3 step
    (td_step_fused_kernel: the forward pass of a chunk's 8 slots inside the workgroup that then runs their software-pipelined
    trace pass; td_reduce_kernel with 16-byte loads) against the same replay with the forward pass and the trace pass as two
    launches (BGAMD_TD_FUSE_STEP=0) and with round 2's unpipelined trace kernel on top (BGAMD_TD_PIPE=0): the same arithmetic in the
    same order, bit for bit; and against the float64 closed form of the same schedule."""
    from backgammon_env.learner import DeviceTDLambdaLearner, TDLambdaLearner, play_round
    n = 12288
    env = bg.VecGame(n, seed=99)
    env.load_weights(weights)
    rows, lengths, p1_won = play_round(env, max_plies=400, epsilon=0.1)
    assert int((lengths > 0).sum()) > 0.99 * n
    # slots per workgroup of the fused launch by step size: 2 048 -> 8 (the chunk of the two-launch route: same partial sums, same bits),
    # 1 024 -> 4, 512 -> 2, 3 072 -> 16 (a full 32-row tile), and 1 (256 slots, only with BGAMD_TD_FUSE_MIN lowered)
    for slots in (2048, 1024, 512, 3072, 256):
        scale = 48.0 / slots
        out = {}
        fused_env = {"BGAMD_TD_FUSE_MIN": "64"} if slots == 256 else {}
        for tag, envs in (("fused", fused_env), ("two_launches", {"BGAMD_TD_FUSE_STEP": "0"}), ("round2", {"BGAMD_TD_FUSE_STEP": "0", "BGAMD_TD_PIPE": "0"})):
            for k in ("BGAMD_TD_FUSE_STEP", "BGAMD_TD_PIPE", "BGAMD_TD_FUSE_MIN"):
                monkeypatch.delenv(k, raising=False)
            for k, v in envs.items():
                monkeypatch.setenv(k, v)
            L = DeviceTDLambdaLearner(weights, max_games=n, alpha=0.1, lam=0.8)
            sq, cnt = L.replay_rows(rows, lengths, p1_won, batch_scale=scale, slots=slots)
            sq2, cnt2 = L.replay_rows(rows, lengths, p1_won, batch_scale=scale, slots=slots)     # a second round on used buffers
            out[tag] = (_np(L.theta).copy(), sq, cnt, sq2, cnt2)
        for k in ("BGAMD_TD_FUSE_STEP", "BGAMD_TD_PIPE", "BGAMD_TD_FUSE_MIN"):
            monkeypatch.delenv(k, raising=False)
        assert out["fused"][2] == out["two_launches"][2] == out["round2"][2] == int(_np(lengths).sum())
        if slots == 2048:          # (at the other sizes the routes group the games differently: other partial sums, other rounding)
            assert np.array_equal(out["fused"][0], out["two_launches"][0]) and out["fused"][1] == out["two_launches"][1] and out["fused"][3] == out["two_launches"][3]
            assert np.array_equal(out["fused"][0], out["round2"][0])
        else:
            assert np.abs(out["fused"][0] - out["round2"][0]).max() < 1e-5
        # the float64 closed form of the same streamed schedule (one round)
        Ld = DeviceTDLambdaLearner(weights, max_games=n, alpha=0.1, lam=0.8)
        Ld.replay_rows(rows, lengths, p1_won, batch_scale=scale, slots=slots)
        Lh = TDLambdaLearner(weights, device="cuda", alpha=0.1, lam=0.8, dtype=torch.float64)
        X = env.encode_rows(rows)
        Lh.replay_stream(X, lengths, p1_won, slots=slots, batch_scale=scale)
        th64 = _np(Lh.theta)
        moved = np.abs(th64 - weights).max()
        gap = np.abs(_np(Ld.theta) - th64).max()
        print("streamed replay through %d slots, fused step: max |theta - float64 closed form| = %.3g (weights moved %.3g)" % (slots, gap, moved))
        assert moved > 1e-3 and gap < 2e-4 * max(1.0, moved)


def test_streamed_round_65536_games_2048_slots_vs_float64(bg, weights):
    """The training configuration the quality study recommends, at config 4's per-GPU size: ONE round of 65 536 epsilon-greedy self-play
    games, its replay STREAMED through 2 048 slots (2 700 training steps of the two-launch step, 96 / 2 048 of every game's update) on the
    HIP learner -- against the float64 closed form of the same schedule over the whole round (PyTorch fp64 on the GPU), twice bit-identical,
    every (game, step) counted."""
    from backgammon_env.learner import DeviceTDLambdaLearner, TDLambdaLearner, play_round
    n, slots = 65536, 2048
    env = bg.VecGame(n, seed=60606)
    env.load_weights(weights)
    rows, lengths, p1_won = play_round(env, max_plies=320, epsilon=0.05)
    ln = _np(lengths)
    assert (ln > 0).mean() > 0.995 and env.stats()["error_flags"] == 0
    scale = 96.0 / slots
    out = []
    for rep in range(2):
        Ld = DeviceTDLambdaLearner(weights, max_games=n, alpha=0.1, lam=0.8)
        t0 = time.time()
        sq, cnt = Ld.replay_rows(rows, lengths, p1_won, batch_scale=scale, slots=slots)
        torch.cuda.synchronize()
        dt = time.time() - t0
        assert cnt == int(ln.sum())
        out.append(_np(Ld.theta).copy())
        del Ld
        torch.cuda.empty_cache()
    assert np.array_equal(out[0], out[1])
    Lh = TDLambdaLearner(weights, device="cuda", alpha=0.1, lam=0.8, dtype=torch.float64)
    X = env.encode_rows(rows)
    Lh.replay_stream(X, lengths, p1_won, slots=slots, batch_scale=scale)
    th64 = _np(Lh.theta)
    moved, gap = np.abs(th64 - weights).max(), np.abs(out[0] - th64).max()
    print("65 536 games (%d turns) streamed through %d slots: device replay %.0f ms; max |theta - float64 closed form| = %.3g at weights moved by %.3g"
          % (int(ln.sum()), slots, dt * 1e3, gap, moved))
    assert moved > 1e-2 and gap < 2e-4 * max(1.0, moved)


