"""Round-3 GPU tests (all through the C ABI):
  * fixture G8: row-level value-net outputs for positions with checkers ON THE BAR (roots with features 194 / 195 set,
    bar-entry moves), written by the unmodified reference while replaying fixture G5's games -- what fixture G7 could
    not reach through the binding's setters; both delta kernels (the VALU one the step uses, the opt-in MFMA one);
  * the MFMA delta kernel (csrc/bg_eval_mfma.h, BGAMD_MFMA_DELTA=1): fixture G7, agreement with the VALU kernel, and the
    property its fixed-point W table exists for -- a row's value does not depend on the rows that share its piece;
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


@pytest.fixture(params=["valu_delta", "mfma_delta"])
def delta_kernel(request, monkeypatch):
    """The env reads BGAMD_MFMA_DELTA when it is created."""
    monkeypatch.setenv("BGAMD_MFMA_DELTA", "1" if request.param == "mfma_delta" else "0")
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


def test_mfma_delta_kernel_g7_and_agreement_with_the_valu_kernel(bg, golden_dir, weights, monkeypatch):
    """The opt-in MFMA delta kernel on fixture G7 (the reference model's own values, 2 390 rows), against the VALU kernel on
    the same rows (the fixed-point W table costs up to ~1.4e-6), and through the stand-alone operator bit for bit."""
    g = np.load(os.path.join(golden_dir, "g7_candidate_values.npz"))
    monkeypatch.setenv("BGAMD_MFMA_DELTA", "1")
    env, info, st, val, n_rows, e32, e64 = _fixture_step(bg, weights, g, "mfma_delta")
    print("eval_rows_mdelta_kernel, %d rows: max |gpu - reference fp32| = %.3g, fp64 %.3g" % (n_rows, e32, e64))
    assert e32 < 1e-5 and e64 < 1e-5
    monkeypatch.setenv("BGAMD_MFMA_DELTA", "0")
    env2, info2, st2, val2, _, e32v, _ = _fixture_step(bg, weights, g, "valu_delta")
    key = {(int(gi),) + tuple(int(x) for x in s): v for s, v, gi in zip(st2, val2, info2[:, 0])}
    gap = max(abs(float(v) - float(key[(int(gi),) + tuple(int(x) for x in s)])) for s, v, gi in zip(st, val, info[:, 0]))
    print("MFMA delta kernel vs VALU delta kernel on the same rows: max |dv| = %.3g (VALU kernel vs reference %.3g)" % (gap, e32v))
    assert gap < 5e-6
    roots, off = g["roots"], g["off"]
    ridx = np.repeat(np.arange(len(roots)), np.diff(off)).astype(np.int32)
    v2 = _np(env.evaluate_incremental(roots[:, :28], roots[:, 28], g["states"].astype(np.int32), ridx))
    by_key = {(int(gi),) + tuple(int(x) for x in s): v for s, v, gi in zip(st, val, info[:, 0])}
    assert all(by_key[(int(r),) + tuple(int(x) for x in s)] == v for s, v, r in zip(g["states"], v2, ridx))


def test_mfma_delta_values_do_not_depend_on_the_piece(bg, weights, monkeypatch):
    """Why bg_eval_mfma.h quantises its W table: the matrix pipe rounds its running sum, so with a floating hi + lo split a
    row's last bits depended on which other rows shared its K-compacted product (on where the leaf stage put the game in the
    arena).  With every term a multiple of the unit's quantum all sums are exact: the SAME rows evaluated in arena order,
    in a random order (every row mostly alone in its piece) and in reversed order give bit-identical values -- and so do
    duplicates of an afterstate, shards of an env and two runs of one seed."""
    monkeypatch.setenv("BGAMD_MFMA_DELTA", "1")
    n = 2048
    env = bg.VecGame(n, seed=515)
    env.load_weights(weights)
    env.run_greedy(30)
    s0, t0 = env.states().clone(), env.turns().clone()
    env.step_greedy(auto_reset=False, precision=bg.F32)
    info, st, val = env.unique_rows()
    assert env.stats()["error_flags"] == 0 and st.shape[0] > 10 * n
    gidx = info[:, 0].to(torch.int32)
    base = _np(env.evaluate_incremental(s0, t0, st, gidx))
    assert np.array_equal(base, _np(val))                     # the stand-alone operator == the step
    rng = np.random.RandomState(7)
    for perm in (rng.permutation(st.shape[0]), np.arange(st.shape[0])[::-1].copy()):
        p = torch.from_numpy(perm).to(st.device)
        out = _np(env.evaluate_incremental(s0, t0, st[p].contiguous(), gidx[p].contiguous()))
        assert np.array_equal(out, base[perm])
    # (the same experiment on the VALU kernel, whose per-row fp32 FMA chain never depended on its neighbours)
    monkeypatch.setenv("BGAMD_MFMA_DELTA", "0")
    env2 = bg.VecGame(n, seed=515)
    env2.load_weights(weights)
    b2 = _np(env2.evaluate_incremental(s0, t0, st, gidx))
    p = torch.from_numpy(rng.permutation(st.shape[0])).to(st.device)
    assert np.array_equal(_np(env2.evaluate_incremental(s0, t0, st[p].contiguous(), gidx[p].contiguous())), b2[_np(p)])
    print("MFMA vs VALU delta kernel on %d mid-game rows: max |dv| = %.3g" % (st.shape[0], np.abs(b2 - base).max()))
    assert np.abs(b2 - base).max() < 5e-6


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


def test_resident_and_lds_staged_root_pass_are_bit_identical(bg, weights, monkeypatch):
    """The root pass of the incremental value net exists twice: with the three bf16 planes of a wave's 32 hidden units resident in
    registers (the default since round 3) and staged through LDS in two K phases (BGAMD_ROOT_RESIDENT=0, rounds 1-2).  Same MFMA
    sequence per accumulator: the same games to the last bit, and the same per-row values through evaluate_incremental."""
    n = 8192
    monkeypatch.delenv("BGAMD_ROOT_RESIDENT", raising=False)
    a = bg.VecGame(n, seed=4711)
    monkeypatch.setenv("BGAMD_ROOT_RESIDENT", "0")
    b = bg.VecGame(n, seed=4711)
    monkeypatch.delenv("BGAMD_ROOT_RESIDENT", raising=False)
    a.load_weights(weights); b.load_weights(weights)
    for k in (1, 7, 40):
        a.run_greedy(k, epsilon=0.05); b.run_greedy(k, epsilon=0.05)
        assert torch.equal(a.states(), b.states()) and torch.equal(a.turns(), b.turns()), k
        la, lb = a.last_choice(), b.last_choice()
        assert torch.equal(la["value"], lb["value"]) and torch.equal(la["seq"], lb["seq"])
    assert a.stats() == b.stats() and a.stats()["error_flags"] == 0


def test_f16x2_with_resident_weights_equals_the_lds_staged_f16x2_kernel(bg, weights, monkeypatch):
    """BGAMD_F16X2_RESIDENT=1 (csrc/bg_eval_dense16.h: the dense f16 hi + lo value net with a wave's weight planes resident in registers,
    -log2 e and b1 folded into the table) against round 1's LDS-staged f16 x 2 kernel: values within 1e-6 of each other and of the fp32
    path's, and the same games for 40 steps (a near-tie may resolve differently: > 99.9 % of the lanes identical is asserted)."""
    n = 4096
    monkeypatch.delenv("BGAMD_F16X2_RESIDENT", raising=False)
    a = bg.VecGame(n, seed=1357)
    monkeypatch.setenv("BGAMD_F16X2_RESIDENT", "1")
    b = bg.VecGame(n, seed=1357)
    monkeypatch.delenv("BGAMD_F16X2_RESIDENT", raising=False)
    c = bg.VecGame(n, seed=1357)
    for e in (a, b, c):
        e.load_weights(weights)
    a.run_greedy(10, precision=bg.F16X2); b.run_greedy(10, precision=bg.F16X2); c.run_greedy(10)
    same = (a.states() == b.states()).all(1)
    assert same.float().mean().item() > 0.999
    # one more step from IDENTICAL positions: chosen values side by side
    idx = torch.nonzero(same & (a.states() == c.states()).all(1)).flatten()
    a.step_greedy(precision=bg.F16X2, auto_reset=False); b.step_greedy(precision=bg.F16X2, auto_reset=False); c.step_greedy(auto_reset=False)
    va, vb, vc = (e.last_choice()["value"][idx] for e in (a, b, c))
    moved = (a.last_choice()["count"][idx] > 0)
    d_ab, d_bc = (va - vb).abs()[moved].max().item(), (vb - vc).abs()[moved].max().item()
    print("f16x2 resident vs LDS-staged: max |dv| = %.3g; vs the fp32 incremental path: %.3g (%d lanes)" % (d_ab, d_bc, int(moved.sum())))
    assert d_ab < 1e-6 and d_bc < 2e-6
    for e in (a, b):
        e.run_greedy(30, precision=bg.F16X2)
    assert ((a.states() == b.states()).all(1)).float().mean().item() > 0.995
    assert a.stats()["error_flags"] == 0 and b.stats()["error_flags"] == 0
