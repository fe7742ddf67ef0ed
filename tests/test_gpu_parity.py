"""GPU parity tests: the HIP path (through the C ABI) against the oracle on the same seeded inputs,
against the golden vectors captured from the unmodified reference, and -- at full size -- through
size-independent properties.  Integer work is bit-exact; value-net outputs within 1e-5 (north_star)."""
import os

import numpy as np
import pytest

from helpers import random_boards as _random_boards

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SEED = 20240603
START = [2, 0, 0, 0, 0, -5, 0, -3, 0, 0, 0, 5, -5, 0, 0, 0, 3, 0, 5, 0, 0, 0, 0, -2]


@pytest.fixture(scope="module")
def bg():
    import backgammon_env
    return backgammon_env


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _np(t):
    return t.cpu().numpy()


# ---- enumeration -----------------------------------------------------------------------------------

def test_g1_edge_calls_bit_exact(bg, golden_dir):
    """Hand-built edge boards (bear-off overrun asymmetry, bar entry, stuck positions, mid-sequence
    wins): ordered sequences + afterstates identical to the reference's evaluateTurnSequences."""
    g = np.load(os.path.join(golden_dir, "g1_edge_calls.npz"))
    inp, counts, off = g["inputs"].astype(np.int32), g["counts"], g["off"]
    n = len(inp)
    env = bg.VecGame(n, arena_rows=int(off[-1]) + 65536)
    env.set_states(inp[:, :28], inp[:, 28])
    offs, cnts, st, sq, ln = [_np(x) for x in env.enumerate(player=inp[:, 28], dice=inp[:, 29:31])]
    assert (cnts == counts).all()
    assert offs.sum() >= 0 and cnts.sum() == off[-1]
    for i in range(n):
        a, b = int(offs[i]), int(offs[i]) + int(cnts[i])
        assert (st[a:b] == g["states"][off[i]:off[i + 1]]).all(), g["names"][i]
        assert (sq[a:b] == g["seq"][off[i]:off[i + 1]]).all(), g["names"][i]
        assert (ln[a:b] == (g["seq"][off[i]:off[i + 1], :, 0] >= 0).sum(1)).all()


def test_enumerate_vs_oracle_4096(bg, O):
    """Config 2 size: 4 096 mid-game boards, every lane's full ordered enumeration vs the oracle."""
    n = 4096
    env = bg.VecGame(n, seed=SEED, arena_rows=4 << 20)
    for _ in range(37):
        env.step_random()
    st0, t0 = _np(env.states()), _np(env.turns())
    env.roll()
    dice = _np(env.dice())
    offs, cnts, st, sq, ln = [_np(x) for x in env.enumerate()]
    tot = 0
    for lane in range(n):
        s = O.State.from28(st0[lane], t0[lane])
        q, l, a = O.evaluate_turn_sequences(s, int(t0[lane]), int(dice[lane, 0]), int(dice[lane, 1]))
        assert cnts[lane] == len(l), lane
        o = int(offs[lane])
        assert (st[o:o + len(l)] == a).all(), lane
        assert (sq[o:o + len(l)] == q).all() and (ln[o:o + len(l)] == l).all(), lane
        tot += len(l)
    assert tot == cnts.sum()
    assert (_np(env.states()) == st0).all()          # enumeration does not mutate (tests.cpp:390-399)


def test_enumerate_arbitrary_positions_vs_oracle(bg, O):
    """8 192 arbitrary positions x random dice x random mover: full ordered enumeration, legalMoves for every die
    and a greedy-free random step (bounded kernels) against the oracle."""
    n = 8192
    st = _random_boards(n, 42)
    rng = np.random.RandomState(43)
    pl = rng.randint(0, 2, n).astype(np.int32)
    dice = rng.randint(1, 7, (n, 2)).astype(np.int32)
    dice[::5, 1] = dice[::5, 0]                             # plenty of doubles
    env = bg.VecGame(n, seed=44, arena_rows=16 << 20)
    env.set_states(st, pl)
    env.set_dice(dice)
    offs, cnts, est, esq, eln = [_np(x) for x in env.enumerate()]
    for lane in range(n):
        s = O.State.from28(st[lane], pl[lane])
        q, l, a = O.evaluate_turn_sequences(s, int(pl[lane]), int(dice[lane, 0]), int(dice[lane, 1]))
        assert cnts[lane] == len(l), (lane, st[lane].tolist(), pl[lane], dice[lane])
        o = int(offs[lane])
        assert (est[o:o + len(l)] == a).all() and (esq[o:o + len(l)] == q).all(), lane
    for die in (1, 3, 6):
        cnt, pairs = env.legal_moves(pl, np.full(n, die))
        cnt, pairs = _np(cnt), _np(pairs)
        for lane in range(0, n, 7):
            exp = O.legal_moves(O.State.from28(st[lane], pl[lane]), int(pl[lane]), die)
            assert [tuple(x) for x in pairs[lane, :cnt[lane]].tolist()] == exp, lane
    u = rng.randint(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    env.step_random(roll=False, auto_reset=False, choice_u32=u)
    post = _np(env.states())
    for lane in range(0, n, 3):
        s = O.State.from28(st[lane], pl[lane])
        O.step(s, int(dice[lane, 0]), int(dice[lane, 1]), 0, choice_u32=int(u[lane]))
        assert (post[lane] == s.to28()).all(), lane


def test_start_position_counts(bg, golden_dir):
    t = np.load(os.path.join(golden_dir, "g2_start_counts.npz"))["counts"]
    env = bg.VecGame(72)
    pl = np.repeat([0, 1], 36).astype(np.int32)
    dice = np.array([[a, b] for _ in (0, 1) for a in range(1, 7) for b in range(1, 7)], dtype=np.int32)
    env.set_states(np.tile(np.array(START + [0, 0, 0, 0], dtype=np.int32), (72, 1)), pl)
    _, cnts, _, _, _ = env.enumerate(player=pl, dice=dice)
    assert (_np(cnts).reshape(2, 6, 6) == t).all()


# ---- random-policy trajectories ---------------------------------------------------------------------

def test_g3_reference_trajectories(bg, golden_dir):
    """200 games played by the unmodified reference on injected Philox dice/choices: the env, started
    from its own reset (opening roll) and stepping with its own streams, reproduces every state."""
    g = np.load(os.path.join(golden_dir, "g3_random_trajectories.npz"))
    rows, stride = g["rows"], int(g["stride"])
    env = bg.VecGame(stride, seed=int(g["seed"]), lane_stride=stride)
    by_lane = [rows[rows[:, 0] == lane] for lane in range(stride)]
    T = max(len(r) for r in by_lane)
    assert (_np(env.turns()) == np.array([r[0, 30] for r in by_lane])).all()     # opening protocol
    for t in range(T):
        st, tn = _np(env.states()), _np(env.turns())
        for lane, r in enumerate(by_lane):
            if t < len(r):
                assert (st[lane] == r[t, 2:30]).all() and tn[lane] == r[t, 30], (lane, t)
        env.step_random(auto_reset=False)
        lc = env.last_choice()
        ch, cn, dice, fl = _np(lc["chosen"]), _np(lc["count"]), _np(env.dice()), _np(env.flags())
        for lane, r in enumerate(by_lane):
            if t < len(r):
                assert (dice[lane] == r[t, 31:33]).all() and cn[lane] == r[t, 33] and ch[lane] == r[t, 34], (lane, t)
                assert (fl[lane] & 1) == r[t, 35] and (not r[t, 35] or ((fl[lane] >> 1) & 1) == r[t, 36])
    s = env.stats()
    assert s["games_finished"] == stride and s["steps"] == len(rows)
    assert s["p1_wins"] == sum(int(r[-1, 36] == 0) for r in by_lane)
    assert s["candidates_raw"] == int(rows[:, 33].sum())


def test_random_trajectories_vs_oracle_4096(bg, O):
    """Config 2: B = 4 096, auto-reset, every lane's state checked after EVERY step for 200 steps."""
    n, steps = 4096, 200
    env = bg.VecGame(n, seed=SEED)
    snaps = np.zeros((steps, n, 29), dtype=np.int32)
    flags = np.zeros((steps, n), dtype=np.int32)
    for t in range(steps):
        env.step_random()
        snaps[t, :, :28] = _np(env.states())
        snaps[t, :, 28] = _np(env.turns())
        flags[t] = (_np(env.flags()) >> 4) & 3
    fin = ctot = 0
    for lane in range(n):
        ref, f, c, _ = O.lane_run(SEED, lane, n, steps, 0)
        assert (ref[:, :29] == snaps[:, lane]).all(), lane
        assert (ref[:, 29] == flags[:, lane]).all(), lane
        fin += f
        ctot += c
    s = env.stats()
    assert s["games_finished"] == fin and s["candidates_raw"] == ctot and s["steps"] == n * steps


def test_random_step_bounded_equals_walk(bg):
    """The bounded task kernels and the whole-tree walk are two implementations of the same reference-order
    choice: identical states, choices and sequences for 16 384 lanes over 120 steps."""
    n = 16384
    a, b = bg.VecGame(n, seed=123), bg.VecGame(n, seed=123)
    for t in range(120):
        a.step_random(); b.step_random(walk=True)
        if t % 10 == 9:
            la, lb = a.last_choice(), b.last_choice()
            for k in ("chosen", "count", "seq", "seq_len"):
                assert (_np(la[k]) == _np(lb[k])).all(), (t, k)
    assert (_np(a.states()) == _np(b.states())).all() and (_np(a.turns()) == _np(b.turns())).all()
    assert a.stats() == b.stats()


def test_shard_invariance(bg):
    """lane g of shard r plays global game r*B/R + g -- results do not depend on R."""
    n, steps = 1024, 60
    whole = bg.VecGame(n, seed=7)
    parts = [bg.VecGame(n // 4, seed=7, lane_offset=r * (n // 4), lane_stride=n) for r in range(4)]
    for _ in range(steps):
        whole.step_random()
        for p in parts:
            p.step_random()
    got = np.concatenate([_np(p.states()) for p in parts])
    assert (got == _np(whole.states())).all()
    assert sum(p.stats()["games_finished"] for p in parts) == whole.stats()["games_finished"]


def test_full_size_properties_65536(bg):
    """BASELINE size: 65 536 lanes. Checker conservation, determinism, turn sanity after 150 steps."""
    n = 65536
    a, b = bg.VecGame(n, seed=11), bg.VecGame(n, seed=11)
    for _ in range(150):
        a.step_random()
        b.step_random()
    sa = _np(a.states())
    assert (sa == _np(b.states())).all() and (_np(a.turns()) == _np(b.turns())).all()
    p1 = np.clip(sa[:, :24], 0, None).sum(1) + sa[:, 24] + sa[:, 26]
    p2 = np.clip(-sa[:, :24], 0, None).sum(1) + sa[:, 25] + sa[:, 27]
    assert (p1 == 15).all() and (p2 == 15).all()
    assert (sa[:, 26] < 15).all() and (sa[:, 27] < 15).all()      # finished games were reset
    st = a.stats()
    assert st["steps"] == n * 150 and st["games_finished"] > 0 and st["error_flags"] == 0
    assert 0.3 < st["p1_wins"] / st["games_finished"] < 0.7       # CLAUDE.md:70 random-vs-random sanity


# ---- single-checker surface ------------------------------------------------------------------------

def test_try_move_and_legal_moves_vs_oracle(bg, O):
    n = 2048
    env = bg.VecGame(n, seed=3)
    for _ in range(45):
        env.step_random()
    rng = np.random.RandomState(0)
    st, tn = _np(env.states()), _np(env.turns())
    for die in range(1, 7):
        for pl in (0, 1):
            cnt, pairs = env.legal_moves(np.full(n, pl), np.full(n, die))
            cnt, pairs = _np(cnt), _np(pairs)
            for lane in range(0, n, 9):
                exp = O.legal_moves(O.State.from28(st[lane], tn[lane]), pl, die)
                assert [tuple(x) for x in pairs[lane, :cnt[lane]].tolist()] == exp
    # probes: half legal moves, half arbitrary
    pl = rng.randint(0, 2, n); dice = rng.randint(1, 7, n)
    org = rng.randint(-1, 27, n); dst = rng.randint(-1, 27, n)
    for lane in range(0, n, 2):
        mv = O.legal_moves(O.State.from28(st[lane], tn[lane]), int(tn[lane]), int(dice[lane]))
        if mv:
            pl[lane] = tn[lane]
            org[lane], dst[lane] = mv[rng.randint(len(mv))]
    err = _np(env.try_move(pl, dice, org, dst))
    after = _np(env.states())
    n_ok = 0
    for lane in range(n):
        s = O.State.from28(st[lane], tn[lane])
        ok, msg = O.try_move(s, int(pl[lane]), int(dice[lane]), int(org[lane]), int(dst[lane]))
        assert bg.ERR_MESSAGES[int(err[lane])] == msg, (lane, err[lane], msg)
        assert (after[lane] == s.to28()).all(), lane
        n_ok += ok
    assert n_ok > n // 4


# ---- encoder + value net ------------------------------------------------------------------------------

def test_encoder_bit_exact(bg, golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_encoder_rows.npz"))
    env = bg.VecGame(1)
    X = _np(env.encode(g["states"].astype(np.int32), g["turn"]))
    assert X.dtype == np.float32 and np.array_equal(X, g["X"])


def test_values_vs_reference_model(bg, O, golden_dir, weights):
    """fp32 MFMA value net vs the reference PyTorch model (fixture) and the fp64 oracle: <= 1e-5."""
    g = np.load(os.path.join(golden_dir, "g5_values.npz"))
    env = bg.VecGame(1, arena_rows=1 << 20)
    env.load_weights(weights)
    v = _np(env.evaluate(g["states"].astype(np.int32), g["turn"]))
    print("max |gpu - torch fp32| = %.3g, max |gpu - torch fp64| = %.3g" % (np.abs(v - g["v32"]).max(), np.abs(v - g["v64"]).max()))
    assert np.abs(v - g["v32"]).max() < 1e-5
    assert np.abs(v - g["v64"]).max() < 1e-5
    # ragged sizes around the 32-row tile, and a large batch
    rng = np.random.RandomState(1)
    for m in (1, 31, 32, 33, 97, 50000):
        idx = rng.randint(0, len(g["turn"]), m)
        vv = _np(env.evaluate(g["states"][idx].astype(np.int32), g["turn"][idx]))
        assert np.abs(vv - g["v64"][idx]).max() < 1e-5, m


def _bf16_round(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def test_bf16_mode_matches_bf16_emulation(bg, golden_dir, weights):
    """bf16 speed mode (v_mfma_f32_32x32x16_bf16): against an fp64 evaluation of the SAME bf16-rounded
    weights and features the kernel agrees to 2e-5; against the fp32 reference it is ~3e-4 (not a parity mode)."""
    g = np.load(os.path.join(golden_dir, "g5_values.npz"))
    g4 = np.load(os.path.join(golden_dir, "g4_encoder_rows.npz"))
    env = bg.VecGame(1, arena_rows=1 << 20)
    env.load_weights(weights)
    v = _np(env.evaluate(g["states"].astype(np.int32), g["turn"], precision=bg.BF16))
    W1 = _bf16_round(weights[:25344].reshape(128, 198)).astype(np.float64)
    b1, W2, b2 = weights[25344:25472].astype(np.float64), weights[25472:25600].astype(np.float64), float(weights[25600])
    X = _bf16_round(g4["X"]).astype(np.float64)
    h = 1.0 / (1.0 + np.exp(-(X @ W1.T + b1)))
    ref = 1.0 / (1.0 + np.exp(-(h @ W2 + b2)))
    print("bf16: max |gpu - bf16 emulation| = %.3g, max |gpu - fp32 reference| = %.3g" % (np.abs(v - ref).max(), np.abs(v - g["v32"]).max()))
    assert np.abs(v - ref).max() < 2e-5
    assert np.abs(v - g["v32"]).max() < 5e-3


def test_f16x2_mode_is_inside_the_parity_bound(bg, O, golden_dir, weights):
    """f16 hi+lo split of W1 on v_mfma_f32_32x32x16_f16 (exact products, fp32 accumulation): values within 1e-5 of
    the reference PyTorch model like the f32 MFMA kernel, and the same greedy choices."""
    g = np.load(os.path.join(golden_dir, "g5_values.npz"))
    env = bg.VecGame(1, arena_rows=1 << 20)
    env.load_weights(weights)
    v = _np(env.evaluate(g["states"].astype(np.int32), g["turn"], precision=bg.F16X2))
    print("f16x2: max |gpu - torch fp32| = %.3g, max |gpu - torch fp64| = %.3g" % (np.abs(v - g["v32"]).max(), np.abs(v - g["v64"]).max()))
    assert np.abs(v - g["v32"]).max() < 1e-5 and np.abs(v - g["v64"]).max() < 1e-5
    rng = np.random.RandomState(2)
    for m in (1, 31, 32, 33, 97, 50000):
        idx = rng.randint(0, len(g["turn"]), m)
        vv = _np(env.evaluate(g["states"][idx].astype(np.int32), g["turn"][idx], precision=bg.F16X2))
        assert np.abs(vv - g["v64"][idx]).max() < 1e-5, m
    n = 4096
    a, b = bg.VecGame(n, seed=19), bg.VecGame(n, seed=19)
    a.load_weights(weights); b.load_weights(weights)
    for t in range(40):
        pre, pt = _np(b.states()), _np(b.turns())
        a.step_greedy(auto_reset=False); b.step_greedy(auto_reset=False, precision=bg.F16X2)
        if t % 8 == 0:
            _check_greedy_step(O, weights, pre, pt, _np(b.dice()), _np(b.states()), range(t % 5, n, 41))
    agree = (_np(a.states()) == _np(b.states())).all(1).mean()
    print("f16x2 vs f32: identical boards after 40 greedy steps on %.2f %% of lanes" % (100 * agree))
    assert agree > 0.97


def test_bf16_mode_choice_agreement(bg, O, weights):
    """Same boards, same dice: the bf16 step picks the same afterstate as the fp32 step on most lanes and a
    near-optimal one (fp64 value within 5e-3 of the best) on all."""
    n = 4096
    a, b = bg.VecGame(n, seed=17), bg.VecGame(n, seed=17)
    a.load_weights(weights); b.load_weights(weights)
    for _ in range(30):
        a.step_greedy(); b.step_greedy()
    pre, pt = _np(a.states()), _np(a.turns())
    assert (pre == _np(b.states())).all()
    a.step_greedy(auto_reset=False); b.step_greedy(auto_reset=False, precision=bg.BF16)
    pa, pb, dice = _np(a.states()), _np(b.states()), _np(a.dice())
    agree = (pa == pb).all(1).mean()
    print("bf16 vs fp32 identical choice on %.2f %% of lanes" % (100 * agree))
    assert agree > 0.93
    for lane in np.where(~(pa == pb).all(1))[0][:200]:
        s = O.State.from28(pre[lane], pt[lane])
        _, _, cand = O.evaluate_turn_sequences(s, int(pt[lane]), int(dice[lane, 0]), int(dice[lane, 1]))
        vv = O.forward_f64(weights, O.encode(cand, int(pt[lane])))
        k = [i for i in range(len(cand)) if (cand[i] == pb[lane]).all()]
        assert k
        best = vv.max() if pt[lane] == 0 else vv.min()
        assert abs(vv[k[0]] - best) < 5e-3


def _check_greedy_step(O, w, pre, pt, dice, post, lanes):
    for lane in lanes:
        s = O.State.from28(pre[lane], pt[lane])
        _, _, cand = O.evaluate_turn_sequences(s, int(pt[lane]), int(dice[lane, 0]), int(dice[lane, 1]))
        if len(cand) == 0:
            assert (post[lane] == pre[lane]).all()
            continue
        v = O.forward_f64(w, O.encode(cand, int(pt[lane])))
        k = [i for i in range(len(cand)) if (cand[i] == post[lane]).all()]
        assert k, lane
        best = v.max() if pt[lane] == 0 else v.min()
        assert abs(v[k[0]] - best) < 1e-5, (lane, v[k[0]], best)


def test_greedy_steps_value_optimal(bg, O, weights):
    """Config 3 semantics at 4 096 lanes: after each greedy step every applied afterstate is a legal
    candidate whose oracle value is within 1e-5 of the arg-max (P1) / arg-min (P2)."""
    n = 4096
    env = bg.VecGame(n, seed=SEED)
    env.load_weights(weights)
    for t in range(40):
        pre, pt = _np(env.states()), _np(env.turns())
        was_live = (_np(env.flags()) & 4) == 0
        env.step_greedy(auto_reset=False)
        post, dice = _np(env.states()), _np(env.dice())
        if t % 4 == 0:
            _check_greedy_step(O, weights, pre, pt, dice, post, [l for l in range(t % 7, n, 29) if was_live[l]])
        live = (_np(env.flags()) & 4) == 0
        assert (_np(env.turns()) != pt)[live].all()
        assert (post == pre)[~was_live].all()
    assert env.stats()["error_flags"] == 0


def test_g5_reference_greedy_games(bg, golden_dir, weights):
    """Games played by the reference make_move: wherever its best/second-best gap exceeds 2e-6 the env
    picks the identical afterstate (index, sequence and board)."""
    g = np.load(os.path.join(golden_dir, "g5_greedy_trajectories.npz"))
    rows = g["rows"]
    n = len(rows)
    env = bg.VecGame(n, arena_rows=max(int(rows[:, 33].sum()) + 65536, 1 << 20))
    env.load_weights(weights)
    env.set_states(rows[:, 2:30], rows[:, 30])
    env.set_dice(rows[:, 31:33])
    env.step_greedy(roll=False, auto_reset=False, want_index=True)
    post = _np(env.states())
    lc = env.last_choice()
    ch, cn = _np(lc["chosen"]), _np(lc["count"])
    assert (cn == rows[:, 33]).all()
    clear = (rows[:, 65] > 2000) | (rows[:, 33] <= 1)
    assert clear.mean() > 0.9
    assert (post[clear] == rows[clear, 37:65]).all()
    assert (ch[clear] == rows[clear, 34]).all()
    fl = _np(env.flags())
    assert ((fl & 1) == rows[:, 35])[clear].all()


def test_greedy_index_matches_ordered_enumeration(bg, weights):
    """The staged greedy step (keys, de-duplicated rows) against the env's own ordered enumeration:
    with BGAMD_WANT_INDEX the reported index points at the applied afterstate in the reference-order
    list, the reported sequence is that entry's sequence, and it is the FIRST entry with that board."""
    n = 4096
    env = bg.VecGame(n, seed=99, arena_rows=4 << 20)
    env.load_weights(weights)
    for _ in range(25):
        env.step_greedy()
    env.roll()
    offs, cnts, st, sq, ln = [_np(x) for x in env.enumerate()]
    env.step_greedy(roll=False, auto_reset=False, want_index=True)
    post = _np(env.states())
    lc = env.last_choice()
    ch, cn, cs, cl = _np(lc["chosen"]), _np(lc["count"]), _np(lc["seq"]), _np(lc["seq_len"])
    assert (cn == cnts).all()
    for lane in range(n):
        if cnts[lane] == 0:
            assert ch[lane] == -1
            continue
        r = int(offs[lane]) + int(ch[lane])
        assert (st[r] == post[lane]).all(), lane
        assert (sq[r] == cs[lane]).all() and ln[r] == cl[lane], lane
        seg = st[int(offs[lane]):r]
        assert not (seg == post[lane]).all(1).any(), lane       # first occurrence
    s = env.stats()
    assert s["rows_evaluated"] < 0.5 * int(cnts.sum()) * 26      # duplicates were dropped before the value net


def test_epsilon_greedy_explores(bg, weights):
    n = 4096
    a, b = bg.VecGame(n, seed=5), bg.VecGame(n, seed=5)
    a.load_weights(weights); b.load_weights(weights)
    a.step_greedy(epsilon=0.0); b.step_greedy(epsilon=0.5)
    ca, cb = _np(a.last_choice()["chosen"]), _np(b.last_choice()["chosen"])
    frac = (ca != cb).mean()
    assert 0.15 < frac < 0.55          # about half explore, some explorations hit the greedy index


def test_epsilon_exploration_matches_stream_definition(bg, O, weights):
    """model.py:205-206 with the Philox TURN stream: a lane explores iff (x3 >> 8) * 2^-24 < eps and then plays
    reference-order candidate k = (x2 * C) >> 32 -- checked lane by lane against the oracle's enumeration."""
    n, eps = 2048, 0.35
    env = bg.VecGame(n, seed=321)
    env.load_weights(weights)
    for _ in range(12):
        env.step_greedy()
    pre, pt = _np(env.states()), _np(env.turns())
    ply, epi = [_np(x) for x in env.progress()]
    env.step_greedy(epsilon=eps, auto_reset=False)
    post, ch = _np(env.states()), _np(env.last_choice()["chosen"])
    n_explore = 0
    for lane in range(n):
        gid = lane + int(epi[lane]) * n
        d1, d2, cu, eu = O.turn_randoms(321, gid, int(ply[lane]))
        if not (np.float32(eu >> 8) * np.float32(1.0 / 16777216.0) < np.float32(eps)):
            continue
        _, _, cand = O.evaluate_turn_sequences(O.State.from28(pre[lane], pt[lane]), int(pt[lane]), d1, d2)
        if len(cand) == 0:
            assert (post[lane] == pre[lane]).all()
            continue
        k = (cu * len(cand)) >> 32
        assert ch[lane] == k and (post[lane] == cand[k]).all(), lane
        n_explore += 1
    assert 0.25 * n < n_explore < 0.45 * n


# ---- scalar Game surface (the reference's own tests, run through the drop-in module) ---------------------

def test_scalar_game_known_answers(bg):
    p1, p2 = bg.Player("A", bg.PlayerType.PLAYER1), bg.Player("B", bg.PlayerType.PLAYER2)
    g = bg.Game(0)
    g.setPlayers(p1, p2)
    assert g.getTurn() == bg.PlayerType.PLAYER1 and bg.Game(3).getTurn() == 1
    assert g.getGameBoard() == START
    assert g.legalMoves(bg.PlayerType.PLAYER1, 1) == [(1, 2), (17, 18), (19, 20)]          # tests.cpp:287
    assert g.legalMoves(bg.PlayerType.PLAYER2, 1) == [(6, 5), (8, 7), (24, 23)]            # :300
    assert g.legalMoves(0, 5) == [(12, 17), (17, 22)]                                      # :313
    seqs = g.legalTurnSequences(0, 1, 2)
    assert [(1, 2), (2, 4)] in seqs and [(1, 3), (3, 4)] in seqs and len(seqs) == 30       # :346
    d = g.legalTurnSequences(0, 1, 1)
    assert len(d) == 245 and all(len(q) == 4 for q in d) and [(19, 20)] * 4 in d           # :367
    assert g.getGameBoard() == START                                                       # :390
    assert g.tryMove(p1, 5, 1, 6) == (False, "Invalid destination.")                       # :116-121
    seqs, states = g.evaluateTurnSequences(0, 3, 1)
    assert states.shape == (len(seqs), 28) and states.dtype == np.int32
    g.setGameBoard([-1] + [0] * 23)                                                        # :433-488
    assert g.legalTurnSequences(1, 3, 2) == [[(1, 0)], [(1, 0)]]
    assert g.legalTurnSequences(1, 1, 1) == [[(1, 0)]]
    g.setBorneOffPieces(1, 14)
    assert g.is_game_over() == (False, -1)
    assert g.tryMove(p2, 1, 1, 0) == (True, "")
    assert g.is_game_over() == (True, 1) and g.getBornOffCount(1) == 15 and g.getPieces().numFreed(1) == 15
    with pytest.raises(TypeError):
        bg.Player("x", 0)
    c = g.clone()
    assert c.getGameBoard() == g.getGameBoard() and c.get_last_dice() == [1, 1]
    r = g.roll_dice()
    assert len(r) == 2 and all(1 <= v <= 6 for v in r) and g.get_last_dice() == r
    assert any(g.roll_dice() != r for _ in range(8))                                       # fresh dice per call


def test_scalar_make_move_like_reference_tests(bg, weights):
    """pysrc/tests.py:38-49 through the mirror classes, plus agreement with VecGame on the same board."""
    from backgammon_env.policy import TDLGammonModel
    m = TDLGammonModel()
    m.load_flat(weights)
    m.eval()
    g = bg.Game(0)
    p1, p2 = bg.Player("White", bg.PlayerType.PLAYER1), bg.Player("Black", bg.PlayerType.PLAYER2)
    g.setPlayers(p1, p2)
    g.setTurn(bg.PlayerType.PLAYER1)
    g.setDice(3, 1)
    seq = m.make_move(g)
    assert seq == [(17, 20), (19, 20)]            # the reference's own choice for (3,1) from the start
    assert g.getTurn() == 0                       # make_move does not flip the turn
    b = g.getGameBoard()
    assert b[16] == 2 and b[18] == 4 and b[19] == 2
    x = m.encode_state_np(g)
    assert x.shape == (198,) and x[192] == 1.0 and x[193] == 0.0


# ---- trajectory log + TD(lambda) learner ---------------------------------------------

def test_trajectory_log_and_learner_round(bg, O, weights):
    """play_round logs the PRE-move state of every turn (train.py:105-106) as 32-byte rows; the rows decode to
    exactly the states the env held, their device encoding equals the oracle's encoder bit for bit, and the
    lock-step TD(λ) replay on the GPU equals the same replay on the CPU."""
    from backgammon_env.learner import TDLambdaLearner, play_round
    n = 256
    env = bg.VecGame(n, seed=77)
    env.load_weights(weights)
    # 1. log vs the states observed from outside
    traj = env.record_trajectory(64)
    env.reset()
    seen_s, seen_t = [], []
    for t in range(20):
        seen_s.append(_np(env.states())); seen_t.append(_np(env.turns()))
        env.step_greedy(auto_reset=False)
    X = _np(env.encode_rows(traj[:20]))
    for t in range(20):
        for tb in (0, 1):
            m = seen_t[t] == tb
            assert np.array_equal(X[t][m], O.encode(seen_s[t][m], tb)), t
    env.record_trajectory(None)
    # 2. a full round, then the replay on GPU vs CPU
    rows, lengths, p1_won = play_round(env, max_plies=400)
    ln = _np(lengths)
    assert (ln > 20).all() and ln.max() <= rows.shape[0]
    st = env.stats()
    Xr = env.encode_rows(rows)
    Lg = TDLambdaLearner(weights, device="cuda", alpha=0.1, lam=0.9)
    sq_g, cnt_g = Lg.replay(Xr, lengths, p1_won)
    Lc = TDLambdaLearner(weights, device="cpu", alpha=0.1, lam=0.9)
    sq_c, cnt_c = Lc.replay(Xr.cpu(), lengths.cpu(), p1_won.cpu())
    assert cnt_g == cnt_c == int(ln.sum())
    d = np.abs(Lg.theta.cpu().numpy() - Lc.theta.numpy()).max()
    moved = np.abs(Lc.theta.numpy() - weights).max()
    assert d < 1e-4 * max(1.0, moved) and moved > 1e-3
    env.load_weights(Lg.theta.cpu().numpy())              # the next round plays with the updated net
    env.reset(); env.step_greedy()
    assert env.stats()["error_flags"] == 0 and st["games_finished"] >= n


def test_head_to_head_two_weight_slots(bg, O, weights):
    """train.py:262-277: PLAYER1's lanes are moved by one net, PLAYER2's by another (two weight slots,
    BGAMD_ONLY_P1/P2).  Each side's move is value-optimal under ITS OWN weights, lanes of the other side do not
    move, and the trained net beats a uniformly random mover."""
    from backgammon_env.arena import head_to_head
    n = 1024
    rng = np.random.RandomState(3)
    wb = (weights + rng.normal(0, 0.05, weights.shape)).astype(np.float32)
    env = bg.VecGame(n, seed=41)
    env.load_weights(weights, slot=0); env.load_weights(wb, slot=1)
    env.reset()
    for t in range(12):
        for player, slot, w in ((0, 0, weights), (1, 1, wb)):
            pre, pt = _np(env.states()), _np(env.turns())
            env.step_greedy(auto_reset=False, only_player=player, slot=slot)
            post, dice = _np(env.states()), _np(env.dice())
            idle = pt != player
            assert (post[idle] == pre[idle]).all() and (_np(env.turns())[idle] == pt[idle]).all()
            if t % 3 == 0:
                _check_greedy_step(O, w, pre, pt, dice, post, [l for l in range(t, n, 37) if not idle[l]])
    r = head_to_head(bg.VecGame(512, seed=9), weights, None)
    print("trained net vs random mover:", r)
    assert r["games"] == 1024 and r["win_rate"] > 0.9


def test_greedy_auto_reset_and_terminal_65536(bg, O, weights):
    """BASELINE size under the greedy policy: games end, winners are the side that just bore off its 15th
    checker, lanes restart from the start position with the opening-roll turn of the NEXT global game id,
    checkers are conserved, and two identically seeded envs stay identical (determinism despite atomics)."""
    n, steps = 65536, 120
    a, b = bg.VecGame(n, seed=2024), bg.VecGame(n, seed=2024)
    a.load_weights(weights); b.load_weights(weights)
    start = np.array(START + [0, 0, 0, 0], dtype=np.int32)
    n_reset = 0
    for t in range(steps):
        mover = _np(a.turns())
        a.step_greedy(); b.step_greedy()
        if t >= 40 and t % 8 == 0:
            fl = (_np(a.flags()) >> 4) & 3
            done = (fl & 1) == 1
            if done.any():
                st, tn = _np(a.states()), _np(a.turns())
                _, epi = [_np(x) for x in a.progress()]
                assert (st[done] == start).all()
                assert (((fl >> 1) & 1)[done] == mover[done]).all()          # the mover won
                for lane in np.where(done)[0][:64]:
                    assert tn[lane] == O.lib().bgo_opening_turn(2024, int(lane) + int(epi[lane]) * n)
                n_reset += int(done.sum())
    sa = _np(a.states())
    assert (sa == _np(b.states())).all() and (_np(a.turns()) == _np(b.turns())).all()
    p1 = np.clip(sa[:, :24], 0, None).sum(1) + sa[:, 24] + sa[:, 26]
    p2 = np.clip(-sa[:, :24], 0, None).sum(1) + sa[:, 25] + sa[:, 27]
    assert (p1 == 15).all() and (p2 == 15).all()
    s = a.stats()
    assert s["steps"] == n * steps and s["games_finished"] > n // 2 and n_reset > 0 and s["error_flags"] == 0
    sb = b.stats()
    # results are deterministic; the WORK counters are not (how many duplicates meet in one block depends on the
    # order in which blocks win the bump allocations)
    assert all(s[k] == sb[k] for k in ("steps", "games_finished", "p1_wins", "error_flags"))


def test_errors_are_loud(bg, weights):
    """Arena overflow, out-of-range counts and missing weights are reported, never silently absorbed."""
    env = bg.VecGame(64, arena_rows=65536)
    with pytest.raises(bg.BgamdError):
        env.step_greedy()                                     # no weights loaded
    bad = np.tile(np.array(START + [0, 0, 0, 0], dtype=np.int32), (64, 1))
    bad[3, 0] = 16
    env.set_states(bad, np.zeros(64, dtype=np.int32))
    with pytest.raises(bg.BgamdError):
        env.stats()                                           # BGAMD_E_STATE
    small = bg.VecGame(70000, arena_rows=65536)               # 70 000 lanes cannot enumerate into 65 536 rows
    small.roll()
    with pytest.raises(bg.BgamdError):
        small.enumerate()                                     # BGAMD_E_ARENA
    tiny = bg.VecGame(70000, arena_rows=65536)                # the greedy step overflows its afterstate arena too:
    tiny.load_weights(weights)                                #   flagged, the value net stays inside the arena
    tiny.step_greedy()
    with pytest.raises(bg.BgamdError):
        tiny.stats()
    g = bg.Game(0)
    with pytest.raises(ValueError):
        g.setGameBoard([0] * 25)


def test_bench_contract_line():
    """bench.py prints ONE JSON line with the contract fields (metric/value/unit/..., roofline, cpu_baseline)."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "8", "--warmup", "2", "--burnin", "20",
                          "--games", "8192"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == 8 and d["warmup"] == 2 and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["value"] > 1e6 and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # a FRACTION of the roof that bounds the executed work (the incremental value net runs on the VALUs, no MFMA)
    assert r["bound"] in ("hbm", "mfma", "valu") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    assert 0.0 < r["frac"] <= 1.0, r["frac"]
    for k in d["kernels"]:
        if isinstance(d["kernels"][k], dict):
            assert 0.0 <= d["kernels"][k]["frac"] <= 1.0, k
    assert d["timed_regions"] >= 1 and d["region_ms"]["min"] <= d["region_ms"]["median"] <= d["region_ms"]["max"]
    assert abs(d["ms_per_step"] * d["steps"] - d["region_ms"]["median"]) < 1e-2
    assert len(d["source_hash"]) == 16
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0 and c["sample"]
    assert c["host"]["cpu_model"] and c["host"]["nproc"] >= 1
    ac = c["all_cores"]
    assert ac["cores"] == c["host"]["usable_cores"] and ac["value"] > 0 and ac["kind"] == "port"


def test_reset_lanes_by_mask(bg, O):
    """reset(mask): masked lanes restart as their next episode (start board, opening-roll turn of the next global
    game id), the others are untouched."""
    n = 512
    env = bg.VecGame(n, seed=55)
    for _ in range(20):
        env.step_random(auto_reset=False)
    st, tn = _np(env.states()), _np(env.turns())
    mask = (np.arange(n) % 3 == 0).astype(np.int32)
    env.reset(mask)
    st2, tn2 = _np(env.states()), _np(env.turns())
    ply, epi = [_np(x) for x in env.progress()]
    keep = mask == 0
    assert (st2[keep] == st[keep]).all() and (tn2[keep] == tn[keep]).all() and (epi[keep] == 0).all()
    assert (st2[~keep] == np.array(START + [0, 0, 0, 0])).all() and (epi[~keep] == 1).all() and (ply[~keep] == 0).all()
    for lane in np.where(~keep)[0][:40]:
        assert tn2[lane] == O.lib().bgo_opening_turn(55, int(lane) + n)


def test_device_learner_single_game_matches_reference_fixture(bg, O, weights, golden_dir):
    """HIP TD(λ) learner (bgamd_td_*): ONE game replayed step by step equals the reference's apply_td_updates
    (fixture G6, written by the reference learner), same bar as the host-side closed form: max |Δθ| < 2e-6."""
    from backgammon_env.learner import DeviceTDLambdaLearner
    g = np.load(os.path.join(golden_dir, "g6_td_lambda.npz"))
    st, turn = g["states"].astype(np.int32), g["turn"].astype(np.int32)
    T = len(st)
    rows = bg.pack_rows(st, turn).reshape(T, 1, 8)
    # the packed rows decode to the oracle's encoding
    env = bg.VecGame(1)
    X = _np(env.encode_rows(rows))[:, 0]
    for i in range(T):
        assert np.array_equal(X[i], O.encode(st[i:i + 1], int(turn[i]))[0])
    alpha, lam = g["alpha_lambda"]
    L = DeviceTDLambdaLearner(weights, max_games=4, alpha=alpha, lam=lam)
    sq, cnt = L.replay_rows(rows, [T], [int(g["winner"][0]) == 0])
    assert cnt == T
    w_after = _np(L.theta)
    assert np.abs(w_after - g["w_after"]).max() < 2e-6
    assert np.abs(g["w_after"] - weights).max() > 1e-4
    # the reference reports Σ δ² without the terminal step; ours adds the terminal one (>= 0, <= 1)
    assert -1e-6 <= sq - float(np.sum(g["losses"])) < 1.0
    sd = L.state_dict()
    assert tuple(sd["fc1.weight"].shape) == (128, 198) and tuple(sd["fc2.weight"].shape) == (1, 128)


def test_device_learner_round_matches_host_closed_form(bg, O, weights):
    """A ragged round (games of different lengths, some lanes not replayed) through the HIP learner equals the
    host-side closed-form replay in float64 on the CPU; the update is deterministic run to run."""
    from backgammon_env.learner import DeviceTDLambdaLearner, TDLambdaLearner, play_round
    n = 200                                               # not a multiple of the kernels' group sizes
    env = bg.VecGame(n, seed=123)
    env.load_weights(weights)
    rows, lengths, p1_won = play_round(env, max_plies=400, epsilon=0.1)
    lengths = lengths.clone()
    lengths[::7] = 0                                       # lanes that are not replayed
    lengths[3] = 1                                         # a one-turn game: only the terminal step
    ln = _np(lengths)
    Xr = env.encode_rows(rows).cpu().double()
    Lc = TDLambdaLearner(weights, device="cpu", alpha=0.1, lam=0.9, dtype=torch.float64)
    sq_c, cnt_c = Lc.replay(Xr, lengths.cpu(), p1_won.cpu(), batch_scale=0.25)
    # ... and the oracle's independent numpy restatement of the reference learner says the same
    th_o, sq_o, cnt_o = O.td_lambda_lockstep(weights, Xr.numpy(), ln, _np(p1_won), 0.1, 0.9, batch_scale=0.25)
    assert cnt_o == cnt_c and np.abs(th_o - Lc.theta.numpy()).max() < 1e-10
    out = []
    for rep in range(2):
        Ld = DeviceTDLambdaLearner(weights, max_games=n, alpha=0.1, lam=0.9)
        sq_d, cnt_d = Ld.replay_rows(rows, lengths, p1_won, batch_scale=0.25)
        out.append(_np(Ld.theta))
        assert cnt_d == cnt_c == int(ln.sum())
        assert abs(sq_d - sq_c) < 1e-3 * max(1.0, sq_c)
    assert np.array_equal(out[0], out[1])
    moved = np.abs(Lc.theta.numpy() - weights).max()
    d = np.abs(out[0] - Lc.theta.numpy()).max()
    assert moved > 1e-3 and d < 2e-5 * max(1.0, moved), (d, moved)
    # a second round on the same learner object (buffers are reused, traces restart at zero)
    Ld.set_weights(weights)
    Ld.replay_rows(rows, lengths, p1_won, batch_scale=0.25)
    assert np.array_equal(_np(Ld.theta), out[0])
    # the distributed route (bgamd_td_step hands the update out, the caller all-reduces, bgamd_td_apply) on one rank
    Ld.set_weights(weights)
    sq_s, cnt_s = Ld.replay_rows(rows, lengths, p1_won, batch_scale=0.25, split_apply=True)
    assert cnt_s == cnt_c and np.array_equal(_np(Ld.theta), out[0])


def test_incremental_value_net_equals_dense_chain(bg, O, weights):
    """BGAMD_F32 evaluates the hidden layer incrementally (root term per game + the W1 columns of the features an
    afterstate changes); BGAMD_F32_DENSE runs the dense fp32 MFMA chain over every afterstate.  Same positions, same
    dice: the chosen values agree to fp32 rounding (the two differ only in the association of one sum), the chosen
    moves are the same except at ties inside that rounding, and both are value-optimal for the oracle (1e-5)."""
    n = 4096
    for burn, seed in ((0, 11), (9, 12), (40, 13)):
        a, b = bg.VecGame(n, seed=seed), bg.VecGame(n, seed=seed)
        os.environ["BGAMD_ROOT_F32"] = "1"             # third env (experimental build only): root term by the f32 MFMA chain
        try:
            c = bg.VecGame(n, seed=seed)
        finally:
            del os.environ["BGAMD_ROOT_F32"]
        a.load_weights(weights); b.load_weights(weights); c.load_weights(weights)
        for _ in range(burn):
            a.step_random(); b.step_random(); c.step_random()
        c.step_greedy(auto_reset=False, precision=bg.F32)
        pre, pt = _np(a.states()), _np(a.turns())
        assert np.array_equal(pre, _np(b.states()))
        a.step_greedy(auto_reset=False, precision=bg.F32)
        b.step_greedy(auto_reset=False, precision=bg.F32_DENSE)
        ca, cb = a.last_choice(), b.last_choice()
        va, vb = _np(ca["value"]), _np(cb["value"])
        moved = _np(ca["count"]) > 0
        assert np.array_equal(moved, _np(cb["count"]) > 0)
        assert np.abs(va[moved] - vb[moved]).max() < 2e-6
        vc = _np(c.last_choice()["value"])
        assert np.abs(va[moved] - vc[moved]).max() < 1e-6      # default root pass (f16 x 2 since round 4) == f32 MFMA root term (fp32 rounding)
        print("burn %d: max |incremental - dense| = %.3g, max |default root pass - f32 MFMA root pass| = %.3g, same move on %.4f of lanes"
              % (burn, np.abs(va[moved] - vb[moved]).max(), np.abs(va[moved] - vc[moved]).max(),
                 (_np(a.states()) == _np(b.states())).all(axis=1).mean()))
        assert ((_np(c.states()) == _np(a.states())).all(axis=1)).mean() > 0.995
        sa, sb = _np(a.states()), _np(b.states())
        same = (sa == sb).all(axis=1)
        assert same.mean() > 0.995, same.mean()
        dice = _np(a.dice())
        lanes = [int(l) for l in np.nonzero(~same)[0][:40]] + list(range(0, n, 97))
        _check_greedy_step(O, weights, pre, pt, dice, sa, lanes)
        _check_greedy_step(O, weights, pre, pt, dice, sb, lanes)
        assert a.stats()["error_flags"] == 0


def test_step_is_graph_capturable(bg, weights):
    """The greedy step is pure stream-ordered work (no host synchronisation, no host-side state): it can be captured in
    a HIP graph through torch and replayed; the replayed games equal the eagerly stepped ones bit for bit."""
    n, per_graph, replays = 2048, 4, 6
    a, b = bg.VecGame(n, seed=99), bg.VecGame(n, seed=99)
    a.load_weights(weights); b.load_weights(weights)
    b.step_greedy(epsilon=0.1)                       # warm-up outside capture (lazy module load, event pools)
    a.step_greedy(epsilon=0.1)
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(per_graph):
                b.step_greedy(epsilon=0.1)
    # capture records, it does not execute: a and b are still in step
    for _ in range(replays):
        g.replay()
    for _ in range(per_graph * replays):
        a.step_greedy(epsilon=0.1)
    torch.cuda.synchronize()
    assert np.array_equal(_np(a.states()), _np(b.states())) and np.array_equal(_np(a.turns()), _np(b.turns()))
    sa, sb = a.stats(), b.stats()
    assert sa["error_flags"] == 0 and sb["error_flags"] == 0
    assert all(sa[k] == sb[k] for k in ("steps", "games_finished", "p1_wins"))


def _replay_key(O, pre28, turn, d1, d2, key):
    """Afterstate of the move sequence a staged row's key describes (pass | o0..o3 | len), through the oracle."""
    ln, pas = key & 7, (key >> 23) & 1
    dA, dB = (d2, d1) if pas else (d1, d2)
    s = O.State.from28(pre28, turn)
    for k in range(ln):
        o, die = (key >> (18 - 5 * k)) & 31, (dB if k & 1 else dA)
        dest = [b for a, b in O.legal_moves(s, turn, die) if a == o]
        assert dest, ("key names an illegal move", key, k, o, die)
        ok, msg = O.try_move(s, turn, die, o, dest[0])
        assert ok, msg
    return tuple(int(v) for v in s.to28())


def test_staged_rows_cover_every_distinct_afterstate(bg, O, weights):
    """What the value net is handed in a greedy step: every key replays (through the oracle's tryMove) to a legal
    afterstate, and the set of those afterstates is EXACTLY the set of distinct afterstates of the reference-order
    enumeration -- the pruning of commuting move orders drops copies, never a position."""
    n = 1024
    env = bg.VecGame(n, seed=2024)
    env.load_weights(weights)
    rows_total = distinct_total = 0
    for ply in range(36):
        pre, pt = _np(env.states()), _np(env.turns())
        live = (_np(env.flags()) & 4) == 0
        env.step_greedy(auto_reset=False)
        if ply % 5:
            continue
        dice = _np(env.dice())
        info = _np(env.unique_rows_info())
        by_game = {}
        for g, k in info:
            by_game.setdefault(int(g), []).append(int(k))
        for lane in range(0, n, 3):
            if not live[lane]:
                assert lane not in by_game
                continue
            turn, d1, d2 = int(pt[lane]), int(dice[lane, 0]), int(dice[lane, 1])
            _, _, cand = O.evaluate_turn_sequences(O.State.from28(pre[lane], turn), turn, d1, d2)
            want = {tuple(int(v) for v in c) for c in cand}
            keys = by_game.get(lane, [])
            assert all((k >> 31) == turn for k in keys)
            got = {_replay_key(O, pre[lane], turn, d1, d2, k & 0x7FFFFFFF) for k in keys}
            assert got == want, (ply, lane, len(got), len(want))
            rows_total += len(keys); distinct_total += len(want)
    assert distinct_total > 20000 and rows_total < 1.25 * distinct_total


def test_staged_rows_on_edge_boards_and_late_game(bg, O, weights, golden_dir):
    """The same coverage check where the commuting-moves rules are most delicate: the reference's hand-built edge
    boards (bear-off overrun asymmetry, bar entry, stuck positions, mid-sequence wins; fixture G1: the afterstates are the
    reference's own) and the bear-off phase of self-play games."""
    g = np.load(os.path.join(golden_dir, "g1_edge_calls.npz"))
    inp, off = g["inputs"].astype(np.int32), g["off"]
    n = len(inp)
    env = bg.VecGame(n, arena_rows=1 << 20)
    env.load_weights(weights)
    env.set_states(inp[:, :28], inp[:, 28])
    env.set_dice(inp[:, 29:31])
    env.step_greedy(roll=False, auto_reset=False)
    info = _np(env.unique_rows_info())
    by_game = {}
    for gm, k in info:
        by_game.setdefault(int(gm), []).append(int(k))
    for i in range(n):
        turn, d1, d2 = int(inp[i, 28]), int(inp[i, 29]), int(inp[i, 30])
        want = {tuple(int(v) for v in c) for c in g["states"][off[i]:off[i + 1]]}
        got = {_replay_key(O, inp[i, :28], turn, d1, d2, k & 0x7FFFFFFF) for k in by_game.get(i, [])}
        assert got == want, (str(g["names"][i]), len(got), len(want))
    # late game: bear-off
    n2 = 512
    env = bg.VecGame(n2, seed=4242)
    env.load_weights(weights)
    checked = 0
    for ply in range(120):
        pre, pt = _np(env.states()), _np(env.turns())
        live = (_np(env.flags()) & 4) == 0
        env.step_greedy(auto_reset=False)
        if ply < 60 or ply % 6:
            continue
        dice = _np(env.dice())
        info = _np(env.unique_rows_info())
        by_game = {}
        for gm, k in info:
            by_game.setdefault(int(gm), []).append(int(k))
        for lane in range(0, n2, 2):
            if not live[lane]:
                continue
            turn, d1, d2 = int(pt[lane]), int(dice[lane, 0]), int(dice[lane, 1])
            _, _, cand = O.evaluate_turn_sequences(O.State.from28(pre[lane], turn), turn, d1, d2)
            want = {tuple(int(v) for v in c) for c in cand}
            got = {_replay_key(O, pre[lane], turn, d1, d2, k & 0x7FFFFFFF) for k in by_game.get(lane, [])}
            assert got == want, (ply, lane, len(got), len(want))
            checked += 1
    assert checked > 300
    # arbitrary (unreachable) positions: heavy stacks, both sides on the bar, late bear-off boards, every roll class
    n3 = 1500
    st3 = _random_boards(n3, 77)
    rng = np.random.RandomState(78)
    turn3 = rng.randint(0, 2, n3).astype(np.int32)
    dice3 = rng.randint(1, 7, (n3, 2)).astype(np.int32)
    env = bg.VecGame(n3, arena_rows=1 << 21)
    env.load_weights(weights)
    env.set_states(st3, turn3)
    env.set_dice(dice3)
    env.step_greedy(roll=False, auto_reset=False)
    assert env.stats()["error_flags"] == 0
    info = _np(env.unique_rows_info())
    by_game = {}
    for gm, k in info:
        by_game.setdefault(int(gm), []).append(int(k))
    for i in range(n3):
        t, d1, d2 = int(turn3[i]), int(dice3[i, 0]), int(dice3[i, 1])
        if O.over(O.State.from28(st3[i], t))[0]:
            continue                                       # a finished board is not stepped
        _, _, cand = O.evaluate_turn_sequences(O.State.from28(st3[i], t), t, d1, d2)
        want = {tuple(int(v) for v in c) for c in cand}
        got = {_replay_key(O, st3[i], t, d1, d2, k & 0x7FFFFFFF) for k in by_game.get(i, [])}
        assert got == want, (i, len(got), len(want))


@pytest.mark.parametrize("n", [1, 63, 65, 777])
def test_odd_lane_counts(bg, O, weights, n):
    """Lane counts that are not multiples of any wave / workgroup / tile size: every kernel's tail handling and the
    size-dependent launch shapes (nodes per stage workgroup, value-net grids) on small envs."""
    env = bg.VecGame(n, seed=1000 + n)
    env.load_weights(weights)
    for t in range(14):
        pre, pt = _np(env.states()), _np(env.turns())
        live = (_np(env.flags()) & 4) == 0
        env.step_greedy(auto_reset=False, epsilon=0.1 if t % 3 == 2 else 0.0)
        post, dice = _np(env.states()), _np(env.dice())
        if t % 3 != 2:
            _check_greedy_step(O, weights, pre, pt, dice, post, [l for l in range(n) if live[l]][:120])
        p1 = np.clip(post[:, :24], 0, None).sum(1) + post[:, 24] + post[:, 26]
        p2 = np.clip(-post[:, :24], 0, None).sum(1) + post[:, 25] + post[:, 27]
        assert (p1 == 15).all() and (p2 == 15).all()
    for _ in range(6):
        env.step_random()
    assert env.stats()["error_flags"] == 0


def test_run_greedy_equals_repeated_steps(bg, weights):
    """bgamd_env_run_greedy(k) plays exactly the games of k calls of step_greedy (fused step boundaries, alternating
    counter sets, trajectory log, exploration), for the incremental and a dense value-net mode."""
    n = 3000
    for prec, eps in ((bg.F32, 0.0), (bg.F32, 0.15), (bg.F16X2, 0.0)):
        a, b = bg.VecGame(n, seed=321), bg.VecGame(n, seed=321)
        a.load_weights(weights); b.load_weights(weights)
        ta, tb = a.record_trajectory(40), b.record_trajectory(40)
        for k in (1, 2, 7, 16):
            for _ in range(k):
                a.step_greedy(precision=prec, epsilon=eps)
            b.run_greedy(k, precision=prec, epsilon=eps)
            assert np.array_equal(_np(a.states()), _np(b.states())) and np.array_equal(_np(a.turns()), _np(b.turns())), (prec, eps, k)
        assert np.array_equal(_np(ta), _np(tb))
        sa, sb = a.stats(), b.stats()
        assert sa["error_flags"] == 0 and all(sa[q] == sb[q] for q in ("steps", "games_finished", "p1_wins", "rows_evaluated"))
        ia, ib = _np(a.unique_rows_info()), _np(b.unique_rows_info())       # block order follows the allocation atomics
        assert np.array_equal(ia[np.lexsort((ia[:, 1], ia[:, 0]))], ib[np.lexsort((ib[:, 1], ib[:, 0]))])
