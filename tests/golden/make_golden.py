#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the UNMODIFIED reference.

Runs only in the build container, where /root/reference exists:
  * the reference engine + bindings compiled as-is into oracle/_ref/ (oracle/Makefile `ref`);
  * the reference Python policy/learner imported in place from
    /root/reference/pysrc/TD(λ) model (never copied, never shipped).

The fixtures are DATA (inputs + the reference's outputs); the GPU box needs only them.
Dice and choice words come from the Philox streams defined in SURVEY.md §8d (implemented in
oracle/bg_oracle.c and verified there against the Random123 known answers) and are injected
into the reference through setDice(), because the reference RNG cannot be seeded.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import random
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path[:0] = [os.path.join(ROOT, "oracle", "_ref"), os.path.join(REF, "pysrc", "TD(λ) model"), ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import backgammon_env as bg  # noqa: E402  (the compiled reference)
from model import TDLGammonModel  # noqa: E402  (the reference policy)
import train as ref_train  # noqa: E402  (the reference learner)
from oracle import oracle as O  # noqa: E402  (Philox streams only)

SEED = 20240603
torch.set_num_threads(1)


_KEEP = []


def new_game(first=0):
    g = bg.Game(first)
    p1 = bg.Player("White", bg.PlayerType.PLAYER1)
    p2 = bg.Player("Black", bg.PlayerType.PLAYER2)
    g.setPlayers(p1, p2)
    _KEEP.append((p1, p2))   # Game stores raw Player* (bindings.cpp:64): keep them alive
    return g


def state28(g):
    return np.array(g.getGameBoard() + [g.getJailedCount(0), g.getJailedCount(1),
                                        g.getBornOffCount(0), g.getBornOffCount(1)], dtype=np.int8)


def call_digest(seqs, states):
    h = hashlib.sha256()
    for q in seqs:
        h.update(bytes([len(q)]))
        for o, d in q:
            h.update(bytes([o, d]))
    h.update(np.ascontiguousarray(states, dtype=np.int32).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


def pack_seqs(seqs):
    out = np.full((len(seqs), 4, 2), -1, dtype=np.int8)
    for i, q in enumerate(seqs):
        for j, (o, d) in enumerate(q):
            out[i, j] = (o, d)
    return out


# ---------------------------------------------------------------------------------------
# G2: start-position count table
def g2_counts():
    g = new_game()
    t = np.zeros((2, 6, 6), dtype=np.int32)
    for pl in (0, 1):
        for a in range(1, 7):
            for b in range(1, 7):
                t[pl, a - 1, b - 1] = len(g.legalTurnSequences(pl, a, b))
    return t


# G3 + G1: random-policy trajectories on the Philox streams, with per-call enumeration digests
def g3_trajectories(n_games=200, stride=200, full_calls_budget=60000):
    rows = []          # [lane, ply, state28(28), turn, d1, d2, C, chosen, over, winner]
    calls = []         # (lane, ply) -> digest for EVERY call
    full = {"idx": [], "off": [0], "seq": [], "states": []}
    rnd = random.Random(7)
    used = 0
    for lane in range(n_games):
        gid = lane                      # episode 0
        g = new_game()
        g.setTurn(O.lib().bgo_opening_turn(SEED, gid))
        ply = 0
        while True:
            pre = state28(g)
            turn = g.getTurn()
            d1, d2, cu, _ = O.turn_randoms(SEED, gid, ply)
            g.setDice(d1, d2)
            seqs, states = g.evaluateTurnSequences(turn, d1, d2)
            assert seqs == g.legalTurnSequences(turn, d1, d2)
            Cn = len(seqs)
            dig = call_digest(seqs, states)
            chosen = -1
            if Cn:
                chosen = (cu * Cn) >> 32
                pl = g.getPlayers(turn)
                for o, d in seqs[chosen]:
                    ok, _ = g.tryMove(pl, abs(o - d), o, d)
                    assert ok
                assert (state28(g) == states[chosen]).all()
            over, winner = g.is_game_over()
            rows.append([lane, ply] + pre.tolist() + [turn, d1, d2, Cn, chosen, int(over), winner])
            calls.append(dig)
            # keep the full ordered output of a sample of calls (all small ones early, a few big)
            if Cn and used + Cn <= full_calls_budget and (Cn <= 40 and rnd.random() < 0.06
                                                          or Cn > 40 and rnd.random() < 0.012):
                full["idx"].append(len(rows) - 1)
                full["seq"].append(pack_seqs(seqs))
                full["states"].append(states.astype(np.int8))
                full["off"].append(full["off"][-1] + Cn)
                used += Cn
            if over:
                break
            g.setTurn(1 - turn)
            ply += 1
    return (np.array(rows, dtype=np.int32), np.array(calls, dtype=np.uint64),
            np.array(full["idx"], dtype=np.int32), np.array(full["off"], dtype=np.int64),
            np.concatenate(full["seq"]), np.concatenate(full["states"]))


# G1 edge cases: hand-built boards through the public surface (setGameBoard / setBorneOffPieces /
# hits for the bar), every dice pair, both players
def g1_edges():
    boards = {
        "tests.cpp:405 failing_moves_prior": ([-8] + [0] * 21 + [1, 4], 0, 0),
        "tests.cpp:444 p2_single_checker": ([-1] + [0] * 23, 0, 14),
        "tests.cpp:528 p2_two_checkers": ([-1, -1] + [0] * 22, 0, 13),
        "tests.cpp:556 freeing": ([-5, -4, -4] + [0] * 19 + [4, 5], 0, 0),
        "pysrc/tests.py:17 basic_game": ([-5, -4] + [0] * 20 + [4, 5], 0, 0),
        "Q1 p2 overrun blocked by p1 on 4": ([0, -1, 0, 1] + [0] * 20, 0, 14),
        "Q1 p2 overrun blocked by p1 on 7": ([0, -1, 0, 0, 0, 0, 1] + [0] * 17, 0, 14),
        "Q1 p2 overrun free, p1 on 8": ([0, -1, 0, 0, 0, 0, 0, 1] + [0] * 16, 0, 14),
        "Q1 p1 overrun highest only": ([0] * 19 + [2, 0, 1, 0, 0], 12, 0),
        "p1 home with gap": ([-2] + [0] * 17 + [3, 0, 2, 0, 0, 1], 9, 0),
        "p2 home with gap": ([-1, 0, -2, 0, 0, -3] + [0] * 17 + [2], 0, 9),
        "blocked six-prime vs p1": ([2, 0, -2, -2, -2, -2, -2, -2] + [0] * 15 + [1], 0, 0),
        "mid-sequence win p1": ([0] * 22 + [1, 1], 13, 0),
        "mid-sequence win p2": ([-1, -1] + [0] * 22, 0, 13),
        "start": ([2, 0, 0, 0, 0, -5, 0, -3, 0, 0, 0, 5, -5, 0, 0, 0, 3, 0, 5, 0, 0, 0, 0, -2], 0, 0),
    }
    names, inp, counts, off, seqs_all, st_all = [], [], [], [0], [], []
    for name, (board, off1, off2) in boards.items():
        for pl in (0, 1):
            for d1 in range(1, 7):
                for d2 in range(1, 7):
                    g = new_game()
                    g.setGameBoard(board)
                    g.setBorneOffPieces(0, off1)
                    g.setBorneOffPieces(1, off2)
                    seqs, states = g.evaluateTurnSequences(pl, d1, d2)
                    names.append(name)
                    inp.append(board + [0, 0, off1, off2, pl, d1, d2])
                    counts.append(len(seqs))
                    seqs_all.append(pack_seqs(seqs))
                    st_all.append(states.astype(np.int8).reshape(-1, 28))
                    off.append(off[-1] + len(seqs))
    # bar cases reached by real hits: play random games and harvest states with bar > 0
    rnd = random.Random(11)
    harvested = 0
    for gi in range(400):
        if harvested >= 300:
            break
        g = new_game(gi)
        for _ in range(60):
            t = g.getTurn()
            d1, d2 = rnd.randint(1, 6), rnd.randint(1, 6)
            if g.getJailedCount(t) > 0 and harvested < 300 and rnd.random() < 0.5:
                seqs, states = g.evaluateTurnSequences(t, d1, d2)
                names.append("bar (harvested)")
                inp.append(state28(g).tolist() + [t, d1, d2])
                counts.append(len(seqs))
                seqs_all.append(pack_seqs(seqs))
                st_all.append(states.astype(np.int8).reshape(-1, 28))
                off.append(off[-1] + len(seqs))
                harvested += 1
            seqs = g.legalTurnSequences(t, d1, d2)
            if seqs:
                pl = g.getPlayers(t)
                for o, d in seqs[rnd.randrange(len(seqs))]:
                    g.tryMove(pl, abs(o - d), o, d)
            if g.is_game_over()[0]:
                break
            g.setTurn(1 - t)
    return (np.array(names), np.array(inp, dtype=np.int8), np.array(counts, dtype=np.int32),
            np.array(off, dtype=np.int64), np.concatenate(seqs_all), np.concatenate(st_all))


# G4/G5: encoder rows and value-net outputs from the reference model
def g45_encoder_values(traj_rows, n_rows=600):
    sd = torch.load(os.path.join(REF, "models", "tdgammonNEW100k.pth"), map_location="cpu",
                    weights_only=True)
    model = TDLGammonModel()
    model.load_state_dict(sd)
    model.eval()
    w = np.concatenate([sd["fc1.weight"].numpy().ravel(), sd["fc1.bias"].numpy().ravel(),
                        sd["fc2.weight"].numpy().ravel(), sd["fc2.bias"].numpy().ravel()]).astype(np.float32)
    rnd = np.random.RandomState(5)
    pick = rnd.choice(len(traj_rows), size=n_rows, replace=False)
    st = traj_rows[pick, 2:30].astype(np.int64)
    turn = traj_rows[pick, 30].astype(np.int32)
    X = np.zeros((n_rows, 198), dtype=np.float32)
    for t in (0, 1):
        m = turn == t
        X[m] = model._encode_states_np(st[m], t)
    with torch.inference_mode():
        v32 = model(torch.from_numpy(X)).squeeze(1).numpy().astype(np.float32)
        m64 = TDLGammonModel().double()
        m64.load_state_dict({k: v.double() for k, v in sd.items()})
        v64 = m64(torch.from_numpy(X).double()).squeeze(1).numpy()
    return w, st.astype(np.int8), turn, X, v32, v64, model


# G5b: greedy games through the reference make_move on injected Philox dice
def g5_greedy(model, n_games=24, lane0=1000):
    rows = []
    for lane in range(lane0, lane0 + n_games):
        gid = lane
        g = new_game()
        g.setTurn(O.lib().bgo_opening_turn(SEED, gid))
        ply = 0
        while True:
            pre = state28(g)
            turn = g.getTurn()
            d1, d2, _, _ = O.turn_randoms(SEED, gid, ply)
            g.setDice(d1, d2)
            seqs, states = g.evaluateTurnSequences(turn, d1, d2)
            best = model.make_move(g)
            post = state28(g)
            chosen = -1
            vals_gap = 0.0
            if seqs:
                chosen = seqs.index(best)       # first index of the returned sequence
                assert (states[chosen] == post).all()
                X = torch.from_numpy(model._encode_states_np(states, turn))
                with torch.inference_mode():
                    v = model(X).squeeze(1).numpy()
                srt = np.sort(np.unique(v))
                if len(srt) > 1:
                    vals_gap = float(srt[-1] - srt[-2]) if turn == 0 else float(srt[1] - srt[0])
            over, winner = g.is_game_over()
            rows.append([lane, ply] + pre.tolist() + [turn, d1, d2, len(seqs), chosen, int(over), winner]
                        + post.tolist() + [int(vals_gap * 1e9) if vals_gap < 2 else 2 ** 31 - 1])
            if over:
                break
            g.setTurn(1 - turn)
            ply += 1
    return np.array(rows, dtype=np.int32)


# G6: one game's TD(lambda) update through the reference learner
def g6_td(traj_rows):
    sd = torch.load(os.path.join(REF, "models", "tdgammonNEW100k.pth"), map_location="cpu",
                    weights_only=True)
    model = TDLGammonModel()
    model.load_state_dict(sd)
    lane = 3
    rows = traj_rows[traj_rows[:, 0] == lane]
    st = rows[:, 2:30].astype(np.int64)
    turn = rows[:, 30]
    states = [model._encode_states_np(st[i:i + 1], int(turn[i]))[0] for i in range(len(rows))]
    winner = int(rows[-1, 36])
    model.learning_rate = 0.1
    model.lambda_decay = 0.9
    model.initialize_traces()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    model.train()
    losses = ref_train.apply_td_updates(model, opt, states, winner == 0)
    sd2 = model.state_dict()
    w_after = np.concatenate([sd2["fc1.weight"].numpy().ravel(), sd2["fc1.bias"].numpy().ravel(),
                              sd2["fc2.weight"].numpy().ravel(), sd2["fc2.bias"].numpy().ravel()]).astype(np.float32)
    return (st.astype(np.int8), turn.astype(np.int8), np.array([winner], dtype=np.int32),
            np.array([0.1, 0.9], dtype=np.float64), w_after, np.array(losses, dtype=np.float64))


def main():
    out = lambda n: os.path.join(HERE, n)  # noqa: E731
    np.savez_compressed(out("g2_start_counts.npz"), counts=g2_counts())
    rows, digests, fidx, foff, fseq, fst = g3_trajectories()
    np.savez_compressed(out("g3_random_trajectories.npz"), seed=np.uint64(SEED), stride=np.int64(200),
                        rows=rows, digests=digests, full_idx=fidx, full_off=foff, full_seq=fseq,
                        full_states=fst)
    names, inp, counts, off, seqs, st = g1_edges()
    np.savez_compressed(out("g1_edge_calls.npz"), names=names, inputs=inp, counts=counts, off=off,
                        seq=seqs, states=st)
    w, st4, turn4, X, v32, v64, model = g45_encoder_values(rows)
    np.savez_compressed(out("g4_encoder_rows.npz"), states=st4, turn=turn4, X=X)
    np.savez_compressed(out("g5_values.npz"), states=st4, turn=turn4, v32=v32, v64=v64)
    w.tofile(out("tdgammonNEW100k.f32"))
    np.savez_compressed(out("g5_greedy_trajectories.npz"), seed=np.uint64(SEED), rows=g5_greedy(model))
    s6, t6, win6, hp6, w6, l6 = g6_td(rows)
    np.savez_compressed(out("g6_td_lambda.npz"), states=s6, turn=t6, winner=win6, alpha_lambda=hp6,
                        w_after=w6, losses=l6)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(out(f)))


if __name__ == "__main__":
    main()
