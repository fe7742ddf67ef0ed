#!/usr/bin/env python3
"""Round-3 fixture from the UNMODIFIED reference (build container only; same rules as make_golden.py, whose helpers it
reuses: the compiled reference engine in oracle/_ref, the reference Python imported in place, nothing copied).

  g8_bar_candidate_values.npz   fixture G7 for the positions G7 could not reach.  make_golden_r2.py sets positions with
        setGameBoard / setBorneOffPieces, and the binding has no setter for the bar counts (only hits fill the bar), so it
        skipped every turn with a checker on the bar: no row of G7 has a root with features 194 / 195 set, none is a
        bar-entry move.  Here fixture G5's 24 greedy games are REPLAYED from the start position through the reference's
        own tryMove (game.cpp:573-663) -- the bar fills the way it does in play -- and every turn whose root has a checker
        on the bar contributes its distinct afterstates with the values of the reference model's forward pass
        (fp32, and fp64 for the error budget).  The replay is checked against G5 turn by turn (pre-move state, post-move state).

    python tests/golden/make_golden_r3.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sets sys.path for the reference, imports it)

import numpy as np  # noqa: E402
import torch  # noqa: E402

REF = MG.REF


def g8(model):
    g5 = np.load(os.path.join(HERE, "g5_greedy_trajectories.npz"))["rows"]
    roots, cand, v32, v64, off = [], [], [], [], [0]
    entry_turns = 0
    for lane in np.unique(g5[:, 0]):
        rows = g5[g5[:, 0] == lane]
        g = MG.new_game()
        for r in rows:
            pre, turn, d1, d2, n_seq, chosen, post = r[2:30], int(r[30]), int(r[31]), int(r[32]), int(r[33]), int(r[34]), r[37:65]
            g.setTurn(turn)
            assert (MG.state28(g) == pre).all(), (lane, int(r[1]))
            g.setDice(d1, d2)
            seqs, states = g.evaluateTurnSequences(turn, d1, d2)
            assert len(seqs) == n_seq
            if seqs and (pre[24] or pre[25]):
                u = np.unique(np.asarray(states, dtype=np.int64), axis=0)
                X = torch.from_numpy(model._encode_states_np(u, turn))
                with torch.inference_mode():
                    a = model(X).squeeze(1).numpy().astype(np.float32)
                    b = model.double()(X.double()).squeeze(1).numpy()
                    model.float()
                roots.append(list(pre) + [turn, d1, d2])
                cand.append(u.astype(np.int8)); v32.append(a); v64.append(b); off.append(off[-1] + len(u))
                entry_turns += int(pre[24 + turn] > 0)
            if seqs:
                pl = g.getPlayers(turn)
                for o, d in seqs[chosen]:
                    ok, _ = g.tryMove(pl, abs(o - d), o, d)
                    assert ok
            assert (MG.state28(g) == post).all(), (lane, int(r[1]))
    return (np.array(roots, dtype=np.int32), np.array(off, dtype=np.int64), np.concatenate(cand), np.concatenate(v32),
            np.concatenate(v64), entry_turns)


def main():
    sd = torch.load(os.path.join(REF, "models", "tdgammonNEW100k.pth"), map_location="cpu", weights_only=True)
    model = MG.TDLGammonModel()
    model.load_state_dict(sd)
    model.eval()
    roots, off, cand, v32, v64, entry = g8(model)
    np.savez_compressed(os.path.join(HERE, "g8_bar_candidate_values.npz"), roots=roots, off=off, states=cand, v32=v32, v64=v64)
    print("g8:", len(roots), "turns with a checker on the bar (%d of them with the MOVER on the bar),"
          % entry, len(cand), "distinct afterstates, max |v32 - v64| =", float(np.abs(v32 - v64).max()))


if __name__ == "__main__":
    main()
