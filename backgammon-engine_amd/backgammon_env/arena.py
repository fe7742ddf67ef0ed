"""Head-to-head evaluation: the reference's `_play_head_to_head` /
`evaluate_parallel` (pysrc/TD(λ) model/train.py:262-302) and `play_vs_random` / `play_vs_model`
(pysrc/benchmark.py:64-130) on the batched env: all games of a match-up advance together, PLAYER1's
lanes are stepped with one policy and PLAYER2's with the other."""
from __future__ import annotations

import torch

from . import F32, VecGame


def _play(env: VecGame, p1_policy, p2_policy, max_turns: int, precision):
    """policies: ("net", slot) or ("random",).  The constructor seats the first mover by game parity
    (`Game(i % 2)`, train.py:265), no opening roll."""
    env.reset()
    env.set_states(None, torch.arange(env.n, dtype=torch.int32) % 2)
    for t in range(max_turns):
        for player, pol in ((0, p1_policy), (1, p2_policy)):
            if pol[0] == "net":
                env.step_greedy(auto_reset=False, only_player=player, slot=pol[1], precision=precision)
            else:
                env.step_random(auto_reset=False, only_player=player)
        if t % 16 == 15 and bool(((env.flags() & 4) != 0).all()):
            break
    f = env.flags()
    done = (f & 4) != 0
    p1_won = done & (((f >> 1) & 1) == 0)
    return int(done.sum()), int(p1_won.sum())


def head_to_head(env: VecGame, weights_a, weights_b=None, max_turns: int = 2000, precision=F32):
    """Win rate of A vs B (B = None: a uniformly random mover), sides alternated 50/50 as in
    evaluate_parallel (train.py:296-302): every lane plays one game with A as PLAYER1 and one with A as
    PLAYER2.  -> dict(games, a_wins, win_rate)."""
    env.load_weights(weights_a, slot=0)
    if weights_b is not None:
        env.load_weights(weights_b, slot=1)
    b = ("net", 1) if weights_b is not None else ("random",)
    n1, w1 = _play(env, ("net", 0), b, max_turns, precision)          # A is PLAYER1
    n2, w2 = _play(env, b, ("net", 0), max_turns, precision)          # A is PLAYER2
    a_wins = w1 + (n2 - w2)
    return {"games": n1 + n2, "a_wins": a_wins, "win_rate": a_wins / max(n1 + n2, 1),
            "a_as_p1": (n1, w1), "a_as_p2": (n2, n2 - w2)}
