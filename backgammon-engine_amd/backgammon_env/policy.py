"""Inference-side mirror of the reference policy class (pysrc/TD(λ) model/model.py:31-222):
same method names and argument meaning, arithmetic in HIP kernels (encoder + MFMA value net).

    m = TDLGammonModel(); m.load_state_dict(torch.load("tdgammonNEW100k.pth"))
    seq = m.make_move(game)            # game: backgammon_env.Game
"""
from __future__ import annotations

import itertools

import numpy as np
import torch

from . import F32, Game, VecGame

_TOKENS = itertools.count(1)       # one per model object, never reused (id() is, once a model has been freed)


def flatten_state_dict(sd) -> np.ndarray:
    """4-tensor state_dict (train.py:513-515) -> flat float32[25601] in W1|b1|W2|b2 order."""
    return np.concatenate([np.asarray(sd[k].detach().cpu().float()).ravel()
                           for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")]).astype(np.float32)


class TDLGammonModel:
    def __init__(self, input_size=198, hidden_size=128):
        if (input_size, hidden_size) != (198, 128):
            raise ValueError("the MI355X kernels implement the 198->128->1 net")
        self.learning_rate = 0.1                       # model.py:39
        self.lambda_decay = 0.7                        # model.py:46
        self._w = None
        self._version = 0
        self._token = next(_TOKENS)
        self._ops = None                               # a small env used for stateless encode / evaluate

    # -- weights -------------------------------------------------------------------------------
    def load_state_dict(self, sd):
        self.load_flat(flatten_state_dict(sd))

    def load_flat(self, w):
        w = np.ascontiguousarray(np.asarray(w, dtype=np.float32).ravel())
        if w.size != 25601:
            raise ValueError("expected 25601 weights")
        self._w = w
        self._version += 1

    def state_dict(self):
        w = self._w
        return {"fc1.weight": torch.from_numpy(w[:25344].reshape(128, 198).copy()),
                "fc1.bias": torch.from_numpy(w[25344:25472].copy()),
                "fc2.weight": torch.from_numpy(w[25472:25600].reshape(1, 128).copy()),
                "fc2.bias": torch.from_numpy(w[25600:].copy())}

    def eval(self):
        return self

    def update_learning_params(self, episode):         # model.py:69-73
        self.learning_rate = max(0.01, 0.1 * (0.96 ** (episode // 40000)))
        self.lambda_decay = max(0.7, 0.9 * (0.96 ** (episode // 30000)))

    def _bind(self, env: VecGame):
        # envs outlive their Game (the one-lane pool): the key must name THIS model for good, not an address that the next
        # model may be given
        if getattr(env, "_w_version", None) != (self._token, self._version):
            env.load_weights(self._w)
            env._w_version = (self._token, self._version)

    def _op_env(self) -> VecGame:
        if self._ops is None:
            self._ops = VecGame(1, arena_rows=1 << 20)
        self._bind(self._ops)
        return self._ops

    # -- encoder (model.py:101-144) ---------------------------------------------------------------
    def _encode_states_np(self, states, turn):
        return self._op_env().encode(np.asarray(states, dtype=np.int32), int(turn)).cpu().numpy()

    def encode_state_np(self, game: Game):
        return self._encode_states_np([game._state()], game.getTurn())[0]

    def encode_state(self, game: Game):
        return torch.from_numpy(self.encode_state_np(game))

    # -- value net (model.py:63-67) on states: forward(encode(states, turn)) fused -----------------
    def values(self, states, turn, precision=F32):
        return self._op_env().evaluate(np.asarray(states, dtype=np.int32), turn, precision)

    # -- move selection (model.py:180-222) ------------------------------------------------------------
    def make_move(self, game: Game, game_idx: int = 1, epsilon: float = 0.0):
        env = game._v
        self._bind(env)
        env.step_greedy(roll=False, auto_reset=False, epsilon=epsilon, no_flip=True)
        game._dirty()
        c = env.last_choice()
        if int(c["count"][0]) == 0:
            return []
        n = int(c["seq_len"][0])
        return [(int(a), int(b)) for a, b in c["seq"][0, :n].cpu().tolist()]
