"""`backgammon_env` -- host-side mirror of the reference's pybind11 module of the same name
(cppsrc/backgammon_bindings.cpp:41-94), running on MI355X through libbgamd.so.

Same names, argument meaning and error behaviour as the reference module:
    PlayerType, Player, Pieces, Game           (scalar surface: one board = a one-lane env)
plus the vectorised surface the batched self-play loop uses:
    VecGame                                    (n boards per device, one board per lane)
and, in `backgammon_env.policy`, the TDLGammonModel operators (encode / forward / make_move).

Everything that computes runs in HIP kernels; there is no CPU fallback.  PyTorch is used only
to own device buffers and streams.
"""
from __future__ import annotations

import ctypes as C
import enum
import os

# The greedy step runs its root pass on a second stream beside the doubles plies; ROCm maps streams onto GPU_MAX_HW_QUEUES (default 4)
# hardware queues, and a process that also holds an RCCL communicator has more streams than that: the env's two then share a queue and
# every step loses the overlap (0.153 -> 0.167 ms at 65 536 lanes).  The variable is read when the HIP runtime starts: this default
# takes effect when the package is imported before the first GPU call (multi-rank launchers: export it).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

from . import _capi
from ._capi import (AUTO_RESET, BF16, F16X2, F32, F32_DENSE, NO_FLIP, ONLY_P1, ONLY_P2, ROLL, WANT_INDEX, WEIGHTS_SLOT1,  # noqa: F401
                    BgamdError)

__all__ = ["PlayerType", "Player", "Pieces", "Game", "VecGame", "BgamdError", "set_seed", "pack_rows"]

ERR_MESSAGES = {                                   # cppsrc/game.cpp:585-642, in source order
    0: "", 1: "Invalid origin", 2: "Origin out of range", 3: "Destination out of range",
    4: "Cannot move in that direction.", 5: "Move does not match dice.", 6: "Invalid destination.",
    7: "Cannot bear off from jail",
}

_seed = int.from_bytes(os.urandom(8), "little")    # reference: random_device (game.hpp:45)
_next_scalar_id = 0


def set_seed(seed: int):
    """Seeds the dice of scalar Game objects created afterwards (the reference cannot be seeded)."""
    global _seed, _next_scalar_id
    _seed = int(seed) & (2 ** 64 - 1)
    _next_scalar_id = 0                  # (pooled envs are re-seeded when they are handed out: Game.__init__)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def pack_rows(states28, turn, device="cuda"):
    """[..., 28] reference-layout states + turn (scalar or [...]) -> int32 [..., 8] 32-byte rows (the trajectory
    log's format: 8 bit planes, turn of the side to move in plane 0 bit 31)."""
    st = torch.as_tensor(states28, dtype=torch.int32).to(device).contiguous()
    lead = st.shape[:-1]
    n = st.numel() // 28
    t = torch.as_tensor(turn, dtype=torch.int32).to(device).expand(lead).contiguous()
    out = torch.empty((n, 8), dtype=torch.int32, device=st.device)
    _capi.check(_capi.load().bgamd_pack_rows(_ptr(st), _ptr(t), n, _ptr(out), _stream()), "pack_rows")
    return out.reshape(*lead, 8)


class PlayerType(enum.IntEnum):                    # bindings.cpp:46-48 (unscoped enum: equals ints)
    PLAYER1 = 0
    PLAYER2 = 1


PLAYER1, PLAYER2 = PlayerType.PLAYER1, PlayerType.PLAYER2


class Player:                                      # bindings.cpp:51-54, cppsrc/player.hpp:6-27
    def __init__(self, name: str, num: PlayerType):
        if not isinstance(name, str) or not isinstance(num, PlayerType):
            raise TypeError("Player(name: str, num: PlayerType)")
        self._name, self._num = name, int(num)

    def getName(self):
        return self._name

    def getNum(self):
        return self._num


class Pieces:                                      # bindings.cpp:57-59: live view of a Game's counters
    def __init__(self, game: "Game"):
        self._game = game

    def numJailed(self, player):
        return self._game.getJailedCount(player)

    def numFreed(self, player):
        return self._game.getBornOffCount(player)


class VecGame:
    """n concurrent games on one MI355X, one board per lane (include/bgamd.h)."""

    def __init__(self, n_games: int, device: int = 0, seed: int = 20240603, lane_offset: int = 0,
                 lane_stride: int | None = None, arena_rows: int = 0):
        self._lib = _capi.load()
        if not torch.cuda.is_available():
            raise BgamdError("no GPU visible: backgammon_env has no CPU fallback")
        self.n = int(n_games)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        _capi.check(self._lib.bgamd_env_create(C.byref(h), self.n, device, seed, lane_offset,
                                               lane_stride or 0, arena_rows), "bgamd_env_create")
        self._h = h
        self._has_weights = False

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bgamd_env_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers
    def _buf(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _dev(self, x, dtype, shape):
        t = torch.as_tensor(x, dtype=dtype).to(self.device).contiguous()
        if tuple(t.shape) != tuple(shape):
            t = t.expand(shape).contiguous()
        return t

    # -- state
    def reset(self, mask=None, episode=None):
        """mask=None: every lane back to episode 0 (counters cleared).  mask [n]: only the lanes with mask != 0 restart
        (next episode).  episode=k: every lane starts its episode k -- global game id lane_offset + lane + k * lane_stride,
        i.e. fresh dice and a fresh opening roll for every round of a one-game-per-lane training loop."""
        if episode is not None:
            _capi.check(self._lib.bgamd_env_reset_episode(self._h, int(episode), _stream()), "reset_episode")
            return
        if mask is None:
            _capi.check(self._lib.bgamd_env_reset(self._h, _stream()), "reset")
            return
        m = self._dev(mask, torch.int32, (self.n,))
        _capi.check(self._lib.bgamd_env_reset_lanes(self._h, _ptr(m), _stream()), "reset_lanes")
        torch.cuda.current_stream().synchronize()

    def reseed(self, seed: int, lane_offset: int = 0, lane_stride: int = 0, null_stream: bool = False):
        """The dice streams of a freshly created env with these arguments, on this one (episode 0, start position)."""
        st = None if null_stream else _stream()
        _capi.check(self._lib.bgamd_env_reseed(self._h, int(seed) & (2 ** 64 - 1), int(lane_offset), int(lane_stride), st), "reseed")

    def set_states(self, states28=None, turn=None):
        s = self._dev(states28, torch.int32, (self.n, 28)) if states28 is not None else None
        t = self._dev(turn, torch.int32, (self.n,)) if turn is not None else None
        _capi.check(self._lib.bgamd_env_set_states(self._h, _ptr(s), _ptr(t), _stream()), "set_states")
        torch.cuda.current_stream().synchronize()     # s / t may be temporaries

    def states(self):
        s = self._buf((self.n, 28), torch.int32)
        _capi.check(self._lib.bgamd_env_get_states(self._h, _ptr(s), None, _stream()), "get_states")
        return s

    def turns(self):
        t = self._buf((self.n,), torch.int32)
        _capi.check(self._lib.bgamd_env_get_states(self._h, None, _ptr(t), _stream()), "get_states")
        return t

    def flags(self):
        f = self._buf((self.n,), torch.int32)
        _capi.check(self._lib.bgamd_env_get_flags(self._h, _ptr(f), _stream()), "get_flags")
        return f

    def snapshot(self):
        """int32 [n, 32] = state28 | turn | die1 | die2 | flags: every scalar getter's data in one launch."""
        out = self._buf((self.n, 32), torch.int32)
        _capi.check(self._lib.bgamd_env_snapshot(self._h, _ptr(out), _stream()), "snapshot")
        return out

    def set_dice(self, dice):
        d = self._dev(dice, torch.int32, (self.n, 2))
        _capi.check(self._lib.bgamd_env_set_dice(self._h, _ptr(d), _stream()), "set_dice")
        torch.cuda.current_stream().synchronize()

    def dice(self):
        d = self._buf((self.n, 2), torch.int32)
        _capi.check(self._lib.bgamd_env_get_dice(self._h, _ptr(d), _stream()), "get_dice")
        return d

    def roll(self, advance_ply: bool = False):
        _capi.check(self._lib.bgamd_env_roll(self._h, int(advance_ply), _stream()), "roll")

    # -- enumeration (evaluateTurnSequences for every lane)
    def enumerate(self, player=None, dice=None):
        """-> (offsets int64[n], counts int32[n], states int32[N,28], seq int8[N,4,2], seq_len int32[N]),
        rows of lane g = [offsets[g], offsets[g]+counts[g]), reference order."""
        p = self._dev(player, torch.int32, (self.n,)) if player is not None else None
        d = self._dev(dice, torch.int32, (self.n, 2)) if dice is not None else None
        _capi.check(self._lib.bgamd_env_enumerate(self._h, _ptr(p), _ptr(d), _stream()), "enumerate")
        offs, cnts = self._buf((self.n,), torch.int64), self._buf((self.n,), torch.int32)
        total = _capi.check(self._lib.bgamd_env_candidates_info(self._h, _ptr(offs), _ptr(cnts), _stream()),
                            "candidates_info")
        st = self._buf((total, 28), torch.int32)
        sq = self._buf((total, 4, 2), torch.int8)
        ln = self._buf((total,), torch.int32)
        _capi.check(self._lib.bgamd_env_candidates_read(self._h, 0, total, _ptr(st), _ptr(sq), _ptr(ln), _stream()),
                    "candidates_read")
        return offs, cnts, st, sq, ln

    # -- the env step
    def load_weights(self, weights, slot: int = 0):
        """slot 0 | 1 (head-to-head play keeps one weight set per side).  weights: flat float32[25601] = W1[128,198] | b1[128] | W2[128] | b2[1], or a state_dict with
        fc1.weight / fc1.bias / fc2.weight / fc2.bias (the reference checkpoints, train.py:513-515)."""
        if isinstance(weights, dict):
            weights = np.concatenate([np.asarray(weights[k].detach().cpu().float()).ravel()
                                      for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")])
        w = np.ascontiguousarray(np.asarray(weights, dtype=np.float32).ravel())
        if w.size != 25601:
            raise ValueError("expected 25601 weights (198->128->1)")
        _capi.check(self._lib.bgamd_env_load_weights_slot(self._h, int(slot), w.ctypes.data_as(C.c_void_p)), "load_weights")
        self._has_weights = True

    @staticmethod
    def _flags(roll, auto_reset, no_flip=False, only_player=None, slot=0):
        f = (ROLL if roll else 0) | (AUTO_RESET if auto_reset else 0) | (NO_FLIP if no_flip else 0)
        if only_player is not None:
            f |= ONLY_P1 if int(only_player) == 0 else ONLY_P2
        return f | (WEIGHTS_SLOT1 if slot else 0)

    def step_random(self, roll=True, auto_reset=True, choice_u32=None, no_flip=False, only_player=None, walk=False):
        """walk=True: the one-lane-per-game whole-tree walk instead of the bounded task kernels (same result)."""
        c = None
        if choice_u32 is not None:
            c = torch.as_tensor(np.asarray(choice_u32, dtype=np.uint32).view(np.int32)).to(self.device).contiguous()
        fn = self._lib.bgamd_env_step_random_walk if walk else self._lib.bgamd_env_step_random
        _capi.check(fn(self._h, self._flags(roll, auto_reset, no_flip, only_player), _ptr(c), _stream()), "step_random")
        if c is not None:
            torch.cuda.current_stream().synchronize()

    def step_greedy(self, roll=True, auto_reset=True, epsilon=0.0, precision=F32, no_flip=False, want_index=False,
                    only_player=None, slot=0):
        """want_index: also report the chosen reference-order index and the list length (slow path).
        only_player / slot: head-to-head play -- step only the lanes of that side with weight slot `slot`."""
        _capi.check(self._lib.bgamd_env_step_greedy(self._h, self._flags(roll, auto_reset, no_flip, only_player, slot) |
                                                    (WANT_INDEX if want_index else 0), float(epsilon),
                                                    int(precision), _stream()), "step_greedy")

    def run_greedy(self, n_steps, roll=True, auto_reset=True, epsilon=0.0, precision=F32, only_player=None, slot=0):
        """n_steps greedy steps in one call: the same games as n_steps calls of step_greedy, with the apply of one
        step and the roots of the next sharing a launch."""
        _capi.check(self._lib.bgamd_env_run_greedy(self._h, self._flags(roll, auto_reset, False, only_player, slot),
                                                   float(epsilon), int(precision), int(n_steps), _stream()), "run_greedy")

    def last_choice(self):
        ch, cnt = self._buf((self.n,), torch.int32), self._buf((self.n,), torch.int32)
        sq, ln = self._buf((self.n, 4, 2), torch.int8), self._buf((self.n,), torch.int32)
        val = self._buf((self.n,), torch.float32)
        _capi.check(self._lib.bgamd_env_last_choice(self._h, _ptr(ch), _ptr(cnt), _ptr(sq), _ptr(ln), _ptr(val), _stream()),
                    "last_choice")
        return {"chosen": ch, "count": cnt, "seq": sq, "seq_len": ln, "value": val}

    def unique_rows_info(self):
        """Diagnostics: (game, key | turn << 31) of every afterstate the value net evaluated in the last greedy step
        -> int64 [U, 2] device tensor (columns game, key)."""
        cap = 1 << 16
        while True:
            buf = self._buf((cap, 2), torch.int32)
            n = _capi.check(self._lib.bgamd_env_unique_rows_info(self._h, _ptr(buf), cap, _stream()), "unique_rows_info")
            if n <= cap:
                torch.cuda.current_stream().synchronize()
                return buf[:n].to(torch.int64) & 0xFFFFFFFF
            cap = int(n)

    def unique_rows(self, want_states=True):
        """Every afterstate the value net evaluated in the last greedy step, in arena order:
        -> (info int64 [U, 2] (game, key | turn << 31), states int32 [U, 28] or None, values float32 [U])."""
        info = self.unique_rows_info()
        u = int(info.shape[0])
        st = self._buf((u, 28), torch.int32) if want_states else None
        val = self._buf((u,), torch.float32)
        _capi.check(self._lib.bgamd_env_unique_rows_read(self._h, 0, u, _ptr(st), _ptr(val), _stream()), "unique_rows_read")
        torch.cuda.current_stream().synchronize()
        return info, st, val

    def stats(self):
        out = (C.c_uint64 * 10)()
        _capi.check(self._lib.bgamd_env_stats(self._h, out), "stats")
        k = ("steps", "games_finished", "p1_wins", "candidates_raw", "rows_evaluated", "error_flags",
             "leaf_parent_nodes", "doubles_inner_nodes", "ksteps_executed")
        return dict(zip(k, [int(v) for v in out]))

    def reset_stats(self):
        _capi.check(self._lib.bgamd_env_reset_stats(self._h, _stream()), "reset_stats")

    # -- single-checker surface
    def legal_moves(self, player, die):
        p = self._dev(player, torch.int32, (self.n,))
        d = self._dev(die, torch.int32, (self.n,))
        n, pairs = self._buf((self.n,), torch.int32), self._buf((self.n, 26, 2), torch.int8)
        _capi.check(self._lib.bgamd_env_legal_moves(self._h, _ptr(p), _ptr(d), _ptr(n), _ptr(pairs), _stream()), "legal_moves")
        torch.cuda.current_stream().synchronize()
        return n, pairs

    def try_move(self, player, dice, origin, dest):
        args = [self._dev(a, torch.int32, (self.n,)) for a in (player, dice, origin, dest)]
        err = self._buf((self.n,), torch.int32)
        _capi.check(self._lib.bgamd_env_try_move(self._h, *[_ptr(a) for a in args], _ptr(err), _stream()), "try_move")
        torch.cuda.current_stream().synchronize()
        return err

    # -- trajectory log for the learner
    def record_trajectory(self, max_plies: int | None):
        """Allocates (or drops, with None) the per-turn log: int32 tensor [max_plies, n, 8] of 32-byte rows."""
        if max_plies is None:
            self._traj = None
            _capi.check(self._lib.bgamd_env_set_trajectory(self._h, None, 0), "set_trajectory")
            return None
        self._traj = torch.zeros((int(max_plies), self.n, 8), dtype=torch.int32, device=self.device)
        _capi.check(self._lib.bgamd_env_set_trajectory(self._h, _ptr(self._traj), int(max_plies)), "set_trajectory")
        return self._traj

    def record_ring(self, ring_steps: int | None):
        """Ring log of continuous self-play (bgamd_env_set_trajectory_ring): greedy steps with auto_reset log the pre-move row of every
        lane by ENV STEP -> (rows int32 [ring_steps, n, 8], end int16 [ring_steps, n]); end != 0 where the turn logged in that slot was
        the last of its game: (end & 0x7FFF) logged turns, end < 0: PLAYER2 won.  None drops it."""
        if ring_steps is None:
            self._ring = None
            _capi.check(self._lib.bgamd_env_set_trajectory_ring(self._h, None, 0, None), "set_trajectory_ring")
            return None
        rows = torch.zeros((int(ring_steps), self.n, 8), dtype=torch.int32, device=self.device)
        end = torch.zeros((int(ring_steps), self.n), dtype=torch.int16, device=self.device)
        self._ring = (rows, end)
        _capi.check(self._lib.bgamd_env_set_trajectory_ring(self._h, _ptr(rows), int(ring_steps), _ptr(end)), "set_trajectory_ring")
        return rows, end

    def trajectory_step(self) -> int:
        """env steps logged into the ring so far (a host counter: no synchronisation)"""
        return int(self._lib.bgamd_env_trajectory_step(self._h))

    def progress(self):
        ply, epi = self._buf((self.n,), torch.int32), self._buf((self.n,), torch.int32)
        _capi.check(self._lib.bgamd_env_get_progress(self._h, _ptr(ply), _ptr(epi), _stream()), "get_progress")
        return ply, epi

    def encode_rows(self, rows):
        """rows: int32 [..., 8] device tensor of 32-byte rows -> float32 [..., 198]."""
        r = rows.contiguous()
        n = r.numel() // 8
        out = self._buf((n, 198), torch.float32)
        _capi.check(self._lib.bgamd_encode_rows(_ptr(r), n, _ptr(out), _stream()), "encode_rows")
        return out.reshape(*rows.shape[:-1], 198)

    # -- stateless operators on caller-provided states
    def encode(self, states28, turn):
        st = torch.as_tensor(states28, dtype=torch.int32).to(self.device).contiguous().reshape(-1, 28)
        n = st.shape[0]
        t = self._dev(turn, torch.int32, (n,))
        out = self._buf((n, 198), torch.float32)
        _capi.check(self._lib.bgamd_encode(_ptr(st), _ptr(t), n, _ptr(out), _stream()), "encode")
        torch.cuda.current_stream().synchronize()
        return out

    def evaluate(self, states28, turn, precision=F32, slot=0):
        st = torch.as_tensor(states28, dtype=torch.int32).to(self.device).contiguous().reshape(-1, 28)
        n = st.shape[0]
        t = self._dev(turn, torch.int32, (n,))
        out = self._buf((n,), torch.float32)
        _capi.check(self._lib.bgamd_evaluate_slot(self._h, int(slot), _ptr(st), _ptr(t), n, int(precision), _ptr(out), _stream()),
                    "evaluate")
        torch.cuda.current_stream().synchronize()
        return out

    def evaluate_incremental(self, root_states28, root_turn, states28, root_index, slot=0):
        """Values of afterstates through the greedy step's incremental fp32 path: root position's hidden layer (dense,
        per root) + the W1 columns of the features afterstate i changes against root root_index[i]."""
        rs = torch.as_tensor(root_states28, dtype=torch.int32).to(self.device).contiguous().reshape(-1, 28)
        nr = rs.shape[0]
        rt = self._dev(root_turn, torch.int32, (nr,))
        st = torch.as_tensor(states28, dtype=torch.int32).to(self.device).contiguous().reshape(-1, 28)
        n = st.shape[0]
        ri = self._dev(root_index, torch.int32, (n,))
        out = self._buf((n,), torch.float32)
        _capi.check(self._lib.bgamd_evaluate_incremental(self._h, int(slot), _ptr(rs), _ptr(rt), nr, _ptr(st), _ptr(ri), n,
                                                         _ptr(out), _stream()), "evaluate_incremental")
        torch.cuda.current_stream().synchronize()
        return out

    # -- kernel timing (bench.py)
    def time_kernels(self, enable=True, groups=None, stride=1):
        """groups: iterable of group names to bracket (default: all): enumerate_ordered, eval, apply, step_random,
        expand, leaves, root (the per-game dense pass of the incremental value net)."""
        names = ("enumerate_ordered", "eval", "apply", "step_random", "expand", "leaves", "root")
        arg = int(bool(enable))
        if enable and (groups is not None or stride > 1):
            arg = (sum(1 << names.index(g) for g in (groups if groups is not None else names)) << 8) | (int(stride) << 20)
        _capi.check(self._lib.bgamd_env_time_kernels(self._h, arg), "time_kernels")

    def kernel_choice(self):
        """What the last greedy step launched (bgamd_env_kernel_choice): names of the value-net kernel and of the root pass, whether the root
        pass ran on the env's second stream, the kernel(s) of the expansion below the roots, and whether the library is the experimental build."""
        out = (C.c_int32 * 4)()
        _capi.check(self._lib.bgamd_env_kernel_choice(self._h, out), "kernel_choice")
        ev = ("eval_rows_delta_kernel", "eval_rows_mdelta_kernel", "eval_rows_f32_kernel", "eval_rows_f16x2_kernel", "eval_rows_d16_kernel",
              "eval_rows_bf16_kernel")
        rt = (None, "root_hidden_resident_kernel", "root_hidden_bf16x3_kernel", "eval_rows_f32_kernel<root>", "inside boundary_kernel<true>")
        return {"eval": ev[out[0]], "root": rt[out[1]], "root_on_second_stream": bool(out[2] & 1),
                "expand": "expand_all_kernel" if out[2] & 2 else "doubles_kernel + expand_kernel<LEAF>", "experimental_build": bool(out[3])}

    def kernel_times(self):
        ms, n = (C.c_double * 8)(), (C.c_uint64 * 8)()
        _capi.check(self._lib.bgamd_env_kernel_times(self._h, ms, n), "kernel_times")
        names = ("enumerate_ordered", "eval", "apply", "step_random", "expand", "leaves", "root")
        return {k: {"ms": ms[i], "launches": int(n[i])} for i, k in enumerate(names)}


_POOL = {}          # device index -> idle one-lane envs (a Game that dies hands its env back)


class Game:
    """The reference `Game` (bindings.cpp:62-93) on a one-lane device env.

    One-lane envs are pooled: constructing a Game (or clone(), which reference-style callers do per candidate) reuses
    an idle env instead of allocating a new one, and the getters read a host mirror that ONE snapshot launch fills after
    each mutation (the reference's getters are plain member reads)."""

    _ARENA = 32768          # > the largest doubles enumeration observed (10 063)

    def __init__(self, player: int = 0):
        global _next_scalar_id
        dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
        pool = _POOL.setdefault(dev, [])
        if pool:
            # a fresh game on a pooled env: the dice a NEW env would roll (seed of set_seed(), game id = the running count),
            # last_dice {1,1} (game.hpp:44), no error flags left over from the previous owner -- issued on the NULL stream,
            # where the host-argument surface (bgamd_game_*) runs: a caller inside `with torch.cuda.stream(s)` must not see
            # the reset overtake the set_state that follows
            self._v = pool.pop()
            self._order()                # ... and not overtake what THIS thread still has queued on its current stream either
            self._v.reseed(_seed, _next_scalar_id, 1 << 40, null_stream=True)
            _capi.check(self._v._lib.bgamd_env_reset_stats(self._v._h, None), "reset_stats")
        else:
            self._v = VecGame(1, device=dev, seed=_seed, lane_offset=_next_scalar_id, lane_stride=1 << 40,
                              arena_rows=self._ARENA)
        _next_scalar_id += 1
        self._players = [None, None]
        self._snap = None
        self._put(None, int(player) % 2)                     # Game::Game(int): turn = parity (game.cpp:44-53)

    def __del__(self):
        v = getattr(self, "_v", None)
        if v is not None and getattr(v, "_h", None) and _POOL is not None:
            self._v = None
            try:                                             # work the dying Game queued on a side stream (make_move's step) must be done
                self._order()                                # before the env's next owner resets it on the NULL stream
            except Exception:
                pass
            pool = _POOL.setdefault(v.device.index or 0, [])
            if len(pool) < 64:
                pool.append(v)
            else:
                v.close()

    def _dirty(self):
        self._snap = None

    @staticmethod
    def _order():
        """The bgamd_game_* calls run on the NULL stream and torch's side streams are non-blocking: work this thread queued
        on its current stream for the same env (step_greedy of make_move, last_choice) has to be done first."""
        st = torch.cuda.current_stream()
        if st != torch.cuda.default_stream(st.device):
            st.synchronize()

    def _s(self):
        """state28 | turn | die1 | die2 | flags: ONE call of the host-argument scalar surface (bgamd_game_snapshot)."""
        if self._snap is None:
            self._order()
            buf = (C.c_int32 * 32)()
            _capi.check(self._v._lib.bgamd_game_snapshot(self._v._h, buf), "snapshot")
            self._snap = list(buf)
        return self._snap

    def _put(self, state28=None, turn=-1):
        arr = (C.c_int32 * 28)(*[int(v) for v in state28]) if state28 is not None else None
        self._order()
        _capi.check(self._v._lib.bgamd_game_set_state(self._v._h, arr, int(turn)), "set_state")
        self._dirty()

    # players ------------------------------------------------------------------------------
    def setPlayers(self, p1: Player, p2: Player):
        self._players = [p1, p2]                              # kept alive here (reference keeps raw pointers)

    def getPlayers(self, num):
        p = self._players[0 if int(num) == 0 else 1]
        return Player(p.getName(), PlayerType(p.getNum()))    # by-value copy, as the binding returns

    # turn / board ---------------------------------------------------------------------------
    def getTurn(self):
        return self._s()[28]

    def setTurn(self, turn):
        self._put(None, int(turn) & 1)

    def _state(self):
        return list(self._s()[:28])

    def getGameBoard(self):
        return self._state()[:24]

    def getPieces(self):
        return Pieces(self)

    def getJailedCount(self, player):
        return self._s()[24 + (0 if int(player) == 0 else 1)]

    def getBornOffCount(self, player):
        return self._s()[26 + (0 if int(player) == 0 else 1)]

    def setGameBoard(self, board):
        board = [int(v) for v in board]
        if len(board) != 24:
            raise ValueError("gameboard must have 24 entries")
        s = self._state()
        self._put(board + s[24:])

    def setBorneOffPieces(self, player, num):
        s = self._state()
        s[26 + (0 if int(player) == 0 else 1)] = int(num)
        self._put(s)

    def _set_jailed(self, player, num):
        """Not in the reference binding (bar counts are only reachable by hits); used by tests/clone."""
        s = self._state()
        s[24 + (0 if int(player) == 0 else 1)] = int(num)
        self._put(s)

    def reset(self):
        self.populateBoard()                                   # binding maps reset -> populateBoard only

    def populateBoard(self):
        s = self._state()
        self._put([2, 0, 0, 0, 0, -5, 0, -3, 0, 0, 0, 5, -5, 0, 0, 0, 3, 0, 5, 0, 0, 0, 0, -2] + s[24:])

    def printGameBoard(self):
        s = self._state()
        print("board", s[:24], "| jail", s[24:26], "| free", s[26:28])

    # dice -------------------------------------------------------------------------------------
    def setDice(self, d1, d2):
        self._order()
        _capi.check(self._v._lib.bgamd_game_set_dice(self._v._h, int(d1), int(d2)), "setDice")
        self._dirty()

    def roll_dice(self):
        d = (C.c_int32 * 2)()
        self._order()
        _capi.check(self._v._lib.bgamd_game_roll(self._v._h, d), "roll_dice")
        self._dirty()
        return [int(d[0]), int(d[1])]

    def get_last_dice(self):
        d = self._s()[29:31]
        return list(d) if d[0] else [1, 1]

    # rules --------------------------------------------------------------------------------------
    def legalMoves(self, player, die):
        pairs = (C.c_int8 * 52)()
        self._order()
        k = _capi.check(self._v._lib.bgamd_game_legal_moves(self._v._h, int(player), int(die), pairs), "legalMoves")
        return [(int(pairs[2 * i]), int(pairs[2 * i + 1])) for i in range(k)]

    def _enumerate(self, player, d1, d2, want_states=True):
        """One call of bgamd_game_enumerate when the list fits the first guess, a second one otherwise."""
        cap = 2048
        for _ in range(2):
            st = np.empty((cap, 28), dtype=np.int32) if want_states else None
            sq, ln = np.empty((cap, 4, 2), dtype=np.int8), np.empty((cap,), dtype=np.int32)
            self._order()
            k = _capi.check(self._v._lib.bgamd_game_enumerate(self._v._h, int(player), int(d1), int(d2),
                                                             st.ctypes.data_as(C.c_void_p) if want_states else None,
                                                             sq.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p), cap), "enumerate")
            if k <= cap:
                break
            cap = k
        sql, lnl = sq[:k].tolist(), ln[:k].tolist()
        seqs = [[(sql[i][j][0], sql[i][j][1]) for j in range(lnl[i])] for i in range(k)]
        return seqs, (st[:k].copy() if want_states else None)

    def legalTurnSequences(self, player, die1, die2):
        return self._enumerate(player, die1, die2, want_states=False)[0]

    def evaluateTurnSequences(self, player, die1, die2):
        return self._enumerate(player, die1, die2)

    def tryMove(self, player: Player, dice, origin, dest):
        self._order()
        err = _capi.check(self._v._lib.bgamd_game_try_move(self._v._h, player.getNum(), int(dice), int(origin), int(dest)), "tryMove")
        if err == 0:
            self._dirty()
        return err == 0, ERR_MESSAGES[err]

    def is_game_over(self):
        f = self._s()[31]
        return (True, (f >> 1) & 1) if f & 1 else (False, -1)

    def clone(self):
        g = Game(0)
        g._players = list(self._players)
        s = self._s()
        g._put(s[:28], s[28])                                  # last_dice stays [1,1] (game.cpp:68-77)
        return g
