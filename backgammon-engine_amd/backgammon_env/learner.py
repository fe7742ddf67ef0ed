"""TD(λ) learner: the reference's `apply_td_updates` (pysrc/TD(λ) model/
train.py:124-172, traces model.py:48-53, schedules model.py:69-73) as batched PyTorch-ROCm tensor
algebra, lock-step over the turn index of many games, with ONE all-reduce of the 25 601-float update
per training step.

Closed form of one step for a game g (identical to autograd on sigmoid(fc2(sigmoid(fc1(x))))):
    a = W1 x + b1, h = σ(a), v = σ(W2·h + b2), g = v(1-v)
    ∇b2 = g, ∇W2 = g h, ∇b1 = g W2 ⊙ h ⊙ (1-h), ∇W1 = ∇b1 ⊗ x
    e ← λ e + ∇ ;  δ = V(s_{t+1}) - V(s_t)  (terminal: z - V(s_{T-1}), z = 1 if PLAYER1 won)
    θ ← θ + Σ_g (α δ_g) e_g                                  (train.py:136-147,165-170)
For ONE game this is the reference's update step for step (verified against fixture G6); for many
games the per-step updates of all games are summed (mini-batch TD(λ)) -- the documented deviation
from the reference, which replays games one after another.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def stream_schedule(lengths, slots: int):
    """The streamed replay's schedule (the host restatement of bgamd_td_stream_schedule, include/bgamd.h: the two must agree,
    tests/test_learner_cpu.py): games (lanes with length > 0) are dealt longest first (ties: the lower lane), each to the slot with the
    fewest turns so far (ties: the lower slot), so the slots' totals end up within a short game of each other and the replay
    takes max-total steps with every slot busy almost to the end; every slot plays its share in a fixed pseudo-random order.
    -> (queue int32 [games] lanes in slot-major play order, qoff int32 [k + 1], n_steps = the longest slot total, k)."""
    import heapq
    lengths = torch.as_tensor(lengths)
    dev = lengths.device
    ln = lengths.cpu().to(torch.int64).tolist()
    games = sorted((l for l in range(len(ln)) if ln[l] > 0), key=lambda l: -ln[l])     # stable: ties keep the lower lane first
    k = max(1, min(int(slots), len(games)))
    heap = [(0, i) for i in range(k)]
    share = [[] for _ in range(k)]
    n_steps = 0
    for g in games:
        tot, i = heapq.heappop(heap)
        share[i].append(g)
        tot += ln[g]
        n_steps = max(n_steps, tot)
        heapq.heappush(heap, (tot, i))
    queue, qoff = [], [0]
    for sh in share:
        queue += sorted(sh, key=lambda l: (l * 2654435761 + 0x9E3779B9) % 4294967296)
        qoff.append(len(queue))
    return (torch.tensor(queue, dtype=torch.int32, device=dev), torch.tensor(qoff, dtype=torch.int32, device=dev),
            n_steps, (k if games else 0))


class TDLambdaLearner:
    def __init__(self, weights_flat, device="cpu", alpha: float = 0.1, lam: float = 0.7, dtype=torch.float32):
        w = torch.as_tensor(weights_flat, dtype=dtype).flatten().to(device)
        if w.numel() != 25601:
            raise ValueError("expected 25601 weights")
        self.theta = w.clone()                      # flat: W1[128,198] | b1[128] | W2[128] | b2[1]
        self.learning_rate, self.lambda_decay = float(alpha), float(lam)

    # -- views into the flat parameter / trace vectors
    @staticmethod
    def _split(flat):
        lead = flat.shape[:-1]
        return (flat[..., :25344].reshape(*lead, 128, 198), flat[..., 25344:25472],
                flat[..., 25472:25600], flat[..., 25600:])

    def update_learning_params(self, episode: int):            # model.py:69-73
        self.learning_rate = max(0.01, 0.1 * (0.96 ** (episode // 40000)))
        self.lambda_decay = max(0.7, 0.9 * (0.96 ** (episode // 30000)))

    def values(self, X):
        W1, b1, W2, b2 = self._split(self.theta)
        h = torch.sigmoid(X @ W1.T + b1)
        return torch.sigmoid(h @ W2 + b2), h

    def replay(self, X, lengths, p1_won, group=None, batch_scale: float = 1.0):
        """X: float [T, G, 198] encodings of the pre-move states of every turn (turn bit = side to move),
        lengths: int [G] number of recorded turns per game, p1_won: bool/int [G].
        Applies the lock-step TD(λ) updates in place; returns the summed squared TD error and count.
        batch_scale multiplies every game's update.  1.0 = the plain sum, which matches the reference to first
        order for rounds of the reference's size (round_size = workers <= 24, train.py:307-312); a round of
        tens of thousands of games moves the weights that many times further per step, so large rounds want
        batch_scale ~ 24 / G (the update of an average reference-sized round)."""
        T, G = X.shape[0], X.shape[1]
        dev, dt = self.theta.device, self.theta.dtype
        lengths = torch.as_tensor(lengths, device=dev).long()
        z = torch.as_tensor(p1_won, device=dev).to(dt)
        e = torch.zeros((G, 25601), dtype=dt, device=dev)       # traces are reset per game (train.py:539-540)
        eW1, eb1, eW2, eb2 = self._split(e)
        sq = torch.zeros((), dtype=torch.float64, device=dev)
        cnt = torch.zeros((), dtype=torch.int64, device=dev)
        # BGAMD_FORCE_COLLECTIVE=1: issue the per-step all-reduce on a group of ONE rank too (tools/train_dist_step.py measures what the
        # collective's launch costs a training step with the RCCL communicator present; more ranks are the driver's to launch)
        distributed = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or
                                                                         os.environ.get("BGAMD_FORCE_COLLECTIVE") == "1")
        n_steps = torch.tensor([min(T, int(lengths.max().item()) if G else 0)], dtype=torch.int64, device=dev)
        if distributed:                      # every rank must issue the same number of all-reduces
            dist.all_reduce(n_steps, op=dist.ReduceOp.MAX, group=group)
        for t in range(int(n_steps.item())):
            if t >= T:                       # this rank's log is shorter: it still joins the collective
                upd = torch.zeros(25601, dtype=dt, device=dev)
                dist.all_reduce(upd, op=dist.ReduceOp.SUM, group=group)
                self.theta.add_(upd)
                continue
            active = (lengths > t)
            terminal = (lengths == t + 1)
            W1, b1, W2, b2 = self._split(self.theta)
            x = X[t].to(dt)                                         # X may be kept in fp32 (it is exact there) under a float64 replay
            v, h = self.values(x)
            if t + 1 < T:
                v_next, _ = self.values(X[t + 1].to(dt))
            else:
                v_next = torch.zeros_like(v)
            delta = torch.where(terminal, z - v, v_next - v)
            delta = torch.where(active, delta, torch.zeros_like(delta))
            g = v * (1 - v) * active.to(dt)
            db1 = (g[:, None] * W2[None, :]) * h * (1 - h)
            lam = self.lambda_decay
            eb2.mul_(lam).add_(g[:, None])
            eW2.mul_(lam).addcmul_(g[:, None], h)
            eb1.mul_(lam).add_(db1)
            # e_W1 <- lam * e_W1 + db1 (x) x as ONE batched rank-1 update: a single read+write pass over the
            # G x 128 x 198 traces (the traffic that bounds the learner, SURVEY hard part 5)
            eW1.baddbmm_(db1[:, :, None], x[:, None, :], beta=lam, alpha=1.0)
            # α·δ is formed in float64 by the reference (python floats, train.py:147) before the fp32 multiply
            coef = (self.learning_rate * batch_scale * delta.double()).to(dt)
            upd = coef @ e                                          # Σ_g (α δ_g) e_g : [25601]
            if distributed:
                dist.all_reduce(upd, op=dist.ReduceOp.SUM, group=group)   # the ONE collective per training step
            self.theta.add_(upd)
            sq += (delta.double() ** 2).sum()
            cnt += active.sum()                                     # no host sync inside the loop
        return float(sq.item()), int(cnt.item())

    def replay_stream(self, X, lengths, p1_won, slots: int, batch_scale: float = 1.0, delay: int = 0):
        """The streamed replay (DeviceTDLambdaLearner.replay_rows(slots=k), bgamd_td_begin_stream) as a host closed form:
        slot i replays its queue of games one after another, every step sums the updates of the slots' current turns.
        delay = 1: the update of step t is applied one step late (bgamd_td_set_delay): step t + 1 runs on the weights of step t plus the
        update of step t - 1; the last update is applied after the last step."""
        T, G = X.shape[0], X.shape[1]
        dev, dt = self.theta.device, self.theta.dtype
        lengths = torch.as_tensor(lengths, device=dev).long()
        z_all = torch.as_tensor(p1_won, device=dev).to(dt)
        queue, qoff, n_steps, k = stream_schedule(lengths, slots)
        queue, qoff = queue.cpu().tolist(), qoff.cpu().tolist()
        lens = lengths.cpu().tolist()
        game = torch.full((n_steps, k), -1, dtype=torch.long)       # the game a slot replays at a step, and that game's own step
        tl = torch.zeros((n_steps, k), dtype=torch.long)
        for i in range(k):
            s0 = 0
            for q in range(qoff[i], qoff[i + 1]):
                n = lens[queue[q]]
                game[s0:s0 + n, i] = queue[q]
                tl[s0:s0 + n, i] = torch.arange(n)
                s0 += n
        game, tl = game.to(dev), tl.to(dev)
        e = torch.zeros((k, 25601), dtype=dt, device=dev)
        eW1, eb1, eW2, eb2 = self._split(e)
        sq, cnt = 0.0, 0
        held = None
        for s in range(n_steps):
            gm, t = game[s], tl[s]
            active = gm >= 0
            gi = gm.clamp(min=0)
            e[t == 0] = 0                                            # a slot's trace restarts with each game (train.py:133)
            terminal = active & (lengths[gi] == t + 1)
            W1, b1, W2, b2 = self._split(self.theta)
            x = X[t.clamp(max=T - 1), gi].to(dt) * active.to(dt)[:, None]
            v, h = self.values(x)
            v_next, _ = self.values(X[(t + 1).clamp(max=T - 1), gi].to(dt))
            delta = torch.where(terminal, z_all[gi] - v, v_next - v) * active.to(dt)
            g = v * (1 - v) * active.to(dt)
            db1 = (g[:, None] * W2[None, :]) * h * (1 - h)
            lam = self.lambda_decay
            eb2.mul_(lam).add_(g[:, None])
            eW2.mul_(lam).addcmul_(g[:, None], h)
            eb1.mul_(lam).add_(db1)
            eW1.baddbmm_(db1[:, :, None], x[:, None, :], beta=lam, alpha=1.0)
            coef = (self.learning_rate * batch_scale * delta.double()).to(dt)
            upd = coef @ e
            if delay:
                if held is not None:
                    self.theta.add_(held)
                held = upd
            else:
                self.theta.add_(upd)
            sq += float((delta.double() ** 2).sum().item())
            cnt += int(active.sum().item())
        if delay and held is not None:
            self.theta.add_(held)
        return sq, cnt

    def state_dict(self):
        W1, b1, W2, b2 = TDLambdaLearner._split(self.theta.detach().cpu())
        return {"fc1.weight": W1.clone(), "fc1.bias": b1.clone(), "fc2.weight": W2.reshape(1, 128).clone(),
                "fc2.bias": b2.clone()}


class DeviceTDLambdaLearner:
    """The same lock-step TD(λ) replay as TDLambdaLearner.replay, in hand-written HIP kernels (csrc/bg_learner.h,
    C ABI bgamd_td_*): it reads the env's 32-byte trajectory rows directly (no [T, G, 198] tensor), keeps the
    per-game traces in HBM and touches them once per step (read + write, the Σ_g fp32(αδ_g)·e_g reduction fused in
    the same pass).  No CPU fallback: needs the HIP library and a device."""

    def __init__(self, weights_flat, max_games: int, device=None, alpha: float = 0.1, lam: float = 0.7):
        import ctypes as C
        from . import _capi
        self._C, self._capi, self._lib = C, _capi, _capi.load()
        if not torch.cuda.is_available():
            raise _capi.BgamdError("DeviceTDLambdaLearner needs a gfx950 device (no CPU fallback)")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.max_games = int(max_games)
        h = C.c_void_p()
        _capi.check(self._lib.bgamd_td_create(C.byref(h), self.max_games, self.device.index or 0), "td_create")
        self._h = h
        self.learning_rate, self.lambda_decay = float(alpha), float(lam)
        self._keep = None
        self.set_weights(weights_flat)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.bgamd_td_destroy(h)

    def _p(self, t):
        return self._C.c_void_p(t.data_ptr())

    def _s(self):
        return self._C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def init_collective(self, group=None):
        """Gives the learner an RCCL communicator of its own (bgamd_td_comm_init) over the ranks of `group`: the per-step all-reduce of
        a distributed replay is then issued by the library, on the learner's stream, between the step's kernels and the apply kernel
        (bgamd_td_replay_allreduce) -- no torch dispatch per training step.  The 128-byte id travels through `group` (any backend).
        A world of one rank gets a communicator too (what tools/train_dist_step.py measures)."""
        C, lib, chk = self._C, self._lib, self._capi.check
        rank, world = 0, 1
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = (C.c_uint8 * 128)()
        if rank == 0:
            chk(lib.bgamd_td_comm_unique_id(buf), "td_comm_unique_id")
        if world > 1:
            cuda = dist.get_backend(group) == "nccl"
            t = torch.tensor(list(buf), dtype=torch.uint8, device=self.device if cuda else "cpu")
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            buf = (C.c_uint8 * 128)(*t.cpu().tolist())
        torch.cuda.synchronize(self.device)
        chk(lib.bgamd_td_comm_init(self._h, buf, rank, world), "td_comm_init")
        self._comm = (rank, world)

    @staticmethod
    def _world(group):
        return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1

    def _distributed(self, group):
        """the replay issues a collective per training step: more than one rank, or BGAMD_FORCE_COLLECTIVE=1 on a world of one (what
        tools/train_dist_step.py and the world-1 test measure: the cost of the call with the communicator present)"""
        forced = os.environ.get("BGAMD_FORCE_COLLECTIVE") == "1"
        return self._world(group) > 1 or (forced and ((dist.is_available() and dist.is_initialized()) or getattr(self, "_comm", None) is not None))

    def _in_library(self, distributed):
        """the library's own collective serves this replay: the learner has a communicator and the replay is a distributed one"""
        return distributed and getattr(self, "_comm", None) is not None and os.environ.get("BGAMD_TD_TORCH_COLLECTIVE") != "1"

    def set_delay(self, delay: int):
        """delay = 1: streamed replays through 512 ... 4 096 slots apply the update of step t one step late, and a training step is one launch
        (bgamd_td_set_delay, include/bgamd.h: a documented deviation, opt-in; 0 = exact)."""
        self._capi.check(self._lib.bgamd_td_set_delay(self._h, int(delay)), "td_set_delay")
        self.update_delay = int(delay)

    def set_weights(self, weights_flat):
        w = torch.as_tensor(weights_flat, dtype=torch.float32).flatten().to(self.device).contiguous()
        if w.numel() != 25601:
            raise ValueError("expected 25601 weights")
        self._capi.check(self._lib.bgamd_td_set_weights(self._h, self._p(w), self._s()), "td_set_weights")
        torch.cuda.current_stream(self.device).synchronize()

    @property
    def theta(self):
        out = torch.empty(25601, dtype=torch.float32, device=self.device)
        self._capi.check(self._lib.bgamd_td_get_weights(self._h, self._p(out), self._s()), "td_get_weights")
        return out

    update_learning_params = TDLambdaLearner.update_learning_params

    def state_dict(self):
        return TDLambdaLearner.state_dict(self)

    def time_trace_kernel(self, enable=True):
        self._capi.check(self._lib.bgamd_td_time(self._h, 1 if enable else 0), "td_time")

    def active_columns(self):
        """Σ over the (game, step) updates of the last replay (the last sub-round) of the W1 trace columns touched (of 198)."""
        c = self._C.c_uint64()
        self._capi.check(self._lib.bgamd_td_active_columns(self._h, self._C.byref(c)), "td_active_columns")
        return int(c.value)

    def written_columns(self):
        """... of the W1 trace columns written (lazily scaled traces: only where x_j != 0, except on the passes that fold the
        scale back in)."""
        c = self._C.c_uint64()
        self._capi.check(self._lib.bgamd_td_written_columns(self._h, self._C.byref(c)), "td_written_columns")
        return int(c.value)

    def trace_kernel_times(self):
        C = self._C
        ms, n, gs = C.c_double(), C.c_uint64(), C.c_uint64()
        self._capi.check(self._lib.bgamd_td_times(self._h, C.byref(ms), C.byref(n), C.byref(gs)), "td_times")
        return ms.value, n.value, gs.value

    def replay_rows(self, rows, lengths, p1_won, group=None, batch_scale: float = 1.0, split_apply: bool = False,
                    sub_round: int = 0, slots: int = 0):
        """rows: int32 [T, n, 8] trajectory log (VecGame.record_trajectory / play_round), lengths: int [n] logged
        turns per lane (0 = do not replay), p1_won: bool [n].  Returns (Σ δ², number of (game, step) updates).
        split_apply: take the step / all-reduce / apply route of the distributed replay even on one rank.
        sub_round = k > 0: the round is replayed as ceil(games / k) sub-rounds of <= k games, one after another, every
        sub-round from the weights the one before left (the reference applies a round's games one after another,
        train.py:536-547; k = 1 is exactly that, k = games is one summed update per step).  The same trace traffic
        whatever k is; sub-rounds take every ceil(games / k)-th game of the length-sorted order, so each has the round's
        mix of game lengths.
        slots = k > 0: the STREAMED replay -- k slots replay the round's games one after another (a slot starts its next
        game the step after its last one ended), so every training step sums the updates of k games at different plies and
        the round takes ~(turns of the round) / k steps with all k slots busy, instead of ceil(games / k) sub-rounds of
        (longest game) steps that mostly run nearly empty.  Games are dealt to the slots longest first in a snake (equal
        slot totals), each slot plays its share in a fixed pseudo-random order."""
        rows = rows.contiguous()
        T, n = int(rows.shape[0]), int(rows.shape[1])
        lengths = torch.as_tensor(lengths, device=self.device).to(torch.int32).contiguous()
        won = torch.as_tensor(p1_won, device=self.device).to(torch.uint8).contiguous()
        if int(lengths.max().item()) > T:
            raise ValueError("a game is longer than the trajectory log")
        sl, order = torch.sort(lengths, descending=True, stable=True)
        n_games = int((sl > 0).sum().item())
        # BGAMD_FORCE_COLLECTIVE=1: issue the per-step all-reduce on a group of ONE rank too (tools/train_dist_step.py measures what the
        # collective's launch costs a training step with the RCCL communicator present; more ranks are the driver's to launch)
        distributed = self._distributed(group)
        if slots and slots > 0:
            return self._replay_stream(rows, T, n, lengths, won, order[:n_games], sl[:n_games], int(slots), group, distributed,
                                       batch_scale, split_apply)
        n_sub = 1
        if sub_round and sub_round > 0:
            n_sub = max(1, -(-n_games // int(sub_round)))
        if distributed and self._world(group) > 1:              # every rank runs the same number of sub-rounds
            ns = torch.tensor([n_sub], dtype=torch.int64, device=self.device)
            dist.all_reduce(ns, op=dist.ReduceOp.MAX, group=group)
            n_sub = int(ns.item())
        sq_tot, cnt_tot = 0.0, 0
        for c in range(n_sub):
            o = order[:n_games][c::n_sub].to(torch.int32).contiguous()
            sq, cnt = self._replay_order(rows, T, n, lengths, won, o, sl[:n_games][c::n_sub], group, distributed,
                                         batch_scale, split_apply)
            sq_tot += sq
            cnt_tot += cnt
        return sq_tot, cnt_tot

    def replay_games(self, ring_rows, game_lane, game_start, lengths, p1_won, slots: int, group=None, batch_scale: float = 1.0):
        """Streamed replay over a GAME TABLE and a ring log (continuous self-play: VecGame.record_ring, ContinuousSelfPlay.finished):
        game i sits in column game_lane[i] of ring_rows [R, n, 8], its turn k in ring slot (game_start[i] + k) % R, lengths / p1_won are
        per game.  The same schedule, kernels and arithmetic as replay_rows(slots=k): a table with one game per lane starting in slot 0
        IS replay_rows.  Returns (Σ δ², number of (game, step) updates)."""
        C, lib, chk = self._C, self._lib, self._capi.check
        ring_rows = ring_rows.contiguous()
        R, n = int(ring_rows.shape[0]), int(ring_rows.shape[1])
        lane = torch.as_tensor(game_lane, device=self.device).to(torch.int32).contiguous()
        start = torch.as_tensor(game_start, device=self.device).to(torch.int32).contiguous()
        lengths = torch.as_tensor(lengths, device=self.device).to(torch.int32).contiguous()
        won = torch.as_tensor(p1_won, device=self.device).to(torch.uint8).contiguous()
        G = int(lengths.numel())
        if not (lane.numel() == start.numel() == won.numel() == G):
            raise ValueError("game table columns differ in length")
        if G and int(lengths.max().item()) > R:
            raise ValueError("a game is longer than the ring log")
        distributed = self._distributed(group)
        h_len = lengths.cpu().numpy().astype(np.int32, copy=False) if G else np.zeros(1, dtype=np.int32)
        k = max(1, min(int(slots), self.max_games))
        h_queue, h_qoff = np.zeros(max(G, 1), dtype=np.int32), np.zeros(k + 1, dtype=np.int32)
        ng, nst = C.c_int64(), C.c_int64()
        chk(lib.bgamd_td_stream_schedule(h_len.ctypes.data, G, k, h_queue.ctypes.data, h_qoff.ctypes.data, C.byref(ng), C.byref(nst)),
            "td_stream_schedule")
        k = min(k, int(ng.value))
        n_steps = int(nst.value)
        queue = torch.from_numpy(h_queue[:max(int(ng.value), 1)]).to(self.device)
        qoff = torch.from_numpy(h_qoff[:k + 1]).to(self.device)
        if G == 0:                                                   # (a rank without finished games still joins the collectives below)
            lane = start = lengths = torch.zeros(1, dtype=torch.int32, device=self.device)
            won = torch.zeros(1, dtype=torch.uint8, device=self.device)
        self._keep = (ring_rows, lane, start, lengths, won, queue, qoff)
        chk(lib.bgamd_td_begin_stream_games(self._h, self._p(ring_rows), R, n, self._p(queue), self._p(qoff), k, max(G, 1), self._p(lane),
                                            self._p(start), self._p(lengths), self._p(won), self._s()), "td_begin_stream_games")
        alpha = float(self.learning_rate) * float(batch_scale)
        lam = float(self.lambda_decay)
        if not distributed:
            arr = (C.c_int64 * max(n_steps, 1))(*([k] * n_steps))
            chk(lib.bgamd_td_replay(self._h, n_steps, arr, alpha, lam, self._s()), "td_replay")
        else:
            ns = torch.tensor([n_steps], dtype=torch.int64, device=self.device)
            if self._world(group) > 1:
                dist.all_reduce(ns, op=dist.ReduceOp.MAX, group=group)
            if self._in_library(True):
                own = n_steps if G else 0
                arr = (C.c_int64 * max(own, 1))(*([k] * own))
                chk(lib.bgamd_td_replay_allreduce(self._h, int(ns.item()), arr, own, alpha, lam, self._s()), "td_replay_allreduce")
            else:
                upd = torch.zeros(25601, dtype=torch.float32, device=self.device)
                for t in range(int(ns.item())):
                    if G and t < n_steps:
                        chk(lib.bgamd_td_step(self._h, t, k, alpha, lam, self._p(upd), self._s()), "td_step")
                    else:
                        upd.zero_()
                    dist.all_reduce(upd, op=dist.ReduceOp.SUM, group=group)
                    chk(lib.bgamd_td_apply(self._h, self._p(upd), self._s()), "td_apply")
        sq, cnt = C.c_double(), C.c_int64()
        chk(lib.bgamd_td_stats(self._h, C.byref(sq), C.byref(cnt)), "td_stats")
        self._keep = None
        return float(sq.value), int(cnt.value)

    def _replay_stream(self, rows, T, n, lengths, won, order, sl, slots, group, distributed, batch_scale, split_apply):
        """Streamed replay (bgamd_td_begin_stream): order = lanes by decreasing length sl."""
        C, lib, chk = self._C, self._lib, self._capi.check
        n_games = int(order.numel())
        # the schedule comes from the library (host code: a copy of the lengths goes down, the two index arrays come up)
        h_len = lengths.cpu().numpy().astype(np.int32, copy=False)
        k = max(1, min(int(slots), self.max_games))
        h_queue, h_qoff = np.zeros(max(n, 1), dtype=np.int32), np.zeros(k + 1, dtype=np.int32)
        ng, nst = C.c_int64(), C.c_int64()
        chk(lib.bgamd_td_stream_schedule(h_len.ctypes.data, n, k, h_queue.ctypes.data, h_qoff.ctypes.data, C.byref(ng), C.byref(nst)),
            "td_stream_schedule")
        k = min(k, int(ng.value))
        n_steps = int(nst.value)
        queue = torch.from_numpy(h_queue[:max(int(ng.value), 1)]).to(self.device)
        qoff = torch.from_numpy(h_qoff[:k + 1]).to(self.device)
        self._keep = (rows, lengths, won, queue, qoff)
        chk(lib.bgamd_td_begin_stream(self._h, self._p(rows), T, n, self._p(queue), self._p(qoff), k,
                                      self._p(lengths), self._p(won), self._s()), "td_begin_stream")
        alpha = float(self.learning_rate) * float(batch_scale)
        lam = float(self.lambda_decay)
        if not distributed and not split_apply:
            arr = (C.c_int64 * max(n_steps, 1))(*([k] * n_steps))
            chk(lib.bgamd_td_replay(self._h, n_steps, arr, alpha, lam, self._s()), "td_replay")
        elif self._in_library(distributed):
            ns = torch.tensor([n_steps], dtype=torch.int64, device=self.device)
            if self._world(group) > 1:
                dist.all_reduce(ns, op=dist.ReduceOp.MAX, group=group)      # every rank issues the same collectives
            own = n_steps if n_games else 0
            arr = (C.c_int64 * max(own, 1))(*([k] * own))
            chk(lib.bgamd_td_replay_allreduce(self._h, int(ns.item()), arr, own, alpha, lam, self._s()), "td_replay_allreduce")
        else:
            ns = torch.tensor([n_steps], dtype=torch.int64, device=self.device)
            if distributed:
                dist.all_reduce(ns, op=dist.ReduceOp.MAX, group=group)  # every rank issues the same collectives
            upd = torch.zeros(25601, dtype=torch.float32, device=self.device)
            for t in range(int(ns.item())):
                if n_games:
                    chk(lib.bgamd_td_step(self._h, t, k, alpha, lam, self._p(upd), self._s()), "td_step")
                else:
                    upd.zero_()
                if distributed:
                    dist.all_reduce(upd, op=dist.ReduceOp.SUM, group=group)  # the ONE collective per training step
                chk(lib.bgamd_td_apply(self._h, self._p(upd), self._s()), "td_apply")
        sq, cnt = C.c_double(), C.c_int64()
        chk(lib.bgamd_td_stats(self._h, C.byref(sq), C.byref(cnt)), "td_stats")
        self._keep = None
        return float(sq.value), int(cnt.value)

    def _replay_order(self, rows, T, n, lengths, won, order, sl, group, distributed, batch_scale, split_apply):
        """Lock-step replay of the games order[...] (lanes, by decreasing length sl)."""
        C, lib, chk = self._C, self._lib, self._capi.check
        n_games = int(order.numel())
        if n_games > self.max_games:
            raise ValueError(f"{n_games} games > max_games={self.max_games}")
        n_steps = int(sl[0].item()) if n_games else 0
        # running games per step: a prefix of the order
        hist = torch.bincount(sl.long(), minlength=n_steps + 1)
        n_active = (n_games - torch.cumsum(hist, 0)[:n_steps]).cpu().tolist()      # games with length > t
        self._keep = (rows, lengths, won, order)
        chk(lib.bgamd_td_begin(self._h, self._p(rows), T, n, self._p(order), n_games, self._p(lengths), self._p(won),
                               self._s()), "td_begin")
        alpha = float(self.learning_rate) * float(batch_scale)
        lam = float(self.lambda_decay)
        if not distributed and not split_apply:
            arr = (C.c_int64 * max(n_steps, 1))(*n_active)
            chk(lib.bgamd_td_replay(self._h, n_steps, arr, alpha, lam, self._s()), "td_replay")
        elif self._in_library(distributed):
            ns = torch.tensor([n_steps], dtype=torch.int64, device=self.device)
            if self._world(group) > 1:
                dist.all_reduce(ns, op=dist.ReduceOp.MAX, group=group)      # every rank issues the same collectives
            arr = (C.c_int64 * max(n_steps, 1))(*n_active)
            chk(lib.bgamd_td_replay_allreduce(self._h, int(ns.item()), arr, n_steps, alpha, lam, self._s()), "td_replay_allreduce")
        else:
            ns = torch.tensor([n_steps], dtype=torch.int64, device=self.device)
            if distributed:
                dist.all_reduce(ns, op=dist.ReduceOp.MAX, group=group)  # every rank issues the same collectives
            upd = torch.zeros(25601, dtype=torch.float32, device=self.device)
            for t in range(int(ns.item())):
                if t < n_steps:
                    chk(lib.bgamd_td_step(self._h, t, n_active[t], alpha, lam, self._p(upd), self._s()), "td_step")
                else:
                    upd.zero_()
                if distributed:
                    dist.all_reduce(upd, op=dist.ReduceOp.SUM, group=group)  # the ONE collective per training step
                chk(lib.bgamd_td_apply(self._h, self._p(upd), self._s()), "td_apply")
        sq, cnt = C.c_double(), C.c_int64()
        chk(lib.bgamd_td_stats(self._h, C.byref(sq), C.byref(cnt)), "td_stats")
        self._keep = None
        return float(sq.value), int(cnt.value)


def play_round(env, max_plies: int = 512, epsilon: float = 0.0, precision=0, episode=None):
    """One round of self-play from a frozen weight snapshot (train.py:527-547 semantics): every lane plays
    ONE game to the end, turns are logged. -> (rows [T, n, 8] int32, lengths [n], p1_won [n] bool).
    Round k of an env plays episode k of every lane (global game id lane_offset + lane + k * lane_stride): new dice,
    new opening roll and new exploration draws for every game, as play_game rolls fresh dice for every game
    (train.py:64-121).  episode=None continues the env's own round counter; an explicit value replays that round."""
    if episode is None:
        episode = getattr(env, "_round", 0)
    env._round = int(episode) + 1
    traj = env.record_trajectory(max_plies)
    env.reset(episode=int(episode))
    done_steps = 0
    while done_steps < max_plies:                      # 16 turns per call; finished games are frozen and skipped
        k = min(16, max_plies - done_steps)
        env.run_greedy(k, auto_reset=False, epsilon=epsilon, precision=precision)
        done_steps += k
        if bool(((env.flags() & 4) != 0).all()):
            break
    flags = env.flags()
    done = (flags & 4) != 0
    ply, _ = env.progress()
    lengths = torch.where(done, ply + 1, torch.zeros_like(ply))        # unfinished games are not replayed
    lengths = torch.clamp(lengths, max=max_plies)
    p1_won = ((flags >> 1) & 1) == 0
    T = int(lengths.max().item()) if lengths.numel() else 0
    env.record_trajectory(None)
    return traj[:max(T, 1)], lengths, p1_won


class ContinuousSelfPlay:
    """Self-play without the tail: every lane starts its next game (the lane's next episode: fresh dice, fresh opening roll) the step
    after its last one ended, so all n lanes are busy at every step, and the turns go to a RING log indexed by the env step
    (VecGame.record_ring, bgamd_env_set_trajectory_ring).  A round of one game per lane (play_round: train.py:527-547 as it stands) spends
    most of its ~460 steps on a few long games; here 65 536 lanes finish ~65 536 games every ~84 steps.  Every game is still ONE episode
    of ONE lane -- with fixed weights the same game play_round(episode=k) plays (tests/test_gpu_round4.py) -- but a caller that refreshes
    the weights between windows lets the games in flight go on under the new ones: the documented deviation of this mode (TD-Gammon's own
    self-play changes the weights after every move).

        sp = ContinuousSelfPlay(env, ring_steps=1024)
        sp.play(84, epsilon=0.1)                      # 84 env steps, all lanes
        table = sp.finished()                          # the games that ended in those steps
        learner.replay_games(sp.rows, *table, slots=2048)
    """

    def __init__(self, env, ring_steps: int = 1024, episode: int = 0):
        self.env = env
        self.R = int(ring_steps)
        self.rows, self.end = env.record_ring(self.R)
        env.reset(episode=int(episode))
        self._collected = 0                        # env steps whose end records finished() has turned into games

    def play(self, n_steps: int, epsilon: float = 0.0, precision=0):
        if self.env.trajectory_step() + int(n_steps) - self._collected > self.R:
            raise ValueError("the window is longer than the ring: collect finished() first")
        self.env.run_greedy(int(n_steps), auto_reset=True, epsilon=epsilon, precision=precision)

    def finished(self, keep_margin: int = 0):
        """The games that ended in the env steps played since the last call -> (game_lane, game_start, lengths, p1_won) device tensors,
        in (end step, lane) order.  A game whose first turns the ring has already overwritten -- longer than ring_steps - window -
        keep_margin (keep_margin: env steps that will be PLAYED while this table is replayed, i.e. the next window of a pipelined
        loop) -- is dropped and counted in .dropped.  Call it on the stream play() was issued on: trajectory_step() counts ENQUEUED env
        steps, and the end records are read by work queued behind them on the current stream."""
        s0, s1 = self._collected, self.env.trajectory_step()
        self._collected = s1
        if s1 - s0 > self.R:
            raise ValueError("more steps played than the ring holds")
        steps = torch.arange(s0, s1, device=self.rows.device)
        rec = self.end[steps % self.R]                               # [window, n] int16
        k, lane = torch.nonzero(rec, as_tuple=True)
        val = rec[k, lane].to(torch.int32)
        length = val & 0x7FFF
        won = val >= 0                                               # bit 15 = winner (1 = PLAYER2): a negative int16
        first = s0 + k.to(torch.int64) - length.to(torch.int64) + 1  # env step of the game's first turn
        ok = (first >= 0) & (first > s1 + int(keep_margin) - 1 - self.R)
        self.dropped = int((~ok).sum().item())
        lane, length, won, first = lane[ok], length[ok], won[ok], first[ok]
        return lane.to(torch.int32), (first % self.R).to(torch.int32), length.to(torch.int32), won

    def close(self):
        self.env.record_ring(None)
