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

import torch
import torch.distributed as dist


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
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
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
            x = X[t]
            v, h = self.values(x)
            if t + 1 < T:
                v_next, _ = self.values(X[t + 1])
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

    def state_dict(self):
        W1, b1, W2, b2 = self._split(self.theta.detach().cpu())
        return {"fc1.weight": W1.clone(), "fc1.bias": b1.clone(), "fc2.weight": W2.reshape(1, 128).clone(),
                "fc2.bias": b2.clone()}


def play_round(env, max_plies: int = 512, epsilon: float = 0.0, precision=0):
    """One round of self-play from a frozen weight snapshot (train.py:527-547 semantics): every lane plays
    ONE game to the end, turns are logged. -> (rows [T, n, 8] int32, lengths [n], p1_won [n] bool)."""
    traj = env.record_trajectory(max_plies)
    env.reset()
    for _ in range(max_plies):
        env.step_greedy(auto_reset=False, epsilon=epsilon, precision=precision)
        if _ % 16 == 15 and bool(((env.flags() & 4) != 0).all()):
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
