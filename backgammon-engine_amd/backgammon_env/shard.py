"""Multi-GPU layout of the env step: games are independent, so the lanes are
split across ranks with NO data-path collective.  Rank r of R owns global lanes
[r*B, (r+1)*B); Philox streams are keyed by the global game id, so every game plays the same
moves whatever R is.  The only cross-rank traffic is the final reduction of counters/timing
(and, in training, one all-reduce of the 25 601-float update per step)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_for_rank(rank: int, world: int, games_per_rank: int):
    """-> (lane_offset, lane_stride) for bgamd_env_create."""
    if not (0 <= rank < world) or games_per_rank <= 0:
        raise ValueError("bad shard spec")
    return rank * games_per_rank, world * games_per_rank


def global_game_id(rank: int, world: int, games_per_rank: int, lane: int, episode: int) -> int:
    off, stride = shard_for_rank(rank, world, games_per_rank)
    return off + lane + episode * stride


def aggregate(counters: dict, elapsed_s: float, device=None):
    """Sum the integer counters over ranks and take the MAX of the elapsed time (bench contract).
    Works on any initialised process group (nccl on GPUs, gloo on CPU); identity when not distributed."""
    if not (dist.is_available() and dist.is_initialized()):
        return dict(counters), float(elapsed_s)
    keys = sorted(counters)
    dev = device if device is not None else torch.device("cpu")
    c = torch.tensor([int(counters[k]) for k in keys], dtype=torch.int64, device=dev)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=dev)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return {k: int(v) for k, v in zip(keys, c.tolist())}, float(t.item())


def census(elapsed_s: float, device=None):
    """-> (ranks_seen, [elapsed_s of every rank, by rank]): an all-reduced count of ones and an all-gather of the ranks' own timed
    regions over the bench's process group -- what the line reports as `ranks_seen` / `per_rank_ms_per_step`, so a launch that
    silently ran fewer ranks than asked for (or one straggling GPU) shows in the record.  (1, [elapsed_s]) when not distributed."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1, [float(elapsed_s)]
    dev = device if device is not None else torch.device("cpu")
    one = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(every, t)
    return int(one.item()), [float(x.item()) for x in every]
