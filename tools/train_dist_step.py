#!/usr/bin/env python3
"""What the per-step collective costs the training loop, measured on ONE rank with the RCCL communicator present (VERDICT r2
item 6; 8-GPU runs are the driver's).  One self-play round, then its streamed TD(lambda) replay through k slots on three routes:
  local   bgamd_td_replay: every step's launches issued by the library, the update applied by the reduce kernel
  split   the distributed route without a collective: td_step (update handed out) -> td_apply, issued step by step from Python
  nccl    the same with dist.all_reduce(update) on a world of one rank (backend nccl = RCCL) between the two
  lib     round 4: the collective issued by the LIBRARY (bgamd_td_replay_allreduce: step kernels -> ncclAllReduce in place -> apply kernel,
          three enqueues from C on the learner's stream; an RCCL communicator of the learner's own, world 1)
    python tools/train_dist_step.py [games] [slots ...]      (rank 0 of a world of 1; GPU_MAX_HW_QUEUES=8 as in bench.py)"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
import sys, time
import numpy as np, torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "backgammon-engine_amd"))
import backgammon_env as bg
from backgammon_env.learner import DeviceTDLambdaLearner, play_round, stream_schedule

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
slots = [int(x) for x in sys.argv[2:]] or [256, 1024, 2048]
w = np.fromfile(os.path.join(ROOT, "tests/golden/tdgammonNEW100k.f32"), dtype=np.float32)
env = bg.VecGame(n, seed=5); env.load_weights(w)
rows, lengths, won = play_round(env, max_plies=600, epsilon=0.05)
turns = int(lengths.sum().item())
L = DeviceTDLambdaLearner(w, max_games=n, alpha=0.1, lam=0.7)


def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, time.perf_counter() - t0


Lc = None                                                      # the learner with the library's own communicator (after init_process_group)


def run(route, k):
    kw = dict(batch_scale=min(1.0, 24.0 / k), slots=k)
    if route not in ("local", "lib"):
        kw["split_apply"] = True
    if route in ("nccl", "lib"):
        kw["group"] = dist.group.WORLD
    lrn = Lc if route == "lib" else L
    best = None
    for _ in range(3):
        lrn.set_weights(w)
        (_, cnt), dt = timed(lambda: lrn.replay_rows(rows, lengths, won, **kw))
        best = dt if best is None else min(best, dt)
    return best, cnt, lrn.theta.clone()


print(f"{n} games, {turns} turns; one rank, world size 1", flush=True)
res = {}
for k in slots:
    _, _, n_steps, kk = stream_schedule(lengths.to(torch.int32), k)
    res[k] = {"steps": n_steps}
    for route in ("local", "split"):
        dt, cnt, _ = run(route, k)
        res[k][route] = dt
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.zeros(25601, device="cuda")
for _ in range(20): dist.all_reduce(x)
(_, dt_ar) = timed(lambda: [dist.all_reduce(x) for _ in range(1000)])
print(f"dist.all_reduce(25 601 floats), world 1, 1000 calls back to back: {1e3 * dt_ar:.2f} ms = {1e3 * dt_ar:.2f} us per call (host + device; "
      f"RCCL has nothing to move on one rank: what is left is the cost of the CALL)", flush=True)
Lc = DeviceTDLambdaLearner(w, max_games=n, alpha=0.1, lam=0.7)
Lc.init_collective(dist.group.WORLD)
for k in slots:
    th = {}
    for route in ("local", "split", "nccl", "lib"):          # local / split again: the communicator's streams are in the process now
        if route in ("nccl", "lib"):
            os.environ["BGAMD_FORCE_COLLECTIVE"] = "1"       # (a group of one rank would otherwise skip the collective)
        else:
            os.environ.pop("BGAMD_FORCE_COLLECTIVE", None)
        dt, cnt, th[route] = run(route, k)
        res[k][route + "_with_comm"] = dt
    assert torch.equal(th["local"], th["lib"]) and torch.equal(th["split"], th["nccl"]), "the routes must leave identical weights"
for k in slots:
    r = res[k]
    us = lambda t: 1e6 * t / r["steps"]
    print(f"{k:5d} slots, {r['steps']:5d} steps: local {us(r['local']):6.1f} us/step | split {us(r['split']):6.1f} | with the RCCL communicator in the process: "
          f"local {us(r['local_with_comm']):6.1f}  split {us(r['split_with_comm']):6.1f}  split + torch all_reduce {us(r['nccl_with_comm']):6.1f} us/step "
          f"(+{us(r['nccl_with_comm']) - us(r['split_with_comm']):.1f} for the collective)  IN-LIBRARY step + ncclAllReduce + apply {us(r['lib_with_comm']):6.1f} us/step "
          f"(+{us(r['lib_with_comm']) - us(r['local_with_comm']):.1f} over the local route, +{us(r['lib_with_comm']) - us(r['split_with_comm']):.1f} over the split route)", flush=True)
dist.destroy_process_group()
