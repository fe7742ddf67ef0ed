#!/usr/bin/env python3
"""A/B of BGAMD_EXPLORE_FORK on one box: epsilon-greedy self-play steps (the training loop's play) with the exploring lanes' task kernels on the caller's
stream (0) and on the env's side stream beside the expansion and the value net (1), interleaved rounds.   python tools/ab_explore_fork.py [--lanes 65536,32768]"""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "backgammon-engine_amd")]
import numpy as np, torch
import backgammon_env as bg
ap = argparse.ArgumentParser()
ap.add_argument("--lanes", default="65536,32768")
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--eps", default="0.0,0.05,0.1")
a = ap.parse_args()
w = np.fromfile(os.path.join(ROOT, "tests/golden/tdgammonNEW100k.f32"), dtype=np.float32)
for n in [int(x) for x in a.lanes.split(",")]:
    envs = {}
    for f in ("0", "1"):
        os.environ["BGAMD_EXPLORE_FORK"] = f
        envs[f] = bg.VecGame(n, seed=3, arena_rows=n * 512)
        envs[f].load_weights(w)
        envs[f].run_greedy(200, epsilon=0.05)
    for eps in [float(x) for x in a.eps.split(",")]:
        res = {"0": [], "1": []}
        for r in range(3):
            for f in ("0", "1"):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                envs[f].run_greedy(a.steps, epsilon=eps)
                torch.cuda.synchronize(); res[f].append((time.perf_counter() - t0) / a.steps * 1e6)
        print("%6d lanes, epsilon %.2f: tasks on the caller's stream %.1f us per step, on the side stream %.1f us (median of 3 x %d steps)"
              % (n, eps, sorted(res["0"])[1], sorted(res["1"])[1], a.steps), flush=True)
    assert torch.equal(envs["0"].states(), envs["1"].states())
