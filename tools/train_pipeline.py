#!/usr/bin/env python3
"""A training round end to end on ONE GPU, four ways (train.py:519-547 is the loop):
  round       play_round (one game per lane, to the end) THEN replay_rows(slots=k): rounds 2-3's loop, 153.7 ms per 65 536 games
  round_pipe  the same, the replay of round r on its own stream / host thread WHILE round r + 1 is played (policy one round staler)
  cont        continuous self-play (every lane restarts the step after its game ended; ring log by env step): a window of K steps of all
              lanes THEN replay_games over the games that ended in it
  cont_pipe   the same, window w played while window w - 1 is replayed
Reports, per mode, the steady-state wall time per round / window and turns replayed per second end to end (weights are refreshed from
the learner before every round / window, as the training loop does).
    python tools/train_pipeline.py [--games 65536] [--slots 2048] [--window-steps 84] [--rounds 6] [--modes round,round_pipe,cont,cont_pipe]"""
import argparse
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "backgammon-engine_amd"))
import backgammon_env as bg  # noqa: E402
from backgammon_env.learner import ContinuousSelfPlay, DeviceTDLambdaLearner, play_round  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=65536)
ap.add_argument("--slots", type=int, default=2048)
ap.add_argument("--window-steps", type=int, default=84)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--warm", type=int, default=3)
ap.add_argument("--eps", type=float, default=0.05)
ap.add_argument("--modes", default="round,round_pipe,cont,cont_pipe")
ap.add_argument("--priority", type=int, default=0, help="stream priority of the learner's stream (-1 = high)")
ap.add_argument("--delay", type=int, default=0, help="1: the learner applies every update one step late, a training step is one launch (bgamd_td_set_delay)")
ap.add_argument("--alpha-scale", type=float, default=96.0, help="games' worth of update per training step")
a = ap.parse_args()
w0 = np.fromfile(os.path.join(ROOT, "tests/golden/tdgammonNEW100k.f32"), dtype=np.float32)
n, k = a.games, a.slots
scale = min(1.0, a.alpha_scale / k)


def sync():
    torch.cuda.synchronize()


def run(mode):
    env = bg.VecGame(n, seed=5)
    L = DeviceTDLambdaLearner(w0, max_games=max(k, 1), alpha=0.1, lam=0.7)
    L.set_delay(a.delay)
    side = torch.cuda.Stream(priority=a.priority)
    pipe = mode.endswith("_pipe")
    cont = mode.startswith("cont")
    sp = ContinuousSelfPlay(env, ring_steps=1024) if cont else None
    out = {}

    def replay(item):
        torch.cuda.set_device(0)
        with torch.cuda.stream(side):
            if cont:
                out["r"] = L.replay_games(sp.rows, *item, slots=k, batch_scale=scale)
            else:
                out["r"] = L.replay_rows(*item, slots=k, batch_scale=scale)
        side.synchronize()

    def play():
        if cont:
            sp.play(a.window_steps, epsilon=a.eps)
            return sp.finished(keep_margin=a.window_steps if pipe else 0)
        return play_round(env, max_plies=600, epsilon=a.eps)

    pending, times, turns, t_play, t_replay = None, [], [], [], []
    for r in range(a.warm + a.rounds + (1 if pipe else 0)):
        sync()
        t0 = time.perf_counter()
        env.load_weights(L.theta.cpu().numpy())
        th = None
        if pipe:
            if pending is not None:
                th = threading.Thread(target=replay, args=(pending,))
                th.start()
            item = play() if r < a.warm + a.rounds else None
            if th is not None:
                th.join()
            pending = item
        else:
            item = play()
            sync()
            t1 = time.perf_counter()
            replay(item)
            sync()
            t_play.append(t1 - t0); t_replay.append(time.perf_counter() - t1)
        sync()
        dt = time.perf_counter() - t0
        if r >= a.warm + (1 if pipe else 0) and "r" in out:
            times.append(dt); turns.append(out["r"][1])
    ms = 1e3 * float(np.median(times))
    tr = float(np.mean(turns))
    line = f"{mode:11s} {n} lanes, {k} slots: {ms:7.1f} ms per {'window of %d steps' % a.window_steps if cont else 'round'} (median of {len(times)}), " \
           f"{tr / 1e6:.2f} M turns replayed each -> {tr / ms / 1e3:.1f} M turns/s end to end"
    if t_play:
        line += f"  [play {1e3 * np.median(t_play[a.warm:]):.1f} + replay {1e3 * np.median(t_replay[a.warm:]):.1f} ms]"
    if cont:
        line += f"  [{sp.dropped} games dropped in the last window]"
    print(line, flush=True)
    del L, env
    return ms, tr


res = {}
for mode in a.modes.split(","):
    res[mode] = run(mode)
if "round" in res:
    base = res["round"][1] / res["round"][0]
    for m, (ms, tr) in res.items():
        print(f"  {m:11s} {tr / ms / base:5.2f} x the turns/s of the sequential round", flush=True)
