"""Experiment: 65 536 lanes as ONE env vs TWO envs of 32 768 lanes stepping concurrently on two streams (do the
kernels of one half fill the ramp-up / drain bubbles of the other's?).  python tools/two_halves.py"""
import os, sys, time, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "backgammon-engine_amd")]
import backgammon_env as bg
w = np.fromfile(os.path.join(ROOT, "tests/golden/tdgammonNEW100k.f32"), dtype=np.float32)
K = 400


def make(n, seed):
    e = bg.VecGame(n, seed=seed); e.load_weights(w); e.run_greedy(160); return e


def timed(fn):
    torch.cuda.synchronize(); t0 = time.time(); fn(); torch.cuda.synchronize(); return time.time() - t0


one = make(65536, 1)
timed(lambda: one.run_greedy(50))
t1 = timed(lambda: one.run_greedy(K))
print("one env of 65536: %.4f ms/step, %.1f M steps/s" % (1e3 * t1 / K, 65536 * K / t1 / 1e6), flush=True)

a, b = make(32768, 2), make(32768, 3)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def worker(env, s):
    with torch.cuda.stream(s):
        env.run_greedy(K)


def both():
    ta = threading.Thread(target=worker, args=(a, sa)); tb = threading.Thread(target=worker, args=(b, sb))
    ta.start(); tb.start(); ta.join(); tb.join()


timed(both)
t2 = timed(both)
print("two envs of 32768 on two streams (two host threads): %.4f ms per pair of steps, %.1f M steps/s" % (1e3 * t2 / K, 65536 * K / t2 / 1e6), flush=True)
t3 = timed(lambda: a.run_greedy(K))
print("one env of 32768 alone: %.4f ms/step, %.1f M steps/s" % (1e3 * t3 / K, 32768 * K / t3 / 1e6), flush=True)
