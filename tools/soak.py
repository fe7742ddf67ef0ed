"""Soak: hundreds of millions of env steps through every value-net mode (epsilon-greedy every other step) and the
random policy, with checker conservation and the error flags checked after each phase."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "backgammon-engine_amd")]
import backgammon_env as bg
w = np.fromfile(os.path.join(ROOT, "tests/golden/tdgammonNEW100k.f32"), dtype=np.float32)
n = 65536
env = bg.VecGame(n, seed=777, arena_rows=n*512); env.load_weights(w)
t0 = time.time()
for prec, steps in ((bg.F32, 6000), (bg.F32_DENSE, 1000), (bg.F16X2, 2000), (bg.BF16, 1000)):
    for i in range(steps):
        env.step_greedy(precision=prec, epsilon=0.02 if i % 2 else 0.0)
    st = env.stats()
    sa = env.states().cpu().numpy()
    p1 = np.clip(sa[:, :24], 0, None).sum(1) + sa[:, 24] + sa[:, 26]
    p2 = np.clip(-sa[:, :24], 0, None).sum(1) + sa[:, 25] + sa[:, 27]
    assert (p1 == 15).all() and (p2 == 15).all() and st["error_flags"] == 0
    print(prec, st, "p1 win rate %.4f" % (st["p1_wins"] / st["games_finished"]), "%.1f s" % (time.time() - t0), flush=True)
for i in range(3000):
    env.step_random()
st = env.stats(); print("random", st, flush=True)
assert st["error_flags"] == 0
print("soak ok: %.1f M env steps in %.1f s" % (st["steps"] / 1e6, time.time() - t0))
