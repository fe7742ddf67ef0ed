import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "backgammon-engine_amd"))
import backgammon_env as bg
from backgammon_env.learner import DeviceTDLambdaLearner, TDLambdaLearner, play_round
w = np.fromfile(os.path.join(ROOT, "tests/golden/tdgammonNEW100k.f32"), dtype=np.float32)
for n in [int(x) for x in (sys.argv[1:] or ["512", "4096", "16384"])]:
    env = bg.VecGame(n, seed=5); env.load_weights(w)
    rows, lengths, won = play_round(env, max_plies=512, epsilon=0.05)
    tot = int(lengths.sum().item())
    L = DeviceTDLambdaLearner(w, max_games=n, alpha=0.1, lam=0.9)
    L.replay_rows(rows, lengths, won, batch_scale=24.0 / n)
    L.set_weights(w)
    L.time_trace_kernel(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sq, cnt = L.replay_rows(rows, lengths, won, batch_scale=24.0 / n)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, nl, gs = L.trace_kernel_times()
    cols, wcols = L.active_columns(), L.written_columns()
    moved = ((cols + wcols) * 128 + 2 * gs * 257) * 4   # bytes read + written: active / written W1 columns + the dense b1 | W2 | b2 tail
    print(f"n={n} T={rows.shape[0]} updates={cnt} device replay {dt*1e3:.1f} ms -> {cnt/dt/1e6:.2f} M updates/s | trace kernel {ms:.1f} ms "
          f"({nl} launches): {cols/max(gs,1):.1f} of 198 W1 columns read, {wcols/max(gs,1):.1f} written per update, {moved/ms/1e6:.0f} GB/s moved "
          f"(dense-equivalent {gs*204808/ms/1e6:.0f} GB/s); sum|theta| {float(L.theta.double().abs().sum()):.9g}", flush=True)
    if n <= 4096:
        X = env.encode_rows(rows)
        Lt = TDLambdaLearner(w, device="cuda", alpha=0.1, lam=0.9)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Lt.replay(X, lengths, won, batch_scale=24.0 / n)
        torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
        print(f"   torch replay {dt2*1e3:.1f} ms -> {tot/dt2/1e6:.2f} M updates/s; max|dtheta| vs device {float((Lt.theta - L.theta).abs().max()):.3g}", flush=True)
    del L, env
