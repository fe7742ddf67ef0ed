"""Where a training round's time goes: one 65 536-game round of self-play with the turn log, then its TD(lambda) replay whole and in
sub-rounds or streamed through k slots.  python tools/train_breakdown.py [games] [sub_round | s<slots> ...]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "backgammon-engine_amd"))
import backgammon_env as bg
from backgammon_env.learner import DeviceTDLambdaLearner, play_round
w = np.fromfile(os.path.join(ROOT, "tests/golden/tdgammonNEW100k.f32"), dtype=np.float32)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
SPLIT = os.environ.get("TB_SPLIT_APPLY") == "1"      # the distributed route on one rank: step / (all-reduce) / apply from a Python loop
subs = sys.argv[2:] or ["0", "4096", "2048", "s4096", "s2048"]
env = bg.VecGame(n, seed=5); env.load_weights(w)
L = DeviceTDLambdaLearner(w, max_games=n, alpha=0.1, lam=0.7)
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, time.perf_counter() - t0
for rep in range(2):
    (rows, lengths, won), dt_play = timed(lambda: play_round(env, max_plies=600, epsilon=0.05))
turns = int(lengths.sum().item())
print(f"{n} games, {turns} turns, log {rows.shape[0]} steps: self-play with the turn log {dt_play*1e3:.1f} ms ({turns/dt_play/1e6:.1f} M turns/s)", flush=True)
for arg in subs:
    stream = arg.startswith("s")
    sub = int(arg[1:]) if stream else int(arg)
    k = sub if sub else n
    for rep in range(2):
        L.set_weights(w)
        (sq, cnt), dt = timed(lambda: L.replay_rows(rows, lengths, won, batch_scale=min(1.0, 24.0 / k), sub_round=0 if stream else sub,
                                                    slots=sub if stream else 0, split_apply=SPLIT))
    if stream:
        from backgammon_env.learner import stream_schedule
        _, _, n_steps, kk = stream_schedule(lengths.to(torch.int32), sub)
        print(f"    ({n_steps} steps x {kk} slots: {turns / (n_steps * kk):.3f} of the slot-steps busy)", end="")
    print(f"  replay {'streamed through slots' if stream else 'in sub-rounds'} of {k}: {dt*1e3:.1f} ms ({cnt/dt/1e6:.1f} M updates/s) -> round {1e3*(dt+dt_play):.1f} ms = "
          f"{turns/(dt+dt_play)/1e6:.1f} M turns/s end to end", flush=True)
