import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "backgammon-engine_amd"))
import backgammon_env as bg
from backgammon_env.learner import DeviceTDLambdaLearner, play_round, stream_schedule
w = np.fromfile(os.path.join(ROOT, "tests/golden/tdgammonNEW100k.f32"), dtype=np.float32)
n = 8192
env = bg.VecGame(n, seed=5); env.load_weights(w)
rows, lengths, won = play_round(env, max_plies=600, epsilon=0.05)
L = DeviceTDLambdaLearner(w, max_games=n, alpha=0.1, lam=0.7)
k = 1024
_, _, n_steps, _ = stream_schedule(lengths.to(torch.int32), k)
def run(tag):
    best = 1e9; host = 0
    for _ in range(3):
        L.set_weights(w)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        L.replay_rows(rows, lengths, won, batch_scale=24.0 / k, slots=k)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        if t2 - t0 < best: best, host = t2 - t0, t1 - t0
    env.run_greedy(4); torch.cuda.synchronize(); t0 = time.perf_counter(); env.run_greedy(40); torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 40
    print(f"{tag}: replay {1e6*best/n_steps:.1f} us/step (host call returns after {1e6*host/n_steps:.1f} us/step); greedy step at {n} lanes {1e6*tg:.1f} us", flush=True)
run("before init")
mode = sys.argv[1] if len(sys.argv) > 1 else "nccl"
if mode != "none":
    dist.init_process_group(mode, rank=0, world_size=1)
    x = torch.zeros(8, device="cuda"); dist.all_reduce(x); torch.cuda.synchronize()
run("after init_process_group(%s)" % mode)
if mode != "none":
    dist.destroy_process_group()
    run("after destroy_process_group")
