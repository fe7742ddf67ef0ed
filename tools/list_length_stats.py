"""How much of the value net's gather work is the wave-wide maximum?  Per 64-row tile the incremental kernel runs as many gather passes as its
LONGEST (feature, delta) list.  Prints, at three game phases of a 65 536-lane env: the mean list length, the mean over tiles of the longest list (arena order),
and what rows sorted by length inside windows of 256 .. 16 384 rows (or globally) would need.  -> profiles/r02_list_length_stats.txt."""
import sys, numpy as np, torch
sys.path.insert(0, "backgammon-engine_amd")
import backgammon_env as bg
w = np.fromfile("tests/golden/tdgammonNEW100k.f32", dtype=np.float32)
n = 65536
env = bg.VecGame(n, device=0, seed=20240603)
env.load_weights(w)
def feats(s):                                  # s int [N,28] -> [N,196] integer feature levels
    b = s[:, :24]
    out = []
    for side in (1, -1):
        c = np.clip(b * side, 0, None)
        out += [(c >= 1), (c >= 2), (c >= 3), np.clip(c - 3, 0, None)]
    f = np.concatenate([x.astype(np.int16) for x in out], axis=1)
    return np.concatenate([f, s[:, 24:28].astype(np.int16)], axis=1)
for warm in (6, 30, 60):
    env.reset(); env.run_greedy(warm)
    s0 = env.states().cpu().numpy()
    env.step_greedy()
    info, st, val = env.unique_rows()
    g = info[:, 0].cpu().numpy(); st = st.cpu().numpy()
    cnt = (feats(st) != feats(s0[g])).sum(1)
    U = len(cnt)
    def tilemax(c, w=64):
        m = len(c) // w * w
        return c[:m].reshape(-1, w).max(1).mean()
    res = {"warm": warm, "U": U, "mean": cnt.mean(), "tile64max": tilemax(cnt), "sorted_global": tilemax(np.sort(cnt))}
    for win in (256, 1024, 4096, 16384):
        m = U // win * win
        res["sorted_%d" % win] = tilemax(np.sort(cnt[:m].reshape(-1, win), axis=1).reshape(-1))
    # strided blocks: block b of G gets tiles b, b+G, ...  (G = 256): sort within the block's 16 concurrent tiles
    res["hist"] = np.bincount(cnt, minlength=17).tolist()
    print(res, flush=True)
