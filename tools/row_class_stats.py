"""What ordering the leaf stage's rows by a cheap class buys the value net (round 4).  The incremental kernel runs, per 64-row tile, as many gather
passes as the LONGEST delta list in the tile.  For one greedy step of a 65 536-lane env at three game phases: the mean list length, the mean over
tiles of the longest list in arena order (as the library writes the rows: two runs per workgroup phase, hits last), and what a stable sort inside
windows of W consecutive rows (a workgroup phase is ~900 rows) by (a) hit / no hit, (b) 0 / 1 / >= 2 hits, (c) hits and "the same checker moved on",
(d) the exact list length would need.  -> profiles/r04_row_class_stats.txt"""
import sys, numpy as np, torch
sys.path.insert(0, "backgammon-engine_amd")
import backgammon_env as bg
w = np.fromfile("tests/golden/tdgammonNEW100k.f32", dtype=np.float32)
n = 65536
env = bg.VecGame(n, device=0, seed=20240603)
env.load_weights(w)
def feats(s):
    b = s[:, :24]
    out = []
    for side in (1, -1):
        c = np.clip(b * side, 0, None)
        out += [(c >= 1), (c >= 2), (c >= 3), np.clip(c - 3, 0, None)]
    f = np.concatenate([x.astype(np.int16) for x in out], axis=1)
    return np.concatenate([f, s[:, 24:28].astype(np.int16)], axis=1)
def tilemax(c, wd=64):
    m = len(c) // wd * wd
    return float(c[:m].reshape(-1, wd).max(1).mean())
for warm in (6, 30, 60):
    env.reset(); env.run_greedy(warm)
    s0 = env.states().cpu().numpy(); t0 = env.turns().cpu().numpy()
    env.step_greedy()
    info, st, val = env.unique_rows()
    g = info[:, 0].cpu().numpy(); st = st.cpu().numpy()
    cnt = (feats(st) != feats(s0[g])).sum(1)
    mover = t0[g]                                      # 0: PLAYER1 moved -> the opponent's bar is state[25]; 1: state[24]
    bar0 = np.where(mover == 0, s0[g, 25], s0[g, 24]); bar1 = np.where(mover == 0, st[:, 25], st[:, 24])
    hits = (bar1 - bar0).astype(np.int64)
    res = {"warm": warm, "rows": len(cnt), "mean": round(float(cnt.mean()), 3), "arena_order": round(tilemax(cnt), 3)}
    for W in (512, 1024, 2048, 1 << 30):
        m = len(cnt) if W > len(cnt) else len(cnt) // W * W
        Wn = m if W > len(cnt) else W
        c2, h2 = cnt[:m].reshape(-1, Wn), hits[:m].reshape(-1, Wn)
        for name, key in (("hit/no", np.minimum(h2, 1)), ("0/1/2+", np.minimum(h2, 2)), ("exact", c2)):
            order = np.argsort(key, axis=1, kind="stable")
            res["W%s %s" % ("all" if W > len(cnt) else W, name)] = round(tilemax(np.take_along_axis(c2, order, axis=1).reshape(-1)), 3)
    print(res, flush=True)
