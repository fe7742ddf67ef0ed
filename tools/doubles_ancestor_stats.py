"""VERDICT r4 item 3, the offline measurement: what diffing a DOUBLES row against the ply-2 node it descends from (instead of the root)
would buy the incremental value net.  Per 64-row tile the kernel runs as many gather passes as the LONGEST (feature, delta) list of the
tile; doubles rows (arenas 2-3: <= 4 moves of one die) carry lists of 8-13 entries against the root.  For one greedy step of a 65 536-lane
env at three game phases, per arena (2 * doubles + hit, each tiled by itself as the kernel tiles it): rows, tiles, the mean list and the
mean longest list per tile (a) against the root -- what the kernel does today -- and (b), for the doubles arenas, against the position after
the row's first two moves; plus the ply-2 nodes such a scheme would have to evaluate first (a_node = a_root + sum delta W1: no sigmoids).
The predicted saving uses the two-point model of profiles/r04_ab_expand_merged.txt (each kind of turn evaluated by itself: doubles 41.5 us,
the others 53.0 us): t = tiles x (c0 + c1 x passes).   -> profiles/r05_doubles_ancestor_stats.txt"""
import sys
import numpy as np

sys.path.insert(0, "backgammon-engine_amd")
import backgammon_env as bg  # noqa: E402

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


def apply_moves(s, mover, die, origins, nmov):
    """s int [N, 28] (a copy is changed), mover [N] 0 = PLAYER1 (moves up, bar = position 0, off = 25), die [N], origins [N, k] absolute
    positions 0..25, nmov [N] how many of the k moves to apply.  game.cpp:573-663 restated on arrays (legal moves only)."""
    s = s.copy()
    idx = np.arange(len(s))
    for j in range(origins.shape[1]):
        act = nmov > j
        o = origins[:, j]
        for pl in (0, 1):
            m = act & (mover == pl)
            if not m.any():
                continue
            r, oo, d = idx[m], o[m], die[m]
            sgn = 1 if pl == 0 else -1
            on_bar = oo == (0 if pl == 0 else 25)
            s[r[on_bar], 24 + pl] -= 1
            rb = r[~on_bar]
            s[rb, oo[~on_bar] - 1] -= sgn
            dest = np.clip(oo + sgn * d, 0, 25)
            off = dest == (25 if pl == 0 else 0)
            s[r[off], 26 + pl] += 1
            rl, dl = r[~off], dest[~off]
            hit = s[rl, dl - 1] == -sgn
            s[rl[hit], dl[hit] - 1] = 0
            s[rl[hit], 24 + (1 - pl)] += 1
            s[rl, dl - 1] += sgn
    return s


def tiles(c, wd=64):
    if len(c) == 0:
        return 0, 0.0
    pad = (-len(c)) % wd
    cc = np.concatenate([c, np.zeros(pad, dtype=c.dtype)]).reshape(-1, wd)
    return cc.shape[0], float(cc.max(1).mean())


tot = {"tiles_o": 0, "pass_o": 0.0, "tiles_d": 0, "pass_d_root": 0.0, "pass_d_anc": 0.0, "anc_nodes": 0, "rows": 0}
for warm in (6, 30, 60):
    env.reset()
    env.run_greedy(warm)
    s0, t0 = env.states().cpu().numpy(), env.turns().cpu().numpy()
    env.step_greedy()
    dice = env.dice().cpu().numpy()
    info, st, val = env.unique_rows()
    info, st = info.cpu().numpy(), st.cpu().numpy()
    g, key = info[:, 0], info[:, 1] & 0x7FFFFFFF
    mover = t0[g]
    dbl = dice[g, 0] == dice[g, 1]
    cnt_root = (feats(st) != feats(s0[g])).sum(1)
    bar0 = np.where(mover == 0, s0[g, 25], s0[g, 24])
    bar1 = np.where(mover == 0, st[:, 25], st[:, 24])
    hit = (bar1 - bar0) > 0
    klen = key & 7
    org = np.stack([(key >> (18 - 5 * i)) & 31 for i in range(4)], axis=1)
    # check of the restatement: all the row's moves applied to the root give the row
    full = apply_moves(s0[g], mover, dice[g, 0], org, np.where(dbl, klen, 0))
    assert (full[dbl] == st[dbl]).all(), "move replay disagrees with the library's rows"
    anc = apply_moves(s0[g], mover, dice[g, 0], org[:, :2], np.where(dbl, np.minimum(klen, 2), 0))
    cnt_anc = (feats(st) != feats(anc)).sum(1)
    # the ply-2 nodes to evaluate first: distinct (game, o0, o1) among the doubles rows of >= 2 moves, and their own lists against the root
    d2 = dbl & (klen >= 2)
    nodes, first = np.unique(np.stack([g[d2], org[d2, 0], org[d2, 1]], axis=1), axis=0, return_index=True)
    cnt_node = (feats(anc[d2][first]) != feats(s0[g[d2]][first])).sum(1)
    res = {"warm": warm, "rows": len(g), "mean_list": round(float(cnt_root.mean()), 3)}
    for a in range(4):
        m = (dbl == bool(a >> 1)) & (hit == bool(a & 1))
        nt, mx = tiles(cnt_root[m])
        res["arena%d" % a] = {"rows": int(m.sum()), "tiles": nt, "mean_list": round(float(cnt_root[m].mean()), 2) if m.any() else 0.0, "longest_per_tile": round(mx, 2)}
        if a >= 2:
            nt2, mx2 = tiles(cnt_anc[m])
            res["arena%d" % a].update(mean_list_vs_ply2=round(float(cnt_anc[m].mean()), 2), longest_per_tile_vs_ply2=round(mx2, 2))
            tot["tiles_d"] += nt; tot["pass_d_root"] += nt * mx; tot["pass_d_anc"] += nt * mx2
        else:
            tot["tiles_o"] += nt; tot["pass_o"] += nt * mx
    ntn, mxn = tiles(cnt_node)
    res["ply2_nodes"] = {"nodes": int(len(nodes)), "tiles": ntn, "mean_list_vs_root": round(float(cnt_node.mean()), 2), "longest_per_tile": round(mxn, 2),
                         "hidden_MB_written_and_read": round(2 * len(nodes) * 512 / 1e6, 1)}
    tot["anc_nodes"] += len(nodes); tot["rows"] += len(g)
    print(res, flush=True)

# two-point model: doubles alone 41.5 us, others alone 53.0 us (r04 A/B, a build whose value net took 80.1 us together; HEAD: 72.0)
Nd, No = tot["tiles_d"] / 3, tot["tiles_o"] / 3
Pd, Po, Pa = tot["pass_d_root"] / tot["tiles_d"], tot["pass_o"] / tot["tiles_o"], tot["pass_d_anc"] / tot["tiles_d"]
# t_d = Nd (c0 + c1 Pd) = 41.5, t_o = No (c0 + c1 Po) = 53.0
A = np.array([[Nd, Nd * Pd], [No, No * Po]])
c0, c1 = np.linalg.solve(A, np.array([41.5, 53.0]))
t_new = Nd * (c0 + c1 * Pa)
scale = 72.0 / 80.1
extra_nodes = (tot["anc_nodes"] / 3) / 64 * (0.35 * c0 + c1 * 4.0)          # the ancestor pass: list + gathers, no sigmoids (~35 % of a tile's fixed cost)
print("per step: %.0f doubles tiles at %.2f passes (vs the ply-2 node: %.2f), %.0f other tiles at %.2f passes" % (Nd, Pd, Pa, No, Po))
print("model t = tiles x (c0 + c1 x passes): c0 = %.2f ns, c1 = %.2f ns per tile and pass (256 CUs x 16 waves side by side)" % (1e3 * c0, 1e3 * c1))
print("doubles tiles alone: 41.5 -> %.1f us; scaled to HEAD's value net (72.0 of 80.1): saving %.1f us; the ancestor pass (%.0f nodes) costs ~%.1f us, "
      "its hidden vectors %.1f MB through HBM/L2 -> predicted net saving %.1f us of 72 (build only from 8)"
      % (t_new, scale * (41.5 - t_new), tot["anc_nodes"] / 3, scale * extra_nodes, 2 * tot["anc_nodes"] / 3 * 512 / 1e6,
         scale * (41.5 - t_new) - scale * extra_nodes))
