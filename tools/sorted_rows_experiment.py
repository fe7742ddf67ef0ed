"""Upper bound of what binning rows by list length could buy the value-net kernel: the afterstates of one 65 536-lane step through
bgamd_evaluate_incremental in arena order, stably sorted into length classes of two, and sorted by exact length (same values, bit for bit); run under
rocprofv3 --kernel-trace and read the eval_rows_delta_kernel durations.  python tools/sorted_rows_experiment.py arena|sorted|exact
-> profiles/r02_sorted_rows_experiment.txt."""
import sys, numpy as np, torch
sys.path.insert(0, "backgammon-engine_amd")
import backgammon_env as bg
mode = sys.argv[1]
w = np.fromfile("tests/golden/tdgammonNEW100k.f32", dtype=np.float32)
n = 65536
env = bg.VecGame(n, device=0, seed=20240603); env.load_weights(w)
env.run_greedy(300)
s0 = env.states().clone(); t0 = env.turns().clone()
env.step_greedy()
info, st, val = env.unique_rows()
g = info[:, 0]
def feats(s):
    b = s[:, :24]
    out = []
    for side in (1, -1):
        c = torch.clamp(b * side, min=0)
        out += [(c >= 1), (c >= 2), (c >= 3), torch.clamp(c - 3, min=0)]
    return torch.cat([x.to(torch.int16) for x in out] + [s[:, 24:28].to(torch.int16)], dim=1)
cnt = (feats(st) != feats(s0[g])).sum(1)
U = st.shape[0]
if mode == "sorted":
    cls = torch.clamp((cnt + 1) // 2, max=6)            # classes <=2, 3-4, 5-6, 7-8, 9-10, 11+
    perm = torch.argsort(cls, stable=True)
elif mode == "exact":
    perm = torch.argsort(cnt, stable=True)
else:
    perm = torch.arange(U, device=st.device)
st2, g2 = st[perm].contiguous(), g[perm].to(torch.int32).contiguous()
tile = cnt[perm][: U // 64 * 64].reshape(-1, 64).max(1).values.float().mean()
print(mode, "rows", U, "mean list", float(cnt.float().mean()), "mean tile max", float(tile), flush=True)
for _ in range(12):
    out = env.evaluate_incremental(s0, t0, st2, g2)
print("max |v - v_step|", float((out - val[perm]).abs().max()))
