#!/usr/bin/env python3
"""Value-net kernel on a FIXED workload: the afterstates of one steady-state 65 536-lane greedy step, captured once with the
library as built (arena order), then evaluated through bgamd_evaluate_incremental by every variant library (tools/ab_build.sh).
Ablation builds change the games that get played, so their in-bench kernel times compare different row sets; this does not.
    python tools/md_fixed_bench.py capture            # writes /tmp/md_fixed_rows.npz (on the GPU box)
    python tools/md_fixed_bench.py run base a1 a2 ... # kernel us per variant (median of `reps` launches), + VALU variant `valu`"""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
NPZ = "/tmp/md_fixed_rows.npz"
CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(%r, "backgammon-engine_amd"))
import backgammon_env as bg
w = np.fromfile(os.path.join(%r, "tests", "golden", "tdgammonNEW100k.f32"), dtype=np.float32)
mode = sys.argv[1]
env = bg.VecGame(65536, seed=20240603, arena_rows=1 << 22)
env.load_weights(w)
if mode == "capture":
    env.run_greedy(300)
    s0 = env.states().clone(); t0 = env.turns().clone()
    env.step_greedy()
    info, st, val = env.unique_rows()
    np.savez(%r, s0=s0.cpu().numpy(), t0=t0.cpu().numpy(), st=st.cpu().numpy(), g=info[:, 0].cpu().numpy().astype(np.int32), val=val.cpu().numpy())
    print("RESULT captured", st.shape[0], "rows")
else:
    d = np.load(%r)
    s0, t0, st, g = [torch.from_numpy(d[k]).cuda() for k in ("s0", "t0", "st", "g")]
    for _ in range(3): out = env.evaluate_incremental(s0, t0, st, g)
    env.time_kernels(True, groups=["eval"])
    reps = 25
    for _ in range(reps): out = env.evaluate_incremental(s0, t0, st, g)
    t = env.kernel_times()["eval"]
    err = float((out.cpu() - torch.from_numpy(d["val"])).abs().max())
    print("RESULT %%.2f us per launch over %%d launches, %%d rows, max |v - captured| %%.3g" %% (1e3 * t["ms"] / max(t["launches"], 1), t["launches"], st.shape[0], err))
''' % (ROOT, ROOT, NPZ, NPZ)
mode = sys.argv[1]
names = sys.argv[2:] if mode == "run" else [None]
for n in names:
    env = dict(os.environ)
    env["BGAMD_MFMA_DELTA"] = "0" if n == "valu" else "1"
    if n and n not in ("built", "valu"): env["BGAMD_LIB"] = os.path.join(ROOT, "backgammon-engine_amd", "variants", "libbgamd_%s.so" % n)
    out = subprocess.run([sys.executable, "-c", CHILD, mode], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    print("%-8s" % (n or mode), line[0][7:] if line else "FAILED " + out.stderr[-500:], flush=True)
