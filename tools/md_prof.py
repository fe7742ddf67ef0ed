#!/usr/bin/env python3
"""Phase profile of eval_rows_mdelta_kernel from diagnostic builds (tools/ab_build.sh p1 "-DBG_MD_PROF=1" ... p7 "-DBG_MD_PROF=7"):
runs 65 536 lanes of steady-state greedy play per variant and prints the mean shader cycles a wave spends in the phase per launch.
    python tools/md_prof.py p1 p2 p3 p4 p5 p6 p7"""
import os, subprocess, sys, json
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CHILD = r'''
import os, sys, numpy as np
sys.path.insert(0, os.path.join(%r, "backgammon-engine_amd"))
import backgammon_env as bg
w = np.fromfile(os.path.join(%r, "tests", "golden", "tdgammonNEW100k.f32"), dtype=np.float32)
env = bg.VecGame(65536, seed=20240603)
env.load_weights(w)
env.run_greedy(160)
env.reset_stats()
K = 40
env.run_greedy(K)
st = env.stats()
print("RESULT", st["ksteps_executed"] / K, st["rows_evaluated"] / K)
''' % (ROOT, ROOT)
names = sys.argv[1:]
for n in names:
    lib = os.path.join(ROOT, "backgammon-engine_amd", "variants", "libbgamd_%s.so" % n)
    out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, BGAMD_LIB=lib, BGAMD_MFMA_DELTA="1"), capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        print(n, "FAILED", out.stderr[-600:]); continue
    cyc, rows = [float(x) for x in line[0].split()[1:]]
    waves = 256 * 12
    print("%-6s %12.0f cycles per wave per launch  (%.1f us at 2.4 GHz; %.1f cycles per 64-row chunk)" % (n, cyc / waves, cyc / waves / 2400, cyc / (rows / 64)))
