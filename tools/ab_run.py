#!/usr/bin/env python3
"""A/B on ONE GPU box: runs bench.py --quick once per variant library (tools/ab_build.sh), interleaved `--rounds` times so
that clock / thermal drift hits every variant alike, and prints per variant the median ms per step and per-kernel times.
    python tools/ab_run.py base v1 v2 [--rounds 3] [--steps 200] [--extra "--games 65536"]"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="+")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--extra", default="")
a = ap.parse_args()
res = {n: [] for n in a.names}
for r in range(a.rounds):
    for n in a.names:
        lib = os.path.join(ROOT, "backgammon-engine_amd", "variants", f"libbgamd_{n}.so")
        env = dict(os.environ, BGAMD_LIB=lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--quick", "--steps", str(a.steps)] + a.extra.split(),
                             env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(n, "FAILED", out.stderr[-800:], flush=True)
            continue
        d = json.loads(line[-1])
        k = d["kernels"]
        res[n].append((d["ms_per_step"], k["eval"]["avg_ms"], k["leaves"]["avg_ms"], k["expand"]["avg_ms"], k["apply_avg_ms"],
                       k["eval"].get("root_pass_avg_ms") or 0.0))
        print(n, r, res[n][-1], flush=True)
print("%-14s %9s %9s %9s %9s %9s %9s" % ("variant", "ms/step", "eval", "leaves", "doubles", "boundary", "root"))
for n in a.names:
    if res[n]:
        print("%-14s " % n + " ".join("%9.4f" % statistics.median(x[i] for x in res[n]) for i in range(6)))
