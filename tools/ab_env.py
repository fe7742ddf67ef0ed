#!/usr/bin/env python3
"""A/B of environment switches on ONE GPU box: runs `bench.py --quick` once per variant, interleaved `--rounds` times so that
clock / thermal drift hits every variant alike, and prints per variant the median ms per step and the per-kernel times.
    python tools/ab_env.py "BGAMD_EXPAND_MERGED=0" "BGAMD_EXPAND_MERGED=1" "BGAMD_EXPAND_MERGED=1 BGAMD_EXPAND_DBL_PCT=40" [--rounds 3] [--extra "--games 32768"]
A variant is a space-separated list of NAME=VALUE (an empty string = the defaults)."""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--extra", default="")
a = ap.parse_args()
res = {v: [] for v in a.variants}
for r in range(a.rounds):
    for v in a.variants:
        env = dict(os.environ, BGAMD_NO_BUILD="1")
        for kv in v.split():
            k, _, val = kv.partition("=")
            env[k] = val
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--quick", "--steps", str(a.steps)] + a.extra.split(),
                             env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(v, "FAILED", out.stderr[-800:], flush=True)
            continue
        d = json.loads(line[-1])
        k = d["kernels"]
        res[v].append((d["ms_per_step"], k["eval"]["avg_ms"], k["leaves"]["avg_ms"], k["expand"]["avg_ms"], k["apply_avg_ms"],
                       k["eval"].get("root_pass_avg_ms") or 0.0, d["value"] / 1e6))
        print("%-60s" % (v or "(defaults)"), r, res[v][-1], flush=True)
print("%-60s %9s %9s %9s %9s %9s %9s %9s" % ("variant", "ms/step", "eval", "leaves", "doubles", "boundary", "root", "M steps/s"))
for v in a.variants:
    if res[v]:
        print("%-60s " % (v or "(defaults)") + " ".join("%9.4f" % statistics.median(x[i] for x in res[v]) for i in range(7)))
