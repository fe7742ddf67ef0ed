#!/usr/bin/env python3
"""Per-kernel average durations of the training loop's two halves ALONE (sequential windows) and BESIDE each other (pipelined windows), from two
`rocprofv3 --kernel-trace --stats` runs of tools/train_pipeline.py (--modes cont / --modes cont_pipe): VERDICT r3 item 1's evidence.
    python tools/pipeline_slowdown.py gpurun_out/<tag>/pipe_cont gpurun_out/<tag>/pipe_cont_pipe"""
import csv
import glob
import sys


def load(d):
    f = glob.glob(d + "/*/*kernel_stats.csv")[0]
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f))}


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("bg::", "").replace("void ", "").split("(")[0][:46]


a, b = load(sys.argv[1]), load(sys.argv[2])
print("%-46s %8s %10s   %8s %10s   %s" % ("kernel", "calls", "alone us", "calls", "beside us", "slowdown"))
for n, (ca, ta) in sorted(a.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:11]:
    if n in b and "elementwise" not in n:
        cb, tb = b[n]
        print("%-46s %8d %10.1f   %8d %10.1f   %+.1f %%" % (short(n), ca, ta, cb, tb, 100 * (tb / ta - 1)))
