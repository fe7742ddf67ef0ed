#!/usr/bin/env python3
"""Per-MODE kernel statistics and the step timeline at the headline size, from the kernel trace of the DEFAULT bench command
(VERDICT r4 item 6).  `rocprofv3 --kernel-trace --stats -- python3 bench.py` averages a kernel over every launch of the process:
expand_all_kernel over the steps of all four value-net modes, the value net over burn-in + timed launches.  This tool reads the same
run's per-dispatch trace (*_kernel_trace.csv) and
  * groups the dispatches into env steps (expansion -> value net -> boundary / apply) and labels every step by its value-net kernel;
  * prints calls / mean / median / min / max per (mode, kernel), and -- for the f32 steps of the TIMED REGION (steps burnin + warmup ..
    burnin + warmup + K of the process, counted from the first greedy step) -- the same plus the timeline: mean start / end of the three
    launches relative to the expansion's start, the gaps between them and the step period.
    python tools/trace_by_mode.py gpurun_out/<tag>/stats [--skip 180 --steps 200] > profiles/<tag>_kernel_trace_by_mode.txt
Runs on the GPU box right after the profile (tools/profile_round.sh): the trace itself is too large to travel."""
import argparse
import collections
import csv
import glob
import os
import statistics


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("bg::", "").replace("void ", "")
    base = n.split("(")[0]
    if base.startswith("boundary_kernel"):
        return base                                   # keep <true> / <false>
    return base.split("<")[0]


VALUE_NETS = {"eval_rows_delta_kernel": "f32", "eval_rows_f32_kernel": "f32_dense", "eval_rows_f16x2_kernel": "f16x2", "eval_rows_bf16_kernel": "bf16",
              "eval_rows_mdelta_kernel": "f32(mfma delta)", "eval_rows_d16_kernel": "f16x2(resident)"}
EXPAND = ("expand_all_kernel",)
CLOSE = ("boundary_kernel<true>", "boundary_kernel<false>", "apply_kernel")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--skip", type=int, default=180, help="f32 steps before the timed region (bench.py: --burnin + --warmup)")
    ap.add_argument("--steps", type=int, default=200, help="steps of the timed region (bench.py --steps)")
    a = ap.parse_args()
    f = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))), key=lambda x: x[0])
    # steps: an expansion launch, the next value-net launch, the next closing launch (in between: the root pass / exploration kernels of
    # other structures are ignored)
    steps, i = [], 0
    while i < len(rows):
        if rows[i][2] in EXPAND:
            j = i + 1
            while j < len(rows) and rows[j][2] not in VALUE_NETS and rows[j][2] not in EXPAND:
                j += 1
            if j < len(rows) and rows[j][2] in VALUE_NETS:
                k = j + 1
                while k < len(rows) and rows[k][2] not in CLOSE and rows[k][2] not in EXPAND:
                    k += 1
                if k < len(rows) and rows[k][2] in CLOSE:
                    steps.append((VALUE_NETS[rows[j][2]], rows[i], rows[j], rows[k]))
                    i = k + 1
                    continue
        i += 1
    by_mode = collections.defaultdict(lambda: collections.defaultdict(list))
    for mode, x, v, c in steps:
        for s, e, name in (x, v, c):
            by_mode[mode][name].append((e - s) / 1e3)

    def line(name, d):
        return "  %-28s calls %5d  mean %7.2f us  median %7.2f  min %7.2f  max %7.2f" % (name, len(d), statistics.mean(d), statistics.median(d), min(d), max(d))
    print(f"{len(rows)} dispatches, {len(steps)} env steps in {os.path.basename(f)}")
    print("\nper value-net mode, every step of the process (burn-in, warm-up, timed region, the bench's extra passes):")
    for mode in by_mode:
        print(f" mode {mode}:")
        for name, d in by_mode[mode].items():
            print(line(name, d))
    f32 = [s for s in steps if s[0] == "f32"]
    timed = f32[a.skip:a.skip + a.steps]
    print(f"\nthe TIMED REGION: f32 steps {a.skip} .. {a.skip + len(timed) - 1} of the process ({len(timed)} steps; its last one closes with apply_kernel):")
    agg = collections.defaultdict(list)
    for _, x, v, c in timed:
        for s, e, name in (x, v, c):
            agg[name].append((e - s) / 1e3)
    for name, d in agg.items():
        print(line(name, d))
    fused = [(x, v, c) for _, x, v, c in timed if c[2] == "boundary_kernel<true>"]
    if len(fused) > 2:
        t0 = [x[0] for x, v, c in fused]
        rel = lambda k, w: statistics.mean((st[k][w] - st[0][0]) / 1e3 for st in fused)          # noqa: E731
        print(f"\ntimeline of a fused step at this size, mean over {len(fused)} steps, us from the start of expand_all_kernel (under the profiler: gaps are stretched):")
        for k, name in ((0, "expand_all_kernel"), (1, fused[0][1][2]), (2, "boundary_kernel<true>")):
            print("  %-28s %7.2f .. %7.2f" % (name, rel(k, 0), rel(k, 1)))
        period = [(b - a_) / 1e3 for a_, b in zip(t0, t0[1:]) if (b - a_) < 1e6]
        busy = statistics.mean(sum(st[k][1] - st[k][0] for k in range(3)) / 1e3 for st in fused)
        print("  next expand_all_kernel       %7.2f   (step period: mean %.2f, median %.2f us; kernels busy %.2f us of it, gaps %.2f us)"
              % (statistics.mean(period), statistics.mean(period), statistics.median(period), busy, statistics.mean(period) - busy))


if __name__ == "__main__":
    main()
