#!/usr/bin/env python3
"""Where a greedy step's time goes at config 5's per-GPU share (32 768 lanes) and around it: per lane count and value-net
mode, the step with the root pass forked onto the env's second stream and without (the env decides by lane count: 28 672 was set on the r01
build), eager (bgamd_env_run_greedy) against a captured HIP graph of 8 steps replayed back to back, and the per-kernel HIP-event times.
    python tools/lanes_study.py [--lanes 16384,32768,65536] [--steps 400]
    python tools/lanes_study.py --timeline gpurun_out/<dir>     parses a `rocprofv3 --kernel-trace` CSV of a run of this tool into the mean
                                                                start / end of every kernel of a step relative to the boundary launch"""
import argparse
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "backgammon-engine_amd")]


def timeline(d):
    import collections, csv, glob
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    def short(n):
        return n.replace("(anonymous namespace)::", "").replace("bg::", "").replace("void ", "").split("(")[0].split("<")[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda x: x[0])
    first = "boundary_kernel" if any(r[2] == "boundary_kernel" for r in rows) else "roots_kernel"     # (the dense value-net modes: roots / apply are launches of their own)
    steps, cur = [], None
    for s, e, name, full in rows:
        if name == first:
            if cur:
                steps.append(cur)
            cur = [(s, e, name)]
        elif cur is not None:
            cur.append((s, e, "expand_kernel<LEAF>" if name == "expand_kernel" else name))
    steps = [st for st in steps if 3 <= len(st) <= 7 and (st[-1][1] - st[0][0]) < 400000][-200:]
    if steps:
        common = max(set(len(st) for st in steps), key=[len(st) for st in steps].count)
        steps = [st for st in steps if len(st) == common]
    agg = collections.defaultdict(list)
    for st in steps:
        t0 = st[0][0]
        for s, e, name in st:
            agg[name].append((s - t0, e - t0))
    nxt = [b[0][0] - a[0][0] for a, b in zip(steps, steps[1:]) if b[0][0] - a[0][0] < 400000]
    print(f"{len(steps)} steady-state steps; mean start / end of each kernel in us from the start of {first} (rocprof stretches the gaps):")
    for name, v in sorted(agg.items(), key=lambda kv: sum(x[0] for x in kv[1]) / len(kv[1])):
        print("  %-34s %7.1f .. %7.1f   (%.1f us)" % (name, sum(x[0] for x in v) / len(v) / 1e3, sum(x[1] for x in v) / len(v) / 1e3,
                                                    sum(x[1] - x[0] for x in v) / len(v) / 1e3))
    if nxt:
        print("  next %-29s %7.1f" % (first, sum(nxt) / len(nxt) / 1e3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", default="16384,32768,65536")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--modes", default="f32,bf16")
    ap.add_argument("--only", default="", help="one configuration for a profiler run: lanes,mode,fork|nofork")
    ap.add_argument("--timeline", default="")
    a = ap.parse_args()
    if a.timeline:
        return timeline(a.timeline)
    import numpy as np, torch
    import backgammon_env as bg
    w = np.fromfile(os.path.join(ROOT, "tests/golden/tdgammonNEW100k.f32"), dtype=np.float32)
    configs = [(int(n), m, f) for n in a.lanes.split(",") for m in a.modes.split(",") for f in ("fork", "nofork")]
    if a.only:
        n, m, f = a.only.split(",")
        configs = [(int(n), m, f)]
    for n, mode, fork in configs:
        prec = {"f32": bg.F32, "bf16": bg.BF16, "f16x2": bg.F16X2}[mode]
        if mode != "f32" and fork == "fork":
            continue                                              # only the incremental path has a root pass to fork
        os.environ.pop("BGAMD_OVERLAP", None); os.environ.pop("BGAMD_NO_OVERLAP", None)
        os.environ["BGAMD_OVERLAP" if fork == "fork" else "BGAMD_NO_OVERLAP"] = "1"
        env = bg.VecGame(n, seed=3, arena_rows=n * 512)
        env.load_weights(w)
        env.run_greedy(200, precision=prec)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.run_greedy(a.steps, precision=prec)
        torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / a.steps
        if a.only:
            print(f"{n} lanes {mode} {fork}: {eager * 1e6:.1f} us per step", flush=True)
            continue
        side, g = torch.cuda.Stream(), torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                env.run_greedy(8, precision=prec)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps // 8):
            g.replay()
        torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / (a.steps // 8 * 8)
        env.time_kernels(True); env.kernel_times()
        env.run_greedy(64, precision=prec)
        kt = env.kernel_times(); env.time_kernels(False)
        per = {k: v["ms"] / 64 * 1e3 for k, v in kt.items() if v["launches"]}
        print(f"{n:6d} lanes {mode:5s} {fork:6s}: eager {eager * 1e6:6.1f} us/step = {n / eager / 1e6:6.1f} M steps/s | graph of 8 steps {graph * 1e6:6.1f} us/step = {n / graph / 1e6:6.1f} M | "
              + "  ".join(f"{k} {v:.1f}" for k, v in per.items()) + f"  (sum {sum(per.values()):.1f} us; bracketing every group costs ~20 us per step of its own)", flush=True)
        del env


if __name__ == "__main__":
    main()
