#!/usr/bin/env python3
"""How much of a run's GPU time has kernels of BOTH streams executing at once (round 5, the co-residency experiment: tools/two_halves.py under
`rocprofv3 --kernel-trace`).  Reads the per-dispatch trace, takes the dispatches of the LAST measurement of the tool (two envs on two streams), and prints: the
span, the time with >= 1 / >= 2 kernels in flight, and for every pair of kernel names the time they ran side by side.
    python tools/overlap_from_trace.py gpurun_out/<dir>"""
import collections, csv, glob, os, sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("bg::", "").replace("void ", "")
    b = n.split("(")[0]
    return b if b.startswith("boundary_kernel") else b.split("<")[0]


f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in csv.DictReader(open(f))), key=lambda x: x[0])
# the two-stream measurement: the longest stretch in which dispatches of two queues alternate; simply: the dispatches whose queue is not the first env's
queues = collections.Counter(q for _, _, _, q in rows)
print("dispatches per queue:", dict(queues))
two = [q for q, _ in queues.most_common()][:3]
# take the window between the first and the last dispatch of the LEAST used of the two busiest non-default queues (the second half-env only runs in the pair measurement)
cand = sorted(queues, key=lambda q: queues[q])
pairq = [q for q in cand if queues[q] > 1000][:1]
sel = [r for r in rows if r[3] in pairq]
t0, t1 = sel[0][0], sel[-1][1]
win = [r for r in rows if r[0] >= t0 and r[1] <= t1]
ev = []
for s, e, n, q in win:
    ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort()
active, last, ge1, ge2 = collections.Counter(), t0, 0, 0
pair = collections.Counter()
for t, d, n in ev:
    dt = t - last
    k = sum(active.values())
    if k >= 1: ge1 += dt
    if k >= 2:
        ge2 += dt
        names = sorted(x for x in active.elements())
        for i in range(len(names)):
            for j in range(i + 1, len(names)):
                pair[(names[i], names[j])] += dt
    active[n] += d
    if active[n] <= 0: del active[n]
    last = t
span = t1 - t0
print("window of the two-stream measurement: %.1f ms, %d dispatches; >= 1 kernel in flight %.1f %%, >= 2 in flight %.1f %% of the window" % (span / 1e6, len(win), 100.0 * ge1 / span, 100.0 * ge2 / span))
for (a, b), dt in pair.most_common(8):
    print("   %-28s beside %-28s %6.1f %% of the window" % (a, b, 100.0 * dt / span))
