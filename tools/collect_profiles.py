"""Copy the summaries tools/profile_round.sh produced (gpurun_out/<tag>/) into profiles/ and rebuild
profiles/<round>_pmc_traffic.json (HBM bytes per launch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
src, dst = os.path.join("gpurun_out", tag), "profiles"
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench_f32.json"))
shutil.copy(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))[0], os.path.join(dst, f"{tag}_kernel_stats_default_bench.csv"))
with open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w") as f:
    f.write([l for l in open(os.path.join(src, "bench_under_rocprof.log")) if l.startswith("{")][-1])
KEYS = ("roots_kernel", "boundary_kernel<true>", "boundary_kernel<false>", "doubles_kernel", "eval_rows_delta_kernel", "root_hidden_bf16x3_kernel", "root_hidden_resident_kernel", "eval_rows_f32_kernel",
        "apply_kernel", "eval_rows_f16x2_kernel", "eval_rows_bf16_kernel", "expand_kernel<3", "expand_all_kernel")


def load(d, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0])):
        if r["Counter_Name"] == name:
            for key in KEYS:
                if key in r["Kernel_Name"]:
                    agg["expand_kernel<3>" if key.startswith("expand_kernel<3") else key].append(float(r["Counter_Value"]))
    return {k: sum(v[-20:]) / len(v[-20:]) for k, v in agg.items()}


f, w = load(os.path.join(src, "pmc_fetch"), "FETCH_SIZE"), load(os.path.join(src, "pmc_write"), "WRITE_SIZE")
out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 20 --burnin 100 at 65 536 lanes, {tag} build, "
                 "mean of the last 20 launches; FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); MB = 1e6 B",
       "kernels": {k: {"FETCH_SIZE_KB_per_launch_raw": round(f[k], 1), "WRITE_SIZE_KB_per_launch_raw": round(w.get(k, 0), 1),
                       "hbm_read_MB_corrected_x2": round(2 * f[k] * 1024 / 1e6, 2), "hbm_write_MB": round(w.get(k, 0) * 1024 / 1e6, 2),
                       "traffic_MB": round((2 * f[k] + w.get(k, 0)) * 1024 / 1e6, 2)} for k in f}}
json.dump(out, open(os.path.join(dst, tag.split("_")[0] + "_pmc_traffic.json"), "w"), indent=1)
tbm = os.path.join(src, "kernel_trace_by_mode.txt")   # tools/trace_by_mode.py: per-mode averages + the step timeline of the same rocprof run
if os.path.exists(tbm):
    shutil.copy(tbm, os.path.join(dst, f"{tag}_kernel_trace_by_mode.txt"))
sq = os.path.join(src, "sq_counters.txt")            # tools/sq_counters.sh <tag>, when it was run for this tag
if os.path.exists(sq):
    shutil.copy(sq, os.path.join(dst, f"{tag}_sq_counters.txt"))
    os.system(f"{sys.executable} tools/valu_occupancy.py {dst}/{tag}_sq_counters.txt --json {dst}/{tag.split('_')[0]}_valu_occupancy.json")
drv = os.path.join(src, "bench_driver_style.json")    # tools/round_check.sh: bench.py --steps 20 --warmup 5, as the driver calls it
if os.path.exists(drv):
    shutil.copy(drv, os.path.join(dst, f"{tag}_bench_driver_style_steps20.json"))
btr = os.path.join(src, "bench_with_training_round.json")
if os.path.exists(btr):
    shutil.copy(btr, os.path.join(dst, f"{tag}_bench_with_training_round.json"))
tdb = os.path.join(src, "td_bench.txt")
if os.path.exists(tdb):                               # the learner's summary: tools/td_bench.py, tools/train_breakdown.py, rocprof of a 65 536-game replay
    with open(os.path.join(dst, f"{tag}_learner.txt"), "w") as f:
        f.write(f"TD(lambda) learner kernels (csrc/bg_learner.h), MI355X, {tag} build.\n\n"
                "tools/td_bench.py (play_round with epsilon 0.05, then one lock-step DeviceTDLambdaLearner.replay_rows of the whole round, HIP-event timing of the\n"
                "trace kernel).  Column-sparse, lazily scaled traces: 'read' = W1 trace columns of features a game has activated so far, 'written' = those whose feature\n"
                "is non-zero at the step (or all of them on the steps that fold the scale back in); 'moved' = bytes read + written; dense-equivalent = 204 808 B per update.\n\n")
        f.write("".join(l for l in open(tdb) if l.startswith("n=") or l.startswith("   torch")))
        tb = os.path.join(src, "train_breakdown.txt")
        if os.path.exists(tb):
            f.write("\ntools/train_breakdown.py: one 65 536-game training round (self-play from a frozen snapshot with the turn log, then its replay: lock-step whole,\n"
                    "lock-step in sub-rounds, streamed through k slots)\n")
            f.write("".join(l for l in open(tb) if "amdgpu" not in l))
        ks = glob.glob(os.path.join(src, "td_stats", "*", "*kernel_stats.csv"))
        if ks:
            f.write("\nrocprofv3 --kernel-trace --stats of 'python3 tools/td_bench.py 65536' (two lock-step replays of one 65 536-game round; the env kernels are the self-play):\n")
            for i, r in enumerate(csv.DictReader(open(ks[0]))):
                if i >= 12:
                    break
                f.write("  %-64s calls %5s  total %9.2f ms  avg %8.1f us  %5s %%\n" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                                     float(r["AverageNs"]) / 1e3, r["Percentage"]))
print("profiles/ updated for", tag)
