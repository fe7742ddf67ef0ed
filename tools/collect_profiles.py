"""Copy the summaries tools/profile_round.sh produced (gpurun_out/<tag>/) into profiles/ and rebuild
profiles/<round>_pmc_traffic.json (HBM bytes per launch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
src, dst = os.path.join("gpurun_out", tag), "profiles"
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench_f32.json"))
shutil.copy(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))[0], os.path.join(dst, f"{tag}_kernel_stats_default_bench.csv"))
with open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w") as f:
    f.write([l for l in open(os.path.join(src, "bench_under_rocprof.log")) if l.startswith("{")][-1])
KEYS = ("roots_kernel", "boundary_kernel", "doubles_kernel", "eval_rows_delta_kernel", "root_hidden_bf16x3_kernel", "eval_rows_f32_kernel",
        "apply_kernel", "eval_rows_f16x2_kernel", "eval_rows_bf16_kernel", "expand_kernel<3")


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
sq = os.path.join(src, "sq_counters.txt")            # tools/sq_counters.sh <tag>, when it was run for this tag
if os.path.exists(sq):
    shutil.copy(sq, os.path.join(dst, f"{tag}_sq_counters.txt"))
    os.system(f"{sys.executable} tools/valu_occupancy.py {dst}/{tag}_sq_counters.txt --json {dst}/{tag.split('_')[0]}_valu_occupancy.json")
print("profiles/ updated for", tag)
