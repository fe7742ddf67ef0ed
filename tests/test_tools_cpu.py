"""The measurement tooling is reproducible from what is committed: the VALU issue occupancy bench.py
quotes comes out of profiles/*_sq_counters.txt by one command, and the roofline record of the committed bench line is a
fraction of the roof that bounds the executed work."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_valu_occupancy_reproduces_from_committed_counters(tmp_path):
    rec = json.load(open(os.path.join(ROOT, "profiles", "r02_valu_occupancy.json")))
    src = os.path.join(ROOT, rec["source"])
    assert os.path.exists(src), rec["source"]
    out = tmp_path / "occ.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_occupancy.py"), src, "--json", str(out)],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    again = json.load(open(out))
    k = "eval_rows_delta_kernel"
    assert again[k]["valu_issue_occupancy"] == rec[k]["valu_issue_occupancy"]
    assert 0.3 < rec[k]["valu_issue_occupancy"] < 0.9 and 4.0 <= rec[k]["cycles_per_valu_instruction"] <= 8.0


def test_committed_bench_line_roofline_is_a_fraction():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02_v*_bench_f32.json")))
    assert files
    d = json.load(open(files[-1]))
    r = d["roofline"]
    assert r["bound"] == "valu" and r["kernel"] == "eval_rows_delta_kernel"
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] * 1e12 * r["avg_ms"] * 1e-3 - r["flop_per_launch"]) < 0.01 * r["flop_per_launch"]
    assert 0.0 < r["lds_frac"] < 1.0 and r["traffic"] > 0
    # rocprof's average duration of the same kernel in the same command agrees with the live HIP-event timing
    stats = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02_v*_kernel_stats_default_bench.csv")))[-1]
    avg_ns = None
    for line in open(stats):
        if "eval_rows_delta_kernel" in line:
            avg_ns = float(line.rsplit('",', 1)[1].split(",")[2])
    assert avg_ns and abs(avg_ns * 1e-6 - r["avg_ms"]) < 0.1 * r["avg_ms"]
    c = d["cpu_baseline"]
    assert c["host"]["cpu_model"] and c["all_cores"]["cores"] == c["host"]["usable_cores"] and c["reference_engine"]["value"]


def test_round5_evidence_is_consistent():
    """The committed round-5 evidence hangs together: the bench line is a fraction of its roof; the trace summary of the SAME rocprof run
    (tools/trace_by_mode.py) puts the value net's timed-region average within 10 % of the bench's HIP-event time and the three launches'
    sum within 5 % of the step period; the occupancy file reproduces from its counter summary; the step-level figure is the duration-
    weighted mean of the three launches."""
    import re
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_v62_bench_f32.json")))
    r = d["roofline"]
    assert r["bound"] == "valu" and 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["ranks_seen"] == 1 and d["training_round"]["replicas_identical"] and "error" not in d["training_round"]
    txt = open(os.path.join(ROOT, "profiles", "r05_v62_kernel_trace_by_mode.txt")).read()
    timed = txt[txt.index("the TIMED REGION"):]
    avg = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\S+)\s+calls\s+\d+\s+mean\s+([\d.]+) us", timed, re.M)}
    assert abs(avg["eval_rows_delta_kernel"] * 1e-3 - r["avg_ms"]) < 0.1 * r["avg_ms"]
    period = float(re.search(r"step period: mean ([\d.]+)", timed).group(1))
    busy = avg["expand_all_kernel"] + avg["eval_rows_delta_kernel"] + avg["boundary_kernel<true>"]
    assert abs(busy - period) < 0.05 * period and abs(period * 1e-3 - d["ms_per_step"]) < 0.05 * d["ms_per_step"]
    occ = json.load(open(os.path.join(ROOT, "profiles", "r05_valu_occupancy.json")))
    parts = [occ[k] for k in ("expand_all_kernel", "eval_rows_delta_kernel", "boundary_kernel<true>")]
    step = sum(x["duration_us"] * x["valu_issue_occupancy"] for x in parts) / sum(x["duration_us"] for x in parts)
    assert 0.3 < step < 0.7
    g = json.load(open(os.path.join(ROOT, "profiles", "r05_v62_bench_2ranks_gloo.json")))
    assert g["n_gpus"] == 2 and g["ranks_seen"] == 2 and g["launched_by"] == "bench.py launch_ranks" and g["training_round"]["replicas_identical"]
