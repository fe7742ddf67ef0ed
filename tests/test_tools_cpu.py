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
