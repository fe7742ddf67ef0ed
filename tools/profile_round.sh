#!/usr/bin/env bash
# Regenerate the judged evidence of a build on the GPU box (run from the repo root THROUGH gpurun):
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r02_v1'
# Writes under gpurun_out/<tag>/ (gpurun merges it back); tools/collect_profiles.py <tag> then copies the summaries
# into profiles/.  PMC counters go in their own passes (never together with the trace domains).
set -e
TAG=${1:?tag, e.g. r02_v1}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# build before profiling: a rocprofv3-preloaded process has the GPU initialised before main() and must not spawn compilers
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null
export BGAMD_NO_BUILD=1
python3 bench.py 2>/dev/null | tail -1 > "$OUT/bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --no-training-round > "$OUT/bench_under_rocprof.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 20 --burnin 100 --no-cpu-baseline --no-training-round > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 20 --burnin 100 --no-cpu-baseline --no-training-round > "$OUT/pmc_write.log" 2>&1
# per-mode kernel statistics + the 65 536-lane step timeline, from the per-dispatch trace of the SAME run (the trace is too large to travel)
python3 tools/trace_by_mode.py "$OUT/stats" > "$OUT/kernel_trace_by_mode.txt" 2>&1 || true
python3 -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_ms'])"
# gpurun merges at most 64 MiB back: the per-dispatch traces are not needed by tools/collect_profiles.py
find "$OUT" -type f \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" \) -delete
du -sh "$OUT"
