#!/usr/bin/env bash
# SQ instruction / occupancy counters per kernel of the greedy step (run THROUGH gpurun; PMC passes only, no trace domains):
#   gpurun --timeout 600 -- 'bash tools/sq_counters.sh r01_v19'
set -e
TAG=${1:?tag}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# build before profiling: a rocprofv3-preloaded process has the GPU initialised before main() and must not spawn compilers
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null
export BGAMD_NO_BUILD=1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT/sq_a" -- python3 bench.py --steps 20 --burnin 100 --no-cpu-baseline --no-training-round > "$OUT/sq_a.log" 2>&1
# pass b is self-contained for tools/valu_occupancy.py: instruction count, active quad-cycles, waves and the clock (GRBM) of the SAME launches
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/sq_b" -- python3 bench.py --steps 20 --burnin 100 --no-cpu-baseline --no-training-round > "$OUT/sq_b.log" 2>&1
python3 tools/pmc_parse.py "$OUT/sq_a" > "$OUT/sq_counters.txt"
python3 tools/pmc_parse.py "$OUT/sq_b" >> "$OUT/sq_counters.txt"
cat "$OUT/sq_counters.txt"
python3 tools/valu_occupancy.py "$OUT/sq_counters.txt" --json "$OUT/valu_occupancy.json"
find "$OUT/sq_a" "$OUT/sq_b" -type f \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" -o -name "*counter_collection.csv" \) -delete   # parsed above; keep the merge small
