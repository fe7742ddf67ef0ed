#!/usr/bin/env bash
# SQ counters of ONE value-net kernel under bench.py (two PMC passes, no trace domains beside --kernel-trace):
#   gpurun -- 'bash tools/sq_kernel.sh TAG eval_rows_d16_kernel --precision f16x2'
# SQ_CMD="tools/train_breakdown.py 65536 s2048" profiles that script instead of bench.py (its arguments replace bench.py's)
set -e
TAG=${1:?tag}; KERNEL=${2:?kernel name substring}; shift 2
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null
export BGAMD_NO_BUILD=1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/a" -- python3 ${SQ_CMD:-bench.py --quick --steps 20 --burnin 100} "$@" > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/b" -- python3 ${SQ_CMD:-bench.py --quick --steps 20 --burnin 100} "$@" > "$OUT/b.log" 2>&1
python3 - "$OUT" "$KERNEL" <<'PY'
import csv, glob, sys, collections
out, key = sys.argv[1], sys.argv[2]
for sub in ("a", "b"):
    f = glob.glob(out + "/" + sub + "/*/*counter_collection.csv")
    if not f:
        print(sub, "no counters (see", out + "/" + sub + ".log)"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if key in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for c, v in sorted(agg.items()):
        v = v[-200:]
        print("   %-32s %.4g" % (c, sum(v) / len(v)))
PY
find "$OUT" -type f \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" -o -name "*counter_collection.csv" \) -delete      # parsed above; keep the merge small
