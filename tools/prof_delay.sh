export TMPDIR=/tmp BGAMD_NO_BUILD=1
mkdir -p gpurun_out/r04_v47
for d in 0 1; do
python tools/train_pipeline.py --modes cont,cont_pipe --delay $d 2>&1 | grep lanes
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_v47/delay$d -- python3 tools/train_pipeline.py --modes cont --rounds 3 --warm 1 --delay $d > gpurun_out/r04_v47/delay${d}.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r04_v47/delay$d/*/*kernel_stats.csv")[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<8: print("  %-70s calls %6s avg %8.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
find gpurun_out/r04_v47 -name "*kernel_trace.csv" -delete; find gpurun_out/r04_v47 -name "*agent_info.csv" -delete
