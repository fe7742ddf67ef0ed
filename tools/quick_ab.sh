#!/usr/bin/env bash
# quick same-box A/B of the value-net stage: BGAMD_MFMA_DELTA=1 (round 3's MFMA delta kernel) vs the default (round 2's VALU kernel), interleaved
# usage (on the GPU box): bash tools/quick_ab.sh TAG [rounds]
TAG=${1:?tag}; R=${2:-2}
mkdir -p gpurun_out/$TAG
for i in $(seq 1 $R); do
  BGAMD_MFMA_DELTA=1 python bench.py --quick 2>/dev/null | tail -1 > gpurun_out/$TAG/mfma_$i.json
  python bench.py --quick 2>/dev/null | tail -1 > gpurun_out/$TAG/valu_$i.json
done
python - <<PY
import json,glob
for k in ("mfma","valu"):
    for f in sorted(glob.glob("gpurun_out/$TAG/%s_*.json"%k)):
        d=json.load(open(f)); e=d["kernels"]["eval"]
        print(k, "step %.4f ms  value-net kernel %.4f ms  stage %.4f  steps/s %.1fM" % (d["ms_per_step"], e["avg_ms"], e.get("value_net_stage_ms",0), d["value"]/1e6))
PY
