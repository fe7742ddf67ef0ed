#!/usr/bin/env bash
# Everything DESIGN.md §4 quotes about the MFMA delta kernel (bg_eval_mfma.h), in one gpurun call:
#   bash tools/ab_build.sh base "" p1 "-DBG_MD_PROF=1" ... p7 "-DBG_MD_PROF=7" a4 "-DBG_MD_ABL=4" a8 ... a64 ... w12 "-DBG_MD_THREADS=768"
#   gpurun --timeout 900 -- 'bash tools/md_record.sh r03_mdelta'
TAG=${1:?tag}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python tools/md_fixed_bench.py capture > $OUT/fixed.txt 2>&1
python tools/md_fixed_bench.py run built valu base w12 a4 a8 a32 a64 >> $OUT/fixed.txt 2>&1
python tools/md_prof.py p1 p2 p3 p4 p5 p6 p7 > $OUT/phases.txt 2>&1
bash tools/quick_ab.sh $TAG 3 > $OUT/bench_ab.txt 2>&1
BGAMD_MFMA_DELTA=1 bash tools/sq_counters.sh ${TAG}_sq > $OUT/sq.log 2>&1
cat $OUT/fixed.txt $OUT/phases.txt $OUT/bench_ab.txt; grep -A22 mdelta gpurun_out/${TAG}_sq/sq_counters.txt | head -40
