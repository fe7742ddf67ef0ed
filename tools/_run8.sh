mkdir -p gpurun_out/r05_c
timeout -k 10 600 python -m pytest tests -q -m gpu -x -k "bf16 or config5 or unique or staged_rows or errors_are_loud" > gpurun_out/r05_c/tests_bf16.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/r05_c/tests_bf16.txt
for v in uni2 bf4; do echo "== $v"; BGAMD_ALLOW_STALE=1 BGAMD_LIB=backgammon-engine_amd/variants/libbgamd_$v.so python tools/lanes_study.py --modes bf16 2>&1 | grep " lanes " | cut -c1-230; done > gpurun_out/r05_c/ab_bf16_four_arenas.txt 2>&1; cat gpurun_out/r05_c/ab_bf16_four_arenas.txt
