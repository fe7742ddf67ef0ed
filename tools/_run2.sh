mkdir -p gpurun_out/r05_b
export BGAMD_NO_BUILD=1
python tools/ab_run.py base d512 d768 x256w1 x256w2 --rounds 2 --steps 200 --extra "--games 32768" > gpurun_out/r05_b/ab_stage_a_32768.txt 2>&1; tail -7 gpurun_out/r05_b/ab_stage_a_32768.txt
python tools/ab_run.py base d512 d768 --rounds 2 --steps 200 > gpurun_out/r05_b/ab_stage_a_65536.txt 2>&1; tail -4 gpurun_out/r05_b/ab_stage_a_65536.txt
for v in base d512x d768x; do echo "== $v"; BGAMD_LIB=backgammon-engine_amd/variants/libbgamd_$v.so python tools/two_halves.py 2>&1 | grep -v amdgpu; echo "== ${v}e, root pass as its own launch"; BGAMD_ROOT_IN_BOUNDARY=0 BGAMD_LIB=backgammon-engine_amd/variants/libbgamd_${v}e.so python tools/two_halves.py 2>&1 | grep -v amdgpu; done > gpurun_out/r05_b/two_halves.txt 2>&1; cat gpurun_out/r05_b/two_halves.txt
python tools/doubles_ancestor_stats.py > gpurun_out/r05_b/doubles_ancestor_stats.txt 2>&1; tail -8 gpurun_out/r05_b/doubles_ancestor_stats.txt | cut -c1-700
