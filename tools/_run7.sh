mkdir -p gpurun_out/r05_c
BGAMD_NO_BUILD=1 BGAMD_ALLOW_STALE=1 python tools/ab_run.py uni2 besthoist --rounds 3 --steps 200 > gpurun_out/r05_c/ab_besthoist_65536.txt 2>&1; tail -3 gpurun_out/r05_c/ab_besthoist_65536.txt
BGAMD_NO_BUILD=1 BGAMD_ALLOW_STALE=1 python tools/ab_run.py uni2 besthoist --rounds 2 --steps 200 --extra "--games 32768" > gpurun_out/r05_c/ab_besthoist_32768.txt 2>&1; tail -3 gpurun_out/r05_c/ab_besthoist_32768.txt
