mkdir -p gpurun_out/r05_c
export BGAMD_NO_BUILD=1 BGAMD_ALLOW_STALE=1
python tools/ab_run.py old uni --rounds 3 --steps 200 > gpurun_out/r05_c/ab_uniform_65536.txt 2>&1; tail -3 gpurun_out/r05_c/ab_uniform_65536.txt
python tools/ab_run.py old uni --rounds 2 --steps 200 --extra "--games 32768" > gpurun_out/r05_c/ab_uniform_32768.txt 2>&1; tail -3 gpurun_out/r05_c/ab_uniform_32768.txt
unset BGAMD_NO_BUILD BGAMD_ALLOW_STALE
timeout -k 10 500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py -q -m gpu -x > gpurun_out/r05_c/tests.txt 2>&1; echo "rc=$?"; tail -4 gpurun_out/r05_c/tests.txt
