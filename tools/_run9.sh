mkdir -p gpurun_out/r05_c
timeout -k 10 300 python -m pytest tests/test_gpu_round5.py -q -m gpu -x -k "exploration or epsilon" > gpurun_out/r05_c/tests_explore.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/r05_c/tests_explore.txt
timeout -k 10 300 python -m pytest tests/test_gpu_full_lanes.py tests/test_gpu_round4.py -q -m gpu -x -k "epsilon or continuous or replay_beside" >> gpurun_out/r05_c/tests_explore.txt 2>&1; echo "rc=$?"; tail -2 gpurun_out/r05_c/tests_explore.txt
python tools/ab_explore_fork.py > gpurun_out/r05_c/ab_explore_fork.txt 2>&1; grep lanes gpurun_out/r05_c/ab_explore_fork.txt
