mkdir -p gpurun_out/r05_a
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py "tests/test_gpu_parity.py::test_bench_contract_line" -q -m gpu -x -s > gpurun_out/r05_a/tests_r5.txt 2>&1; echo "rc=$?"; tail -5 gpurun_out/r05_a/tests_r5.txt
python bench.py --steps 20 --warmup 5 2> gpurun_out/r05_a/bench_driver.err | tail -1 > gpurun_out/r05_a/bench_driver_style.json; echo "bench rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r05_a/bench_driver_style.json')); print(d['value'], d['ms_per_step'], d['ranks_seen'], json.dumps(d['training_round'])[:1500])"
