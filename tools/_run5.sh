mkdir -p gpurun_out/r05_v61
python tools/train_pipeline.py > gpurun_out/r05_v61/train_pipeline.txt 2>&1; grep -v amdgpu gpurun_out/r05_v61/train_pipeline.txt | tail -9 | cut -c1-300
python tools/train_pipeline.py --modes cont,cont_pipe --delay 1 > gpurun_out/r05_v61/train_pipeline_delay1.txt 2>&1; grep lanes gpurun_out/r05_v61/train_pipeline_delay1.txt | cut -c1-300
python tools/td_bench.py 512 4096 16384 32768 65536 > gpurun_out/r05_v61/td_bench.txt 2>&1; grep "^n=" gpurun_out/r05_v61/td_bench.txt | cut -c1-160
python tools/train_breakdown.py 65536 0 s4096 s2048 s1024 > gpurun_out/r05_v61/train_breakdown.txt 2>&1; grep -v amdgpu gpurun_out/r05_v61/train_breakdown.txt | cut -c1-200
