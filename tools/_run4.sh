mkdir -p gpurun_out/r05_v61
export TMPDIR=/tmp
python tools/lanes_study.py > gpurun_out/r05_v61/lanes_study.txt 2>&1; grep "lanes" gpurun_out/r05_v61/lanes_study.txt | grep nofork | cut -c1-200
export BGAMD_NO_BUILD=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_v61/lanes_32768_trace -- python3 tools/lanes_study.py --only 32768,f32,nofork > gpurun_out/r05_v61/lanes_32768_trace.log 2>&1
python tools/lanes_study.py --timeline gpurun_out/r05_v61/lanes_32768_trace > gpurun_out/r05_v61/lanes_32768_timeline.txt 2>&1; cat gpurun_out/r05_v61/lanes_32768_timeline.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_v61/lanes_32768_trace_bf16 -- python3 tools/lanes_study.py --only 32768,bf16,nofork > gpurun_out/r05_v61/lanes_32768_trace_bf16.log 2>&1
python tools/lanes_study.py --timeline gpurun_out/r05_v61/lanes_32768_trace_bf16 > gpurun_out/r05_v61/lanes_32768_timeline_bf16.txt 2>&1; cat gpurun_out/r05_v61/lanes_32768_timeline_bf16.txt
unset BGAMD_NO_BUILD
python tools/soak.py > gpurun_out/r05_v61/soak.txt 2>&1; grep -v amdgpu gpurun_out/r05_v61/soak.txt | cut -c1-330
find gpurun_out/r05_v61 -type f \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" \) -delete
