S=r5_e
timeout 2400 python -m pytest tests/test_anyhit.py tests/test_gpu_parity.py tests/test_gpu_bvh_build.py tests/test_fog.py -m gpu -x -q > gpurun_out/${S}_gputests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/${S}_gputests.txt | tail -3
rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "-" "- --workload c1" "- --workload c3 --steps 1" "- --workload c4 --steps 1"
cp gpurun_out/sweep.log gpurun_out/${S}_sweep.log
python tools/init_time.py c2 2>&1 | tail -12
