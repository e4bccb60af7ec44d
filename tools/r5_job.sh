timeout 900 python -m pytest tests/test_anyhit.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5_h_tests.txt 2>&1; grep -E "passed|failed|error|Error" gpurun_out/r5_h_tests.txt | tail -5
timeout 600 python tests/tools/fuzz_parity.py 200 7010 2>&1 | tail -1
rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "-" "- --workload c1" "- --workload c3 --steps 1" "- --workload c4 --steps 1"
cp gpurun_out/sweep.log gpurun_out/r5_h_sweep.log
