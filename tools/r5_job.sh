timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_subsurface.py tests/test_compositing.py tests/test_fog.py -m gpu -x -q > gpurun_out/r5_j_tests.txt 2>&1; grep -E "passed|failed|error|Error" gpurun_out/r5_j_tests.txt | tail -3
timeout 600 python tests/tools/fuzz_parity.py 150 7020 2>&1 | tail -1
timeout 600 python tests/tools/fuzz_parity.py 100 7021 --queue 2>&1 | tail -1
rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "-" "- --workload c1" "- --workload c3 --steps 1" "- --workload c4 --steps 1"
cp gpurun_out/sweep.log gpurun_out/r5_j_sweep.log
