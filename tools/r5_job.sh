timeout 900 python -m pytest tests/test_progressive.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5_n_tests.txt 2>&1; grep -E "passed|failed|error|Error" gpurun_out/r5_n_tests.txt | tail -3
python tools/progressive_rate.py 2>&1 | tail -1
for l in 1 4 8 16; do python tools/progressive_rate.py progressive_lookahead=$l 2>&1 | tail -1; done
