timeout 900 python -m pytest tests/test_progressive.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5_f_tests.txt 2>&1; grep -E "passed|failed|error|Error" gpurun_out/r5_f_tests.txt | tail -5
for l in 1 2 4 8 16; do python tools/progressive_rate.py progressive_lookahead=$l 2>&1 | tail -1; done
bash tools/progressive_trace.sh after > /dev/null 2>&1; head -45 gpurun_out/ptrace_after.txt; tail -1 gpurun_out/ptrace_after.txt
