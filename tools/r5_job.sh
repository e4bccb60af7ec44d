timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r5_o_gputests.txt 2>&1; grep -E "passed|failed|error|Error" gpurun_out/r5_o_gputests.txt | tail -3
timeout 600 python tests/tools/fuzz_parity.py 200 7040 2>&1 | tail -1
python tools/init_time.py c2 2>&1 | tail -2
python tools/init_time.py c4 2>&1 | tail -2
