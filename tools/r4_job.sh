mkdir -p gpurun_out/r4m
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4m/gputests.txt 2>&1; grep -E "passed|failed" gpurun_out/r4m/gputests.txt; grep -E "^E |FAILED" gpurun_out/r4m/gputests.txt | head
timeout 900 python tests/tools/fuzz_parity.py 300 4405 > gpurun_out/r4m/fuzz_300.txt 2>&1; tail -1 gpurun_out/r4m/fuzz_300.txt
timeout 900 python tests/tools/fuzz_parity.py 150 4406 --queue > gpurun_out/r4m/fuzz_q150.txt 2>&1; tail -1 gpurun_out/r4m/fuzz_q150.txt
python tools/progressive_rate.py > gpurun_out/r4m/progressive.txt 2>&1; tail -5 gpurun_out/r4m/progressive.txt
