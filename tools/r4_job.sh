mkdir -p gpurun_out/r4m
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4m/gputests.txt 2>&1; grep -E "passed|failed" gpurun_out/r4m/gputests.txt; grep -E "^E |FAILED" gpurun_out/r4m/gputests.txt | head
for i in 1 2; do timeout 600 python tools/queue_kernel_rate.py 64 > gpurun_out/r4m/queue_rate$i.jsonl 2>&1; cut -c1-330 gpurun_out/r4m/queue_rate$i.jsonl | grep -v '"none"' | cut -c1-40,100-140,180-300; done
timeout 900 python tests/tools/fuzz_parity.py 150 4420 --queue > gpurun_out/r4m/fuzz_q150.txt 2>&1; tail -1 gpurun_out/r4m/fuzz_q150.txt
