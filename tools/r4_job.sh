mkdir -p gpurun_out/r4n
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4n/gputests.txt 2>&1; head -6 gpurun_out/r4n/gputests.txt | tail -2; grep -E "^E |FAILED" gpurun_out/r4n/gputests.txt | head
timeout 600 python tools/queue_kernel_rate.py 64 > gpurun_out/r4n/queue_rate.jsonl 2>&1; cut -c1-330 gpurun_out/r4n/queue_rate.jsonl
timeout 300 python tools/queue_kernel_rate.py 64 queue_lambert=0 only=fog > gpurun_out/r4n/queue_rate_fog_general.jsonl 2>&1; cut -c1-330 gpurun_out/r4n/queue_rate_fog_general.jsonl
timeout 600 python tests/tools/fuzz_parity.py 150 77 --queue > gpurun_out/r4n/fuzz_queue_150.txt 2>&1; tail -1 gpurun_out/r4n/fuzz_queue_150.txt
