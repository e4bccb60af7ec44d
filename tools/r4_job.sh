mkdir -p gpurun_out/r4k; rm -f gpurun_out/sweep.log
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4k/gputests.txt 2>&1; head -6 gpurun_out/r4k/gputests.txt | tail -2; grep -E "^E |FAILED" gpurun_out/r4k/gputests.txt | head
timeout 600 python tools/queue_kernel_rate.py 64 > gpurun_out/r4k/queue_rate_wavefront_64spp.jsonl 2>&1; cut -c1-330 gpurun_out/r4k/queue_rate_wavefront_64spp.jsonl
bash tools/sweep_libs.sh "-" "- --workload c1" "- --workload c3" 2>&1 | tee gpurun_out/r4k/sweep.txt
