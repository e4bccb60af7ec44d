mkdir -p gpurun_out/r4l
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4l/gputests.txt 2>&1; head -6 gpurun_out/r4l/gputests.txt | tail -2; grep -E "^E |FAILED" gpurun_out/r4l/gputests.txt | head
