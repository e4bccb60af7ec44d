mkdir -p gpurun_out/r4i
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4i/gputests.txt 2>&1; head -8 gpurun_out/r4i/gputests.txt | tail -5; grep -E "^E |FAILED" gpurun_out/r4i/gputests.txt | head
timeout 900 python tests/tools/fuzz_parity.py 250 41 --queue --spheres > gpurun_out/r4i/fuzz_queue_spheres_250.txt 2>&1; tail -2 gpurun_out/r4i/fuzz_queue_spheres_250.txt; grep -c "s " gpurun_out/r4i/fuzz_queue_spheres_250.txt
timeout 900 python tests/tools/fuzz_parity.py 200 42 --queue --bare-spheres > gpurun_out/r4i/fuzz_queue_bare_spheres_200.txt 2>&1; tail -2 gpurun_out/r4i/fuzz_queue_bare_spheres_200.txt
grep -v "identical 1.000000 max|err|/white 0$" gpurun_out/r4i/fuzz_*.txt | grep -v "identical 1.000000 max|err|/white 0  p" | head
