timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2s10_tests_all.log 2>&1; grep -n "passed\|failed" gpurun_out/r2s10_tests_all.log | tail -3
timeout 900 python tests/tools/fuzz_parity.py 150 2032 --queue > gpurun_out/r2s10_fuzz_queue.txt 2>&1; tail -1 gpurun_out/r2s10_fuzz_queue.txt
timeout 900 python tests/tools/fuzz_parity.py 150 2033 > gpurun_out/r2s10_fuzz_plain.txt 2>&1; tail -1 gpurun_out/r2s10_fuzz_plain.txt
rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "- --workload c2 --spp-per-step 256" "- --workload c1 --spp-per-step 256" "- --workload c3 --spp-per-step 256"
cp gpurun_out/sweep.log gpurun_out/r2s10_sweep.log
