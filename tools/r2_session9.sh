timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2s9_tests_all.log 2>&1; grep -n "passed\|failed" gpurun_out/r2s9_tests_all.log | tail -3; grep -n "Error\|assert" gpurun_out/r2s9_tests_all.log | head -5
timeout 900 python tests/tools/fuzz_parity.py 200 2030 --queue > gpurun_out/r2s9_fuzz_queue.txt 2>&1; tail -1 gpurun_out/r2s9_fuzz_queue.txt
timeout 900 python tests/tools/fuzz_parity.py 200 2031 > gpurun_out/r2s9_fuzz_plain.txt 2>&1; tail -1 gpurun_out/r2s9_fuzz_plain.txt
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2s9_bench.json 2> gpurun_out/r2s9_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r2s9_bench.json')); print(d['value'], d['stage_ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_hbm_measured'))"
