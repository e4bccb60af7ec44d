timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_l_gputests.txt 2>&1; grep -E "passed|failed|error|Error" gpurun_out/r5_l_gputests.txt | tail -3
bash tools/kstats.sh r5_l --workload c2 > gpurun_out/r5_l_kstats.txt 2>&1; head -8 gpurun_out/r5_l_kstats.txt; rm -rf gpurun_out/kstats_r5_l
