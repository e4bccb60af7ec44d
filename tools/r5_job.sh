timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -x -q > gpurun_out/r5_r_tests.txt 2>&1; grep -E "passed|failed|error|Error" gpurun_out/r5_r_tests.txt | tail -3; tail -30 gpurun_out/r5_r_tests.txt | grep -v "^$" | tail -12
python -c "
import __graft_entry__ as g
g.smoke()"
