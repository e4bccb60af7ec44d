# round 2, GPU session 1: parity of the new resolve kernel + new tests, then A/B of the traversal variants and resolve bands
python -m pytest tests/test_gpu_parity.py tests/test_scene_files.py tests/test_multi_rank_cpu.py -m gpu -q -x > gpurun_out/r2s1_tests.log 2>&1; tail -3 gpurun_out/r2s1_tests.log
rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "- --workload c2" "- --workload c2 --opt resolve_rows=0" "- --workload c2 --opt resolve_rows=8" "- --workload c2 --opt resolve_rows=32" "- --workload c2 --opt resolve_rows=64" \
  "blk512 --workload c2" "top63 --workload c2" "top127 --workload c2" "top255 --workload c2" \
  "- --workload c1" "top127 --workload c1" "- --workload c3" "top127 --workload c3"
python -m pytest tests -m gpu -q -x > gpurun_out/r2s1_tests_all.log 2>&1; tail -3 gpurun_out/r2s1_tests_all.log
