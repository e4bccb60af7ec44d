rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "- --workload c2" "- --workload c2 --opt merge_traverse=1" "- --workload c3" "- --workload c3 --opt merge_traverse=1" "- --workload c2" "- --workload c2 --opt merge_traverse=1"
