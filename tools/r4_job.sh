mkdir -p gpurun_out/r4m
rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "- --workload c4 --opt merl_batch=2" "- --workload c4 --opt merl_batch=1" "- --workload c4" "- --workload c4 --opt merl_batch=2"
