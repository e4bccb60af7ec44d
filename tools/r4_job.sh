mkdir -p gpurun_out/r4j; rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "- --workload c4" "merl1w2 --workload c4" "merl1w3 --workload c4" "- --workload c4" 2>&1 | tee gpurun_out/r4j/sweep_c4_merl.txt
