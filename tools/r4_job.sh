mkdir -p gpurun_out/r4m; rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "-" "- --opt state_skew=1" "- --opt state_skew=17" "- --opt state_skew=65" "- --opt state_skew=1025" "- --opt state_skew=4113" "-" "- --opt state_skew=17" 2>&1 | tee gpurun_out/r4m/sweep_skew.txt
