rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "- --workload c2 --spp-per-step 256" "perturb1 --workload c2 --spp-per-step 256" "perturb2 --workload c2 --spp-per-step 256"
cat gpurun_out/sweep.log >> gpurun_out/r2s5_sweep.log
