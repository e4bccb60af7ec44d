mkdir -p gpurun_out/r4e; rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "norl" "rl6" "rl6c" "rl5" "norl6" "rl6 --opt inner_min=36" "rl6 --opt refill_threshold=16" "rl6 --workload c3" "norl --workload c3" "rl6 --workload c1" "norl --workload c1" 2>&1 | tee gpurun_out/r4e/sweep_waves.txt
