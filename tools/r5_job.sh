bash tools/refresh_counters.sh r5_k 7a050fa 2>&1 | tail -6
