python tools/rank_probe.py c2 > gpurun_out/r4_k_rank_cost_c2_rerun.jsonl 2>/dev/null; cut -c1-140 gpurun_out/r4_k_rank_cost_c2_rerun.jsonl
