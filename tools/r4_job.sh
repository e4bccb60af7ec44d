python tools/rank_probe.py c3 > gpurun_out/r4_i_rank_cost_c3_rerun.jsonl 2>/dev/null; cut -c1-200 gpurun_out/r4_i_rank_cost_c3_rerun.jsonl
