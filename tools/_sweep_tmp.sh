python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "beside or packed or slices" > gpurun_out/ov_tests.txt 2>&1; grep -E "passed|failed" gpurun_out/ov_tests.txt | tail -1
for ov in 0 1; do python tools/rank_probe.py c2 ranks=1,8 overlap_anyhit=$ov > gpurun_out/rp_s.jsonl
python - $ov <<PY
import json,sys
for l in open("gpurun_out/rp_s.jsonl"):
    d=json.loads(l); print("overlap",sys.argv[1], "N", d["nranks"], d["t_max_ms"], d["t_mean_ms"], d["n_x_mean_over_t1"], d["predicted_speedup"], {k:round(sum(p[k] for p in d["ranks"])/d["nranks"],1) for k in ("extend","shadow","shade","resolve")})
PY
done
