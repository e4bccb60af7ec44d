for o in "refill_threshold=36 inner_min=16" "refill_threshold=30 inner_min=16" "refill_threshold=42 inner_min=16" "refill_threshold=36 inner_min=12" "refill_threshold=36 inner_min=20" "refill_threshold=32 inner_min=20"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  python bench.py --no-cpu-baseline --steps 2 --warmup 1 $args > gpurun_out/sw.json 2>gpurun_out/sw.err
  python - "$o" <<PY
import json,sys
d=json.loads(open("gpurun_out/sw.json").read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],1), {k:round(v,1) for k,v in d["stage_ms_per_step"].items()})
PY
done
