# usage: tools/sweep.sh "opt=val opt=val" ...   each argument = one bench configuration
for cfg in "$@"; do
  opts=""; for kv in $cfg; do opts="$opts --opt $kv"; done
  python bench.py --steps 4 --warmup 1 --pmc $opts > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open('/tmp/b.json'))
print("%-40s Mrays/s %7.0f  ms/step %6.2f" % (sys.argv[1], d["value"], d["ms_per_step"]), d.get("stage_ms_per_step"), flush=True)
PY
done
