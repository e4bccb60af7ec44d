# usage: tools/sweep_libs.sh "<lib-suffix or '-'> <bench args>" ...   one bench run per argument
for cfg in "$@"; do
  set -- $cfg; lib=$1; shift
  if [ "$lib" = "-" ]; then unset MIPT_LIB_OVERRIDE; else export MIPT_LIB_OVERRIDE=$PWD/pathtracer_amd/libmipt_$lib.so; fi
  python bench.py --steps 3 --warmup 1 --pmc "$@" > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open('/tmp/b.json'))
print("%-44s Mrays/s %7.0f  ms/step %6.2f" % (sys.argv[1], d["value"], d["ms_per_step"]), {k: round(v,2) for k,v in d.get("stage_ms_per_step",{}).items()}, flush=True)
open('gpurun_out/sweep.log','a').write("%-44s Mrays/s %7.0f  ms/step %6.2f %s\n" % (sys.argv[1], d["value"], d["ms_per_step"], {k: round(v,2) for k,v in d.get("stage_ms_per_step",{}).items()}))
PY
done
