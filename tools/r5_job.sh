for v in "" sw5 "" sw5; do
  if [ -n "$v" ]; then export MIPT_LIB_OVERRIDE=$PWD/pathtracer_amd/libmipt_$v.so; else unset MIPT_LIB_OVERRIDE; fi
  for wl in c2 c1; do python bench.py --workload $wl --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [$v]', '$wl', 'Mrays/s %.0f'%d['value'], {k:round(v,1) for k,v in d['stage_ms_per_step'].items()})"; done
done
