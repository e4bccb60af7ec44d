python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_lenticular.py -q -m gpu -x > gpurun_out/cam_tests.txt 2>&1; grep -E "passed|failed" gpurun_out/cam_tests.txt | tail -1
for v in "" nocam "" nocam; do
  if [ -n "$v" ]; then export MIPT_LIB_OVERRIDE=$PWD/pathtracer_amd/libmipt_$v.so; else unset MIPT_LIB_OVERRIDE; fi
  for wl in c2 c4; do python bench.py --workload $wl --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [$v]', '$wl', 'Mrays/s %.0f'%d['value'], {k:round(v,1) for k,v in d['stage_ms_per_step'].items()})"; done
done
