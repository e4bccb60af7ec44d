python tools/init_time.py c2 2>&1 | grep -v "^\[" | cut -c88-260
for i in 1 2 3; do python bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp-per-step 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench init', round(d['host_bvh_build_s'],4), round(d['prepare_s'],4), d['bvh_build'])"; done
