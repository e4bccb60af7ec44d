mkdir -p gpurun_out/r4h
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4h/gputests.txt 2>&1; head -6 gpurun_out/r4h/gputests.txt | tail -3
python tools/init_time.py c2 2>&1 | grep -v "^\[" | tail -3
python tools/init_time.py c4 > gpurun_out/r4h/init_c4.txt 2>&1; tail -18 gpurun_out/r4h/init_c4.txt
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4h/bench_c2.json 2> gpurun_out/r4h/bench_c2.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4h/bench_c2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_bvh_build_s','prepare_s','bvh_build')})
PY
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload c4 > gpurun_out/r4h/bench_c4.json 2> gpurun_out/r4h/bench_c4.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4h/bench_c4.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_bvh_build_s','prepare_s','bvh_build')})
PY
