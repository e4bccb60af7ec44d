python bench.py --workload c3 --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; tail -2 gpurun_out/bench_c3.err
python bench.py --workload c4 --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; tail -2 gpurun_out/bench_c4.err
bash tools/pmc.sh c1 --workload c1 > /dev/null 2>&1
python - <<'PY'
import json
for n in ("c3","c4"):
    try:
        d=json.load(open(f"gpurun_out/bench_{n}.json"))
        print(n, 'Mrays/s %.0f frac %.3f'%(d['value'], d['roofline']['frac']), d.get('stage_ms_per_step'), 'ms/step %.2f'%d['ms_per_step'], 'build %.1f'%d['host_bvh_build_s'], d['config']['workload'])
    except Exception as e: print(n, "failed", e)
PY
