# GPU session: tests, then the default bench (configs[2]) timed end to end, then configs[1]
python -m pytest tests -m gpu -q -x > gpurun_out/gputests.log 2>&1; tail -2 gpurun_out/gputests.log
t0=$(date +%s)
python bench.py --steps 32 --warmup 2 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; tail -2 gpurun_out/bench_c2.err
echo "c2 bench wall $(( $(date +%s) - t0 )) s"
python bench.py --workload c1 --steps 32 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c1.json 2> gpurun_out/bench_c1.err; tail -2 gpurun_out/bench_c1.err
python - <<'PY'
import json
for n in ("c2","c1"):
    try:
        d=json.load(open(f"gpurun_out/bench_{n}.json"))
        print(n, 'Mrays/s %.0f frac_ext %.3f'%(d['value'], d['roofline']['frac']), d.get('stage_ms_per_step'), 'ms/step %.2f'%d['ms_per_step'], 'build %.1f prep %.1f'%(d['host_bvh_build_s'], d['prepare_s']), d.get('cpu_baseline',{}).get('value'))
    except Exception as e: print(n, "failed", e)
PY
