# GPU session for the record: tests, default bench (with CPU baseline), configs[1] bench, kernel stats, PMC
python -m pytest tests -m gpu -q -x > gpurun_out/gputests.log 2>&1; tail -2 gpurun_out/gputests.log
t0=$(date +%s)
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -2 gpurun_out/bench_default.err
echo "default bench wall $(( $(date +%s) - t0 )) s"
python bench.py --workload c1 --no-cpu-baseline > gpurun_out/bench_c1.json 2> gpurun_out/bench_c1.err; tail -2 gpurun_out/bench_c1.err
bash tools/kstats.sh c2 --workload c2 > gpurun_out/kstats_c2.txt 2>&1; head -8 gpurun_out/kstats_c2.txt
bash tools/pmc.sh c2 --workload c2 > /dev/null 2>&1
python - <<'PY'
import json
for n in ("default","c1"):
    try:
        d=json.load(open(f"gpurun_out/bench_{n}.json"))
        print(n, 'Mrays/s %.0f frac %.3f'%(d['value'], d['roofline']['frac']), d.get('stage_ms_per_step'), 'ms/step %.2f'%d['ms_per_step'], d.get('cpu_baseline',{}).get('value'))
    except Exception as e: print(n, "failed", e)
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/gpu_more.sh
