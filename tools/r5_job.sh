S=r5_k
python -m pytest tests/test_bench_contract.py -q -m gpu > gpurun_out/contract.txt 2>&1; grep -E "passed|failed" gpurun_out/contract.txt | tail -1
python bench.py > gpurun_out/${S}_c2_bench.json 2> gpurun_out/${S}_c2_bench.err; tail -1 gpurun_out/${S}_c2_bench.err
for wl in c1 c3 c4; do python bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/${S}_${wl}_bench.json 2> gpurun_out/${S}_${wl}_bench.err; done
python - $S <<'PY'
import json, sys
S = sys.argv[1]
for n in ("c2","c1","c3","c4"):
    d=json.loads(open(f"gpurun_out/{S}_{n}_bench.json").read().strip().splitlines()[-1]); r=d['roofline']
    print(n, 'Mrays/s %.0f'%d['value'], {k:round(v) for k,v in d['stage_ms_per_step'].items()}, 'ms/step %.1f'%d['ms_per_step'], 'frac %.3f'%r['frac'], {k:round(v,3) for k,v in r['issue_model'].items() if k.endswith('_busy')}, r['device_rates_measured_in_this_run'].get('vmem_ns_per_wave_instruction_and_cu'), r['derived_from_pmc_run']['same_library_build'])
PY
