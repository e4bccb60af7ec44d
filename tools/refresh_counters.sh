# usage: tools/refresh_counters.sh <series> <git commit>   (on the GPU box)
# The part of the closing record that is tied to the library's source hash, after a change of csrc/ that does not call for the whole record again:
# PMC passes of the four configs -> profiles/pmc_counters.json (copy: gpurun_out/<series>_pmc_counters.json), then the four bench lines.
S=${1:-r5_k}; export MIPT_GIT_COMMIT=${2:-unknown}
ARGS=""
for wl in c2 c1 c3 c4; do
  bash tools/pmc.sh ${S}_${wl} --workload ${wl} > /dev/null 2>&1
  cp gpurun_out/pmc_${S}_${wl}_p1.log gpurun_out/${S}_${wl}_pmc_bench_line.log; cp gpurun_out/pmc_${S}_${wl}_summary.txt gpurun_out/${S}_${wl}_pmc_summary.txt; rm -rf gpurun_out/pmc_${S}_${wl} gpurun_out/pmc_${S}_${wl}_p*.log
  ARGS="$ARGS ${wl}=gpurun_out/${S}_${wl}_pmc_summary.txt:gpurun_out/${S}_${wl}_pmc_bench_line.log:profiles/${S}_${wl}_pmc_summary.txt"
done
python tools/pmc_to_json.py profiles/r2_fetch_calibration.json $ARGS > gpurun_out/${S}_pmc_to_json.txt && cp profiles/pmc_counters.json gpurun_out/${S}_pmc_counters.json
python bench.py > gpurun_out/${S}_c2_bench.json 2> gpurun_out/${S}_c2_bench.err; tail -1 gpurun_out/${S}_c2_bench.err
for wl in c1 c3 c4; do python bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/${S}_${wl}_bench.json 2> gpurun_out/${S}_${wl}_bench.err; done
python - $S <<'PY'
import json, sys
S = sys.argv[1]
for n in ("c2","c1","c3","c4"):
    try:
        d=json.loads(open(f"gpurun_out/{S}_{n}_bench.json").read().strip().splitlines()[-1])
        r=d.get('roofline',{})
        print(n, 'Mrays/s %.0f'%d['value'], {k:round(v,1) for k,v in d.get('stage_ms_per_step',{}).items()}, 'ms/step %.1f'%d['ms_per_step'], 'frac %.3f'%r.get('frac'), 'hbm', r.get('frac_hbm_measured'), r.get('derived_from_pmc_run',{}).get('same_library_build'))
    except Exception as e: print(n, "failed", e)
PY
