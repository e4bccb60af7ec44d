# usage: tools/closing_record.sh <series>   (e.g. r3_f)
# The closing record of a round (files gpurun_out/<series>_*, to be copied into profiles/): default bench (with CPU baseline),
# the other configs, kernel stats, PMC counters of configs[2], queue rates, in-process / gloo two-rank lines on one device.
S=${1:-r3_f}
python bench.py > gpurun_out/${S}_c2_bench.json 2> gpurun_out/${S}_c2_bench.err; tail -1 gpurun_out/${S}_c2_bench.err
for wl in c1 c3 c4; do python bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/${S}_${wl}_bench.json 2> gpurun_out/${S}_${wl}_bench.err; done
python bench.py --steps 2 --warmup 1 --gpus 2 --in-process 0,0 --no-cpu-baseline > gpurun_out/${S}_c2_bench_in_process_2x_same_gpu.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --share-gpu > gpurun_out/${S}_c2_bench_two_processes_same_gpu_gloo.json 2>/dev/null
bash tools/kstats.sh ${S} --workload c2 > gpurun_out/${S}_kstats.txt 2>&1; cp gpurun_out/kstats_${S}.csv gpurun_out/${S}_c2_kernel_stats.csv; rm -rf gpurun_out/kstats_${S}
bash tools/pmc.sh ${S} --workload c2 > /dev/null 2>&1; cp gpurun_out/pmc_${S}_p1.log gpurun_out/${S}_c2_pmc_bench_line.log; cp gpurun_out/pmc_${S}_summary.txt gpurun_out/${S}_c2_pmc_summary.txt; rm -rf gpurun_out/pmc_${S}
timeout 600 python tools/queue_kernel_rate.py 64 > gpurun_out/${S}_queue_rate_wavefront_64spp.jsonl 2>&1
python - $S <<'PY'
import json, sys
S = sys.argv[1]
for n in ("c2","c1","c3","c4"):
    try:
        d=json.loads(open(f"gpurun_out/{S}_{n}_bench.json").read().strip().splitlines()[-1])
        print(n, 'Mrays/s %.0f'%d['value'], d.get('stage_ms_per_step'), 'ms/step %.1f'%d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('cpu_baseline',{}).get('value'))
    except Exception as e: print(n, "failed", e)
PY
head -7 gpurun_out/${S}_kstats.txt
ls gpurun_out | grep ${S}
