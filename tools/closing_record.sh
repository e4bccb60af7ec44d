# usage: tools/closing_record.sh <series> <git commit of the snapshot>   (e.g. r3_h abc1234; run on the GPU box)
# The closing record of a round (files gpurun_out/<series>_*, to be copied into profiles/): kernel stats and PMC counters of
# configs[2] FIRST, so that profiles/r3_pmc_counters.json (copy: gpurun_out/<series>_pmc_counters.json) describes the very library
# the bench lines below are measured with; then the default bench (with CPU baseline), the other configs, the in-process / gloo
# two-rank lines on one device, the queue-pipeline rates with the kernel trace and the counters of one feature.
S=${1:-r3_h}; export MIPT_GIT_COMMIT=${2:-unknown}
bash tools/kstats.sh ${S} --workload c2 > gpurun_out/${S}_kstats.txt 2>&1; cp gpurun_out/kstats_${S}.csv gpurun_out/${S}_c2_kernel_stats.csv; rm -rf gpurun_out/kstats_${S}
bash tools/pmc.sh ${S} --workload c2 > /dev/null 2>&1; cp gpurun_out/pmc_${S}_p1.log gpurun_out/${S}_c2_pmc_bench_line.log; cp gpurun_out/pmc_${S}_summary.txt gpurun_out/${S}_c2_pmc_summary.txt; rm -rf gpurun_out/pmc_${S}
python tools/pmc_to_json.py profiles/r2_fetch_calibration.json c2=gpurun_out/${S}_c2_pmc_summary.txt:gpurun_out/${S}_c2_pmc_bench_line.log:profiles/${S}_c2_pmc_summary.txt > /dev/null && cp profiles/r3_pmc_counters.json gpurun_out/${S}_pmc_counters.json
python bench.py > gpurun_out/${S}_c2_bench.json 2> gpurun_out/${S}_c2_bench.err; tail -1 gpurun_out/${S}_c2_bench.err
for wl in c1 c3 c4; do python bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/${S}_${wl}_bench.json 2> gpurun_out/${S}_${wl}_bench.err; done
python bench.py --steps 2 --warmup 1 --gpus 2 --in-process 0,0 --no-cpu-baseline > gpurun_out/${S}_c2_bench_in_process_2x_same_gpu.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --share-gpu > gpurun_out/${S}_c2_bench_two_processes_same_gpu_gloo.json 2>/dev/null
timeout 600 python tools/queue_kernel_rate.py 64 > gpurun_out/${S}_queue_rate_wavefront_64spp.jsonl 2>&1
bash tools/queue_trace.sh ghost 64 > /dev/null 2>&1; cp gpurun_out/qtrace_ghost.txt gpurun_out/${S}_queue_ghost_photo_launch_sequence.txt; rm -rf gpurun_out/qtrace_ghost
for f in ghost fog; do bash tools/pmc_queue.sh ${S}_q$f $f > /dev/null 2>&1; bash tools/pmc_queue_bytes.sh ${S}_q$f $f > /dev/null 2>&1; cp gpurun_out/pmc_${S}_q${f}_summary.txt gpurun_out/${S}_queue_${f}_pmc_summary.txt; rm -rf gpurun_out/pmc_${S}_q$f; done
python - $S <<'PY'
import json, sys
S = sys.argv[1]
for n in ("c2","c1","c3","c4"):
    try:
        d=json.loads(open(f"gpurun_out/{S}_{n}_bench.json").read().strip().splitlines()[-1])
        print(n, 'Mrays/s %.0f'%d['value'], d.get('stage_ms_per_step'), 'ms/step %.1f'%d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('frac_vmem_issue'), d.get('roofline',{}).get('derived_from_pmc_run'), d.get('cpu_baseline',{}).get('value'))
    except Exception as e: print(n, "failed", e)
PY
head -7 gpurun_out/${S}_kstats.txt
cat gpurun_out/${S}_queue_rate_wavefront_64spp.jsonl | cut -c1-240
ls gpurun_out | grep ${S}
