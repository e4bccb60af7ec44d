# round 2, the record: default bench (with CPU baseline), the other configs, kernel stats, PMC counters, queue rates, in-process / gloo two-rank lines
python bench.py > gpurun_out/r2_d_c2_bench.json 2> gpurun_out/r2_d_c2_bench.err; tail -1 gpurun_out/r2_d_c2_bench.err
for wl in c1 c3 c4; do python bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r2_d_${wl}_bench.json 2> gpurun_out/r2_d_${wl}_bench.err; done
python bench.py --steps 2 --warmup 1 --in-process 0,0 --no-cpu-baseline > gpurun_out/r2_d_c2_bench_in_process_2x_same_gpu.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --share-gpu > gpurun_out/r2_d_c2_bench_two_processes_same_gpu_gloo.json 2>/dev/null
bash tools/kstats.sh r2d --workload c2 > gpurun_out/r2_d_kstats.txt 2>&1; cp gpurun_out/kstats_r2d.csv gpurun_out/r2_d_c2_kernel_stats.csv; rm -rf gpurun_out/kstats_r2d
bash tools/pmc.sh r2d --workload c2 > /dev/null 2>&1; rm -rf gpurun_out/pmc_r2d
timeout 600 python tools/queue_kernel_rate.py 64 > gpurun_out/r2_d_queue_rate_wavefront_64spp.jsonl 2>&1
python - <<'PY'
import json
for n in ("c2","c1","c3","c4"):
    try:
        d=json.load(open(f"gpurun_out/r2_d_{n}_bench.json"))
        print(n, 'Mrays/s %.0f'%d['value'], d.get('stage_ms_per_step'), 'ms/step %.1f'%d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('cpu_baseline',{}).get('value'))
    except Exception as e: print(n, "failed", e)
PY
head -7 gpurun_out/r2_d_kstats.txt
