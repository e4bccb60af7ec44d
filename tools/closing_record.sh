# usage: tools/closing_record.sh <series> <git commit of the snapshot>   (e.g. r4_h abc1234; run on the GPU box)
# The closing record of a round (files gpurun_out/<series>_*, to be copied into profiles/): kernel stats and PMC counters of all four
# configs FIRST, so that profiles/pmc_counters.json (copy: gpurun_out/<series>_pmc_counters.json) describes the very library the
# bench lines below are measured with (bench.py compares __graft_entry__.source_hash()); then the default bench (with CPU baseline),
# the other configs, the in-process / gloo two-rank lines on one device, the rank-cost probes and the queue-pipeline rates.
S=${1:-r6_z}; export MIPT_GIT_COMMIT=${2:-unknown}
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${S}_gputests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/${S}_gputests.txt | tail -2
bash tools/kstats.sh ${S} --workload c2 > gpurun_out/${S}_kstats.txt 2>&1; cp gpurun_out/kstats_${S}.csv gpurun_out/${S}_c2_kernel_stats.csv; rm -rf gpurun_out/kstats_${S}
ARGS=""
for wl in c2 c1 c3 c4; do
  bash tools/pmc.sh ${S}_${wl} --workload ${wl} > /dev/null 2>&1
  cp gpurun_out/pmc_${S}_${wl}_p1.log gpurun_out/${S}_${wl}_pmc_bench_line.log; cp gpurun_out/pmc_${S}_${wl}_summary.txt gpurun_out/${S}_${wl}_pmc_summary.txt; rm -rf gpurun_out/pmc_${S}_${wl} gpurun_out/pmc_${S}_${wl}_p*.log
  ARGS="$ARGS ${wl}=gpurun_out/${S}_${wl}_pmc_summary.txt:gpurun_out/${S}_${wl}_pmc_bench_line.log:profiles/${S}_${wl}_pmc_summary.txt"
done
python tools/pmc_to_json.py profiles/r2_fetch_calibration.json $ARGS > gpurun_out/${S}_pmc_to_json.txt && cp profiles/pmc_counters.json gpurun_out/${S}_pmc_counters.json
# configs[4]'s shade stage is fp64 arithmetic: its instruction mix by precision (bench.py --workload c4: roofline_shade_kernel {bound: fp64})
bash tools/pmc_fp64.sh ${S}_c4 --workload c4 > /dev/null 2>&1; cp gpurun_out/pmcf_${S}_c4_summary.txt gpurun_out/${S}_c4_fp64_mix_pmc_summary.txt; cp gpurun_out/pmcf_${S}_c4_bench_line.log gpurun_out/${S}_c4_fp64_mix_pmc_bench_line.log
python tools/fp64_to_json.py c4=gpurun_out/${S}_c4_fp64_mix_pmc_summary.txt:gpurun_out/${S}_c4_fp64_mix_pmc_bench_line.log:profiles/${S}_c4_fp64_mix_pmc_summary.txt > gpurun_out/${S}_fp64_to_json.txt && cp profiles/fp64_counters.json gpurun_out/${S}_fp64_counters.json
python bench.py > gpurun_out/${S}_c2_bench.json 2> gpurun_out/${S}_c2_bench.err; tail -1 gpurun_out/${S}_c2_bench.err
for wl in c1 c3 c4; do python bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/${S}_${wl}_bench.json 2> gpurun_out/${S}_${wl}_bench.err; done
python bench.py --steps 2 --warmup 1 --gpus 2 --in-process 0,0 --no-cpu-baseline > gpurun_out/${S}_c2_bench_in_process_2x_same_gpu.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --share-gpu > gpurun_out/${S}_c2_bench_two_processes_same_gpu_gloo.json 2>/dev/null
for wl in c2 c3 c4; do python tools/rank_probe.py $wl > gpurun_out/${S}_all_ranks_${wl}.jsonl 2>/dev/null; done
bash tools/pmc_issue.sh ${S} > /dev/null 2>&1; cp gpurun_out/pmci_${S}_summary.txt gpurun_out/${S}_c2_pmc_issue_summary.txt; rm -f gpurun_out/pmci_${S}_p*.log
for l in 0 1 8 16; do python tools/progressive_rate.py progressive_lookahead=$l spp=64,256 2>&1 | tail -1; done > gpurun_out/${S}_progressive_rate.txt
timeout 900 python tools/scale_preflight.py > gpurun_out/${S}_scale_preflight.jsonl 2> gpurun_out/${S}_scale_preflight.err
timeout 600 python tools/queue_kernel_rate.py 64 > gpurun_out/${S}_queue_rate_wavefront_64spp.jsonl 2>&1
python - $S <<'PY'
import json, sys
S = sys.argv[1]
for n in ("c2","c1","c3","c4"):
    try:
        d=json.loads(open(f"gpurun_out/{S}_{n}_bench.json").read().strip().splitlines()[-1])
        r=d.get('roofline',{})
        print(n, 'Mrays/s %.0f'%d['value'], d.get('stage_ms_per_step'), 'ms/step %.1f'%d['ms_per_step'], 'frac', r.get('frac'), 'hbm', r.get('frac_hbm_measured'), {k:round(v,3) for k,v in (r.get('issue_model') or {}).items() if k.endswith('_busy')}, r.get('derived_from_pmc_run',{}).get('same_library_build'), d.get('cpu_baseline',{}).get('value'), 'init+upload', round(d['host_bvh_build_s']+d['prepare_s'],4))
    except Exception as e: print(n, "failed", e)
PY
head -7 gpurun_out/${S}_kstats.txt
cut -c1-240 gpurun_out/${S}_queue_rate_wavefront_64spp.jsonl
ls gpurun_out | grep ${S}
# differential fuzzing on the same library (test infrastructure: the oracle is the checker)
timeout 900 python tests/tools/fuzz_parity.py 500 401 > gpurun_out/${S}_fuzz_parity_500_scenes.txt 2>&1; tail -1 gpurun_out/${S}_fuzz_parity_500_scenes.txt
timeout 900 python tests/tools/fuzz_parity.py 300 402 --queue > gpurun_out/${S}_fuzz_parity_300_scenes_queue.txt 2>&1; tail -1 gpurun_out/${S}_fuzz_parity_300_scenes_queue.txt
timeout 900 python tests/tools/fuzz_parity.py 250 403 --queue --spheres > gpurun_out/${S}_fuzz_parity_250_scenes_queue_spheres.txt 2>&1; tail -1 gpurun_out/${S}_fuzz_parity_250_scenes_queue_spheres.txt
timeout 900 python tests/tools/fuzz_parity.py 150 404 --queue --bare-spheres > gpurun_out/${S}_fuzz_parity_150_scenes_queue_bare_spheres.txt 2>&1; tail -1 gpurun_out/${S}_fuzz_parity_150_scenes_queue_bare_spheres.txt
timeout 900 python tests/tools/fuzz_parity.py 200 405 --kind=merl --merl-tiers > gpurun_out/${S}_fuzz_parity_200_scenes_measured_brdf_three_tiers.txt 2>&1; tail -1 gpurun_out/${S}_fuzz_parity_200_scenes_measured_brdf_three_tiers.txt
