# round 2, GPU session 8: full GPU suite with the wavefront queue, big fuzz runs, kernel stats of the queue stages
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2s8_tests_all.log 2>&1; grep -n "passed\|failed" gpurun_out/r2s8_tests_all.log | tail -3
timeout 1200 python tests/tools/fuzz_parity.py 600 2026 --queue > gpurun_out/r2_fuzz_parity_600_scenes_queue_wavefront.txt 2>&1; tail -1 gpurun_out/r2_fuzz_parity_600_scenes_queue_wavefront.txt
timeout 900 python tests/tools/fuzz_parity.py 300 2027 > gpurun_out/r2_fuzz_parity_300_scenes.txt 2>&1; tail -1 gpurun_out/r2_fuzz_parity_300_scenes.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstats_queue -- python3 $R/tools/queue_kernel_rate.py 64 > $R/gpurun_out/r2s8_kstats_queue.log 2>&1
cp $(ls $R/gpurun_out/kstats_queue/*/*kernel_stats.csv | head -1) $R/gpurun_out/r2_queue_wavefront_kernel_stats.csv; rm -rf $R/gpurun_out/kstats_queue
head -12 $R/gpurun_out/r2_queue_wavefront_kernel_stats.csv | cut -c1-160
