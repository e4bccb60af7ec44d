# round 2, GPU session 3: multi-GPU tests, bench modes, FETCH_SIZE calibration, PMC counters of the current build
python -m pytest tests/test_multi_gpu.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r2s3_tests.log 2>&1; tail -5 gpurun_out/r2s3_tests.log
python bench.py --steps 2 --warmup 1 > gpurun_out/r2s3_bench_default.json 2> gpurun_out/r2s3_bench_default.err; tail -2 gpurun_out/r2s3_bench_default.err; cut -c1-600 gpurun_out/r2s3_bench_default.json
python bench.py --steps 2 --warmup 1 --in-process 0,0 --no-cpu-baseline > gpurun_out/r2s3_bench_inproc.json 2> gpurun_out/r2s3_bench_inproc.err; tail -2 gpurun_out/r2s3_bench_inproc.err; cut -c1-900 gpurun_out/r2s3_bench_inproc.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --share-gpu > gpurun_out/r2s3_bench_gloo2.json 2> gpurun_out/r2s3_bench_gloo2.err; tail -2 gpurun_out/r2s3_bench_gloo2.err; cut -c1-900 gpurun_out/r2s3_bench_gloo2.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/fetchcal -- python3 $R/tools/fetch_calibration.py run > $R/gpurun_out/r2s3_fetchcal_run.log 2>&1; tail -1 $R/gpurun_out/r2s3_fetchcal_run.log
cd $R && python3 tools/fetch_calibration.py report gpurun_out/fetchcal > gpurun_out/r2_fetch_calibration.json; cat gpurun_out/r2_fetch_calibration.json; rm -rf gpurun_out/fetchcal
bash tools/pmc.sh r2c2 --workload c2 > /dev/null 2>&1; rm -rf gpurun_out/pmc_r2c2; tail -3 gpurun_out/pmc_r2c2_p1.log | cut -c1-300
