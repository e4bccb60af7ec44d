# round 2, GPU session 4: shade kernel rolled vs unrolled (instruction-cache footprint), resolve unroll 8, FETCH_SIZE calibration
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -x > gpurun_out/r2s4_tests.log 2>&1; grep -n "passed\|failed" gpurun_out/r2s4_tests.log | tail -2
rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "- --workload c2 --spp-per-step 256" "unrolled --workload c2 --spp-per-step 256" "rolled4 --workload c2 --spp-per-step 256" "- --workload c1 --spp-per-step 256" "unrolled --workload c1 --spp-per-step 256" "rolled4 --workload c1 --spp-per-step 256" \
   "- --workload c3 --spp-per-step 256" "unrolled --workload c3 --spp-per-step 256" "- --workload c2 --spp-per-step 256 --opt resolve_rows=8" "- --workload c2 --spp-per-step 256 --opt resolve_rows=16"
cp gpurun_out/sweep.log gpurun_out/r2s4_sweep.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/fetchcal -- python3 $R/tools/fetch_calibration.py run > $R/gpurun_out/r2s4_fetchcal_run.log 2>&1; grep "stream_gb" $R/gpurun_out/r2s4_fetchcal_run.log
cd $R && python3 tools/fetch_calibration.py report gpurun_out/fetchcal > gpurun_out/r2_fetch_calibration.json; cat gpurun_out/r2_fetch_calibration.json; rm -rf gpurun_out/fetchcal
