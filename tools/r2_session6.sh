# round 2, GPU session 6: the contribution queue as wavefront stages — parity first, then rates
timeout 900 python -m pytest tests/test_compositing.py tests/test_fog.py tests/test_subsurface.py tests/test_denoiser_inputs.py -m gpu -q -x > gpurun_out/r2s6_tests.log 2>&1; grep -n "passed\|failed\|rror" gpurun_out/r2s6_tests.log | tail -5
timeout 600 python tests/tools/fuzz_parity.py 60 777 --queue > gpurun_out/r2s6_fuzz60.txt 2>&1; tail -3 gpurun_out/r2s6_fuzz60.txt
timeout 600 python tools/queue_kernel_rate.py 8 > gpurun_out/r2s6_rate_wave.txt 2>&1; cat gpurun_out/r2s6_rate_wave.txt
timeout 600 python tools/queue_kernel_rate.py 8 queue_wavefront=0 > gpurun_out/r2s6_rate_thread.txt 2>&1; cat gpurun_out/r2s6_rate_thread.txt
