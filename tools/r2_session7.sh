timeout 900 python -m pytest tests/test_compositing.py tests/test_fog.py tests/test_subsurface.py tests/test_denoiser_inputs.py -m gpu -q > gpurun_out/r2s7_tests.log 2>&1; grep -n "passed\|failed" gpurun_out/r2s7_tests.log | tail -3
timeout 600 python tools/queue_kernel_rate.py 64 > gpurun_out/r2s7_rate_wave64.txt 2>&1; cat gpurun_out/r2s7_rate_wave64.txt
MIPT_LIB_OVERRIDE=$PWD/pathtracer_amd/libmipt_qw3.so timeout 600 python tools/queue_kernel_rate.py 64 > gpurun_out/r2s7_rate_wave64_qw3.txt 2>&1; cat gpurun_out/r2s7_rate_wave64_qw3.txt
