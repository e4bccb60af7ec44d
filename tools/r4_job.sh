mkdir -p gpurun_out/r4m
timeout 900 python -m pytest tests/test_compositing.py tests/test_fog.py tests/test_subsurface.py tests/test_spheres.py tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/r4m/gputests.txt 2>&1; grep -E "passed|failed" gpurun_out/r4m/gputests.txt; grep -E "^E |FAILED" gpurun_out/r4m/gputests.txt | head
for i in 1 2; do timeout 600 python tools/queue_kernel_rate.py 64 2>&1 | grep -v '"none"' | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d['feature'], d['Mrays_per_s'], 'passes', d['passes'], 'fallback', d['samples_through_the_fallback'], d['seconds'], d['kernel_ms'])"; done
