for i in 1 2 3; do for lib in - qlw4 qlw2; do
if [ "$lib" = "-" ]; then unset MIPT_LIB_OVERRIDE; else export MIPT_LIB_OVERRIDE=$PWD/pathtracer_amd/libmipt_$lib.so; fi
timeout 600 python tools/queue_kernel_rate.py 64 only=fog 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if d['feature']=='fog': print('$lib', d['Mrays_per_s'], d['seconds'], d['kernel_ms'])"; done; done
