rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "-" "notail" "-" "notail" "- --workload c3 --steps 1" "notail --workload c3 --steps 1"
cp gpurun_out/sweep.log gpurun_out/r5_i_sweep.log
for lib in "" notail; do
  if [ -n "$lib" ]; then export MIPT_LIB_OVERRIDE=$PWD/pathtracer_amd/libmipt_$lib.so; else unset MIPT_LIB_OVERRIDE; fi
  python tools/rank_probe.py c2 ranks=1,8 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d.pop('ranks'); print('$lib', d['nranks'], 't_max', d['t_max_ms'], 't_mean', d['t_mean_ms'], 'n_x_mean_over_t1', d['n_x_mean_over_t1'], 'pred', d['predicted_speedup'])"
done
