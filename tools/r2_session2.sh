# round 2, GPU session 2: phase profile of the traversal kernels, scheduling-parameter sweep, PMC of the LDS-top variant
python tools/simd_prof.py c2 > gpurun_out/r2s2_simd_prof_c2.txt 2>&1; tail -12 gpurun_out/r2s2_simd_prof_c2.txt
rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "- --workload c2 --opt inner_min=8" "- --workload c2 --opt inner_min=24" "- --workload c2 --opt inner_min=32" "- --workload c2 --opt inner_min=40" \
  "- --workload c2 --opt refill_threshold=24" "- --workload c2 --opt refill_threshold=30" "- --workload c2 --opt refill_threshold=44" \
  "- --workload c2 --opt resolve_rows=4" "- --workload c2 --opt resolve_rows=6" "- --workload c2 --opt resolve_rows=12"
cp gpurun_out/sweep.log gpurun_out/r2s2_sweep.log
# PMC: L1 accesses / LDS instructions of the traversal kernels, default build vs top-of-tree in LDS
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in base top127; do
  if [ $lib = base ]; then unset MIPT_LIB_OVERRIDE; else export MIPT_LIB_OVERRIDE=$R/pathtracer_amd/libmipt_$lib.so; fi
  i=0
  for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
             "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_r2s2_$lib/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --pmc --workload c2 > $R/gpurun_out/pmc_r2s2_${lib}_p$i.log 2>&1 || tail -3 $R/gpurun_out/pmc_r2s2_${lib}_p$i.log
  done
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_r2s2_$lib > $R/gpurun_out/r2s2_pmc_${lib}_summary.txt
  rm -rf $R/gpurun_out/pmc_r2s2_$lib
done
grep -A12 "k_wf_traverse<0>" $R/gpurun_out/r2s2_pmc_base_summary.txt | head -16; grep -A12 "k_wf_traverse<0>" $R/gpurun_out/r2s2_pmc_top127_summary.txt | head -16
