# usage: tools/pmc_fp64.sh <tag> [bench args...]   (on the GPU box)
# The floating-point instruction mix of every kernel of a bench run, by precision: rocprofv3 --pmc passes with the SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F{32,64}
# counters (wave-instructions) beside SQ_INSTS_VALU and the active-lane pair.  For the measured-BRDF shade tiers of configs[4], whose time is fp64 arithmetic
# (glibc's acos / atan2 / sincos restated), not bytes: bench.py turns the summary into roofline_shade_kernel {bound: "fp64"} (profiles/fp64_counters.json).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_MOPS_F64" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcf_${tag}/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --pmc "$@" > $R/gpurun_out/pmcf_${tag}_p$i.log 2>&1 || tail -3 $R/gpurun_out/pmcf_${tag}_p$i.log
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcf_${tag} > $R/gpurun_out/pmcf_${tag}_summary.txt
cp $R/gpurun_out/pmcf_${tag}_p1.log $R/gpurun_out/pmcf_${tag}_bench_line.log
rm -rf $R/gpurun_out/pmcf_${tag} $R/gpurun_out/pmcf_${tag}_p[0-9].log
grep -A40 "k_wf_shade<4" $R/gpurun_out/pmcf_${tag}_summary.txt | head -60
