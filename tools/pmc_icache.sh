# usage: tools/pmc_icache.sh <tag> [bench args...]: instruction-fetch and scalar-cache counters of a short bench run (separate rocprofv3 --pmc passes,
# counters only) -> gpurun_out/pmc_<tag>_summary.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_IFETCH SQ_INSTS_VALU" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" \
           "SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --pmc "$@" > $R/gpurun_out/pmc_${tag}_p$i.log 2>&1 || tail -3 $R/gpurun_out/pmc_${tag}_p$i.log
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag} > $R/gpurun_out/pmc_${tag}_summary.txt
