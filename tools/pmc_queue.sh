# usage: tools/pmc_queue.sh <tag> <feature> [option=value ...]; counters of the contribution-queue kernels of one tools/queue_kernel_rate.py render
# (separate rocprofv3 --pmc passes, counters only) -> gpurun_out/pmc_<tag>_summary.txt (mean per dispatch x dispatches = total)
tag=$1; f=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_IFETCH SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_ANY" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" \
           "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}/p$i -- python3 $R/tools/queue_kernel_rate.py 64 only=$f "$@" > $R/gpurun_out/pmc_${tag}_p$i.log 2>&1 || tail -3 $R/gpurun_out/pmc_${tag}_p$i.log
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag} > $R/gpurun_out/pmc_${tag}_summary.txt
grep -c . $R/gpurun_out/pmc_${tag}_summary.txt
