# usage: tools/pmc_issue.sh <tag> [bench args...]  — the issue-side account of the traversal kernels (no thread trace / PC
# sampling on this pool, profiles/r5_a_thread_trace_and_pc_sampling_unavailable.txt): separate rocprofv3 --pmc passes, counters only.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "FETCH_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  if [ -n "$LIMIT" ] && [ $i -gt $LIMIT ]; then break; fi
  timeout 240 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmci_${tag}/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --pmc "$@" > $R/gpurun_out/pmci_${tag}_p$i.log 2>&1 || tail -3 $R/gpurun_out/pmci_${tag}_p$i.log
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmci_${tag} > $R/gpurun_out/pmci_${tag}_summary.txt
grep -A60 "^k_wf_traverse<0>\|^k_wf_traverse<1>\|^k_wf_anyhit" $R/gpurun_out/pmci_${tag}_summary.txt | head -150
rm -rf $R/gpurun_out/pmci_${tag}
