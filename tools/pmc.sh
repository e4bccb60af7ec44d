# usage: tools/pmc.sh <tag> [bench args...]; runs separate rocprofv3 --pmc passes (counters only, no tracing)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
SKIP=${SKIP:-0}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
           "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  if [ $i -le $SKIP ]; then continue; fi
  if [ -n "$LIMIT" ] && [ $i -gt $LIMIT ]; then break; fi
  date +%T
  timeout 240 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --pmc "$@" > $R/gpurun_out/pmc_${tag}_p$i.log 2>&1 || tail -3 $R/gpurun_out/pmc_${tag}_p$i.log
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag} > $R/gpurun_out/pmc_${tag}_summary.txt; cat $R/gpurun_out/pmc_${tag}_summary.txt
