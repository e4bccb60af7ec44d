# usage: tools/pmc_queue_bytes.sh <tag> <feature> [option=value ...]: fabric bytes (FETCH_SIZE / WRITE_SIZE, separate passes) and L2 requests of the queue kernels
tag=$1; f=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=10
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_WRREQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}/p$i -- python3 $R/tools/queue_kernel_rate.py 64 only=$f "$@" > $R/gpurun_out/pmc_${tag}_p$i.log 2>&1 || tail -3 $R/gpurun_out/pmc_${tag}_p$i.log
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag} > $R/gpurun_out/pmc_${tag}_summary.txt
grep -c . $R/gpurun_out/pmc_${tag}_summary.txt
