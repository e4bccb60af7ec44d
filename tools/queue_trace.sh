# usage: tools/queue_trace.sh <feature> [spp]: rocprofv3 kernel trace of one contribution-queue render -> gpurun_out/qtrace_<feature>.txt (launch sequence with durations)
f=${1:-ghost}; spp=${2:-64}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/qtrace_$f -- python3 $R/tools/queue_kernel_rate.py $spp only=$f > $R/gpurun_out/qtrace_$f.log 2>&1
t=$(ls $R/gpurun_out/qtrace_$f/*/*kernel_trace.csv | head -1)
python3 - $t > $R/gpurun_out/qtrace_$f.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last render = the launches after the last k_q_begin group: print the second half
names=[r["Kernel_Name"] for r in rows]
half=len(rows)//2
t0=int(rows[half]["Start_Timestamp"])
prev_end=t0
for r in rows[half:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    n=r["Kernel_Name"].split("(")[0][:60]
    print("%10.3f ms  gap %8.1f us  dur %9.1f us  grid %8s  %s" % ((s-t0)/1e6,(s-prev_end)/1e3,(e-s)/1e3,r.get("Grid_Size_X", r.get("Grid_Size","")),n))
    prev_end=e
PY
tail -3 $R/gpurun_out/qtrace_$f.log
