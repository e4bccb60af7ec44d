cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktrace_c2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/ktrace_c2.log 2>&1
python3 - $(ls $R/gpurun_out/ktrace_c2/*/*kernel_trace.csv | head -1) <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: take the last 30 launches of wavefront kernels
sel=[r for r in rows if r["Kernel_Name"].startswith(("void k_wf","k_wf","void k_resolve","__amd_rocclr"))]
last=sel[-34:]
prev=None; gaps=0; tot=0
for r in last:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    g=(s-prev)/1e3 if prev else 0
    print("gap %9.1f us  dur %10.1f us  %s" % (g,(e-s)/1e3,r["Kernel_Name"].split("(")[0][:50]))
    if prev and g<5000: gaps+=g
    tot+=(e-s)/1e3
    prev=e
print("sum dur ms", tot/1e3, "gaps ms", gaps/1e3)
PY
