# usage: tools/progressive_trace.sh [tag]: rocprofv3 kernel + memory-copy trace of Raytracer::render_image (one pass per sample, buffers published after
# every pass) on configs[1]'s scene -> gpurun_out/ptrace_<tag>.txt: the launches and copies of three passes in the middle with durations and gaps
tag=${1:-a}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/ptrace_$tag -- python3 $R/tools/progressive_rate.py > $R/gpurun_out/ptrace_$tag.log 2>&1
k=$(ls $R/gpurun_out/ptrace_$tag/*/*kernel_trace.csv | head -1); m=$(ls $R/gpurun_out/ptrace_$tag/*/*memory_copy_trace.csv | head -1)
python3 - $k $m > $R/gpurun_out/ptrace_$tag.txt <<'PY'
import csv,sys
ev=[]
for r in csv.DictReader(open(sys.argv[1])): ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:48]))
for r in csv.DictReader(open(sys.argv[2])): ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")+" "+r.get("Bytes", r.get("Size",""))))
ev.sort()
# the last k_wf_generate launches mark passes of the last render: print passes 30..32 of the final render_image
gens=[i for i,e in enumerate(ev) if e[2].startswith("k_wf_generate")]
i0=gens[-34]; i1=gens[-31]
t0=ev[i0][0]; prev=t0
for s,e,n in ev[i0:i1]:
    print("%9.3f ms  gap %8.1f us  dur %8.1f us  %s"%((s-t0)/1e6,(s-prev)/1e3,(e-s)/1e3,n)); prev=max(prev,e)
print("three passes: %.3f ms"%((ev[i1][0]-t0)/1e6))
PY
tail -4 $R/gpurun_out/ptrace_$tag.log; cat $R/gpurun_out/ptrace_$tag.txt | head -80
rm -rf $R/gpurun_out/ptrace_$tag
