# usage: tools/rank_kstats.sh <wl>: per-kernel time of the whole frame rendered as 1 rank and as 8 ranks (one after the other on this GPU), from rocprofv3 --kernel-trace --stats
wl=${1:-c2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in 1 8; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rks_$n -- python3 $R/tools/rank_probe.py $wl ranks=$n > $R/gpurun_out/rks_$n.log 2>&1
done
python3 - $R <<'PY'
import csv,sys,glob
R=sys.argv[1]
t={}
for n in (1,8):
    f=glob.glob(R+"/gpurun_out/rks_%d/*/*kernel_stats.csv"%n)[0]
    t[n]={r["Name"].split("(")[0][:40]:(float(r["TotalDurationNs"])/1e6,int(r["Calls"])) for r in csv.DictReader(open(f))}
# renders: n=1 -> 2 full frames (warm-up + timed); n=8 -> 9 eighths
print("%-40s %10s %10s %7s   calls 1 / 8"%("kernel","1 rank ms","8 ranks ms","ratio"))
for k in sorted(t[1], key=lambda k:-t[1][k][0]):
    a=t[1][k][0]/2; b=t[8].get(k,(0,0))[0]/9*8
    if a>0.5: print("%-40s %10.1f %10.1f %7.3f   %d / %d"%(k,a,b,b/a,t[1][k][1],t[8].get(k,(0,0))[1]))
PY
rm -rf $R/gpurun_out/rks_1 $R/gpurun_out/rks_8
