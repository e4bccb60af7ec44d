# usage: tools/kstats.sh <tag> [bench args]: rocprofv3 kernel trace + stats of a short bench run -> gpurun_out/kstats_<tag>/
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstats_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --pmc "$@" > $R/gpurun_out/kstats_$tag.log 2>&1
f=$(ls $R/gpurun_out/kstats_$tag/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/kstats_$tag.csv
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows: print("%-60s calls %5s total_ms %9.2f avg_us %9.1f  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
