# usage: tools/kstats_build.sh <grid>: rocprofv3 kernel trace + stats of the GPU BVH build (two builds of a grid x grid blob) -> gpurun_out/kstats_build_<grid>.csv
g=${1:-1120}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstats_build_$g -- python3 $R/tools/bvh_build_bench.py $g > $R/gpurun_out/kstats_build_$g.log 2>&1
f=$(ls $R/gpurun_out/kstats_build_$g/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/kstats_build_$g.csv
tail -2 $R/gpurun_out/kstats_build_$g.log
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows: print("%-50s calls %5s total_ms %9.2f avg_us %9.1f  %5.1f%%" % (r["Name"][:50], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
