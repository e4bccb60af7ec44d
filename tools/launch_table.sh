# usage: tools/launch_table.sh <wl> [option=value ...]   (on the GPU box)
# Every stage launch of ONE frame of the workload in launch order with its duration (rocprofv3 --kernel-trace; the trace itself is too large
# to travel and is deleted): which depth's extend / any-hit / shade launch costs what.  Output: gpurun_out/launch_table_<wl>.txt
wl=${1:-c2}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lt_$wl -- python3 $R/tools/rank_probe.py $wl ranks=1 "$@" > $R/gpurun_out/lt_$wl.log 2>&1
python3 - $R $wl <<'PY' > $R/gpurun_out/launch_table_$wl.txt
import csv, glob, json, sys
R, wl = sys.argv[1], sys.argv[2]
f = glob.glob(R + "/gpurun_out/lt_%s/*/*kernel_trace.csv" % wl)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "") for r in rows]
gen = [i for i, n in enumerate(names) if n.startswith("k_wf_generate")]
# rank_probe renders the frame twice (warm-up + timed): the second half of the generate launches starts the timed frame
first = gen[len(gen) // 2]
line = [l for l in open(R + "/gpurun_out/lt_%s.log" % wl) if l.startswith("{")]
if line:
    r = json.loads(line[-1])["ranks"][0]
    print("frame: render %.1f ms, extend %.1f, any-hit %.1f, shade %.1f, splat %.1f, %d rays (under the tracer)" % (r["render_ms"], r["extend"], r["shadow"], r["shade"], r["resolve"], r["rays"]))
print("%4s %-28s %10s %10s" % ("#", "kernel", "ms", "start ms"))
t0 = int(rows[first]["Start_Timestamp"])
tot = {}
depth = -1
for i in range(first, len(rows)):
    n = names[i]
    if not (n.startswith("k_wf_") or n.startswith("k_q_") or n.startswith("k_resolve")):
        continue
    d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e6
    if n.startswith("k_wf_generate"):
        depth = -1
    if n.startswith("k_wf_traverse") or n.startswith("k_wf_extend"):
        depth += 1
    key = n[:28] + (" depth %d" % depth if depth >= 0 and not n.startswith("k_resolve") else "")
    tot[key] = tot.get(key, 0.0) + d
    print("%4d %-28s %10.2f %10.1f" % (i - first, key, d, (int(rows[i]["Start_Timestamp"]) - t0) / 1e6))
print("\nsum over the frame's passes, by kernel and depth:")
for k, v in tot.items():
    if v >= 0.05:
        print("  %-40s %9.1f ms" % (k, v))
PY
rm -rf $R/gpurun_out/lt_$wl
cat $R/gpurun_out/launch_table_$wl.txt | tail -30
