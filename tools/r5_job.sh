S=r5_b
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_spheres.py -m gpu -x -q 2>&1 | tail -3
rm -f gpurun_out/sweep.log
bash tools/sweep_libs.sh "- --opt anyhit_wide=0" "-" "any7" "- --workload c1 --opt anyhit_wide=0" "- --workload c1" "any7 --workload c1"  "- --workload c3 --steps 1 --opt anyhit_wide=0" "- --workload c3 --steps 1" "- --workload c4 --steps 1 --opt anyhit_wide=0" "- --workload c4 --steps 1"
cp gpurun_out/sweep.log gpurun_out/${S}_sweep.log
bash tools/pmc_issue.sh ordered --opt anyhit_wide=0 > gpurun_out/${S}_pmci_ordered.txt 2>&1
bash tools/pmc_issue.sh wide8 > gpurun_out/${S}_pmci_wide8.txt 2>&1
tail -70 gpurun_out/${S}_pmci_wide8.txt
for w in c2 c3 c4; do timeout 900 python tools/rank_probe.py $w > gpurun_out/r5_all_ranks_$w.jsonl 2> gpurun_out/r5_all_ranks_$w.err; python - $w <<'PY'
import json,sys
for l in open('gpurun_out/r5_all_ranks_%s.jsonl'%sys.argv[1]):
    d=json.loads(l); d.pop('ranks'); print(d)
PY
done
